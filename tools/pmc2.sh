cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "SPI_RA_[A-Z_0-9]+|SPI_CSN[A-Z_0-9]*|SQ_BUSY_CU_CYCLES|SQ_LEVEL_WAVES|SQ_WAVES_[A-Z_]+" | sort -u | tr '\n' ' ' > gpurun_out/spi_list.txt; cat gpurun_out/spi_list.txt | cut -c1-1500
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 5 150 rocprofv3 --kernel-trace --pmc SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_SGPR_SIMD_FULL_CSN SPI_RA_BAR_CU_FULL_CSN SPI_RA_TMP_STALL_CSN --kernel-include-regex "xgm_and" --output-format csv -d gpurun_out/pmc_spi -- $B > gpurun_out/pmc_spi.log 2>&1
tail -3 gpurun_out/pmc_spi.log | cut -c1-300
