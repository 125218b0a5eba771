cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-include-regex "xgm_merge" --output-format csv -d gpurun_out/${tag}_pmc_merge -- python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-latency --threads 0 > gpurun_out/${tag}_pmc_merge.log 2>&1
python tools/pmc_parse.py gpurun_out/${tag}_pmc_merge gpurun_out/${tag}_pmc_merge gpurun_out/${tag}_pmc_merge 2>/dev/null | sort -u | head -20
