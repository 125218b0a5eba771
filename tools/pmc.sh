# usage: tools/pmc.sh <tag> ; collects two PMC passes for the AND kernel on a short bench run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-include-regex "xgm_and|xgm_match" --output-format csv -d gpurun_out/pmc_${tag}_a -- $B > gpurun_out/pmc_${tag}_a.log 2>&1
timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM --kernel-include-regex "xgm_and|xgm_match" --output-format csv -d gpurun_out/pmc_${tag}_b -- $B > gpurun_out/pmc_${tag}_b.log 2>&1
timeout -k 5 150 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-include-regex "xgm_and|xgm_match" --output-format csv -d gpurun_out/pmc_${tag}_c -- $B > gpurun_out/pmc_${tag}_c.log 2>&1
find gpurun_out/pmc_${tag}_* -name "*.csv" | head; 
