# usage: tools/pmc3.sh <tag> <kernel-regex> <bench args...>: SQ pass + HBM read pass for one kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; rx=$2; shift 2
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency $*"
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-include-regex "$rx" --output-format csv -d gpurun_out/pmc_${tag}_a -- $B > gpurun_out/pmc_${tag}_a.log 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_REQ_sum --kernel-include-regex "$rx" --output-format csv -d gpurun_out/pmc_${tag}_b -- $B > gpurun_out/pmc_${tag}_b.log 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-include-regex "$rx" --output-format csv -d gpurun_out/pmc_${tag}_c -- $B > gpurun_out/pmc_${tag}_c.log 2>&1
python tools/pmc_parse.py gpurun_out/pmc_${tag}_a gpurun_out/pmc_${tag}_b gpurun_out/pmc_${tag}_c
