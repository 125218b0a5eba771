# round 5, call A: A/B main vs branch (stripe-loop offsets lane-resident, OR 16-byte loads) on C2 / C3 / C5 + parity of the branch build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
bash tools/ab_run.sh "--no-other-configs" default next
bash tools/ab_run.sh "--no-other-configs --op OR --terms 5 --topk 100 --steps 8" default next
bash tools/ab_run.sh "--no-other-configs --op PHRASE --topk 10 --steps 8" default next
done > gpurun_out/r5a_ab.txt 2>&1
cat gpurun_out/r5a_ab.txt
XGM_LIB_PATH=$PWD/xapiand_amd/csrc/ab/libxgm_next.so timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_flat.py -m gpu -x -q > gpurun_out/r5a_parity_next.log 2>&1; tail -3 gpurun_out/r5a_parity_next.log
