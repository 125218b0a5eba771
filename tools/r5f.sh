cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OR="--no-other-configs --op OR --terms 5 --topk 100 --steps 8"
{
XGM_NO_OR_FLAT=1 bash tools/ab_run.sh "$OR" default | sed 's/^default/block-decode/'
bash tools/ab_run.sh "$OR" default | sed 's/^default/flat/'
XGM_NO_OR_FLAT=1 bash tools/ab_run.sh "$OR" default | sed 's/^default/block-decode/'
bash tools/ab_run.sh "$OR" default | sed 's/^default/flat/'
} > gpurun_out/r5f_ab.txt 2>&1
cat gpurun_out/r5f_ab.txt
timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "C3 or or5" > gpurun_out/r5f_pytest.log 2>&1; tail -3 gpurun_out/r5f_pytest.log
timeout 600 python -m pytest tests/test_gpu_variants.py -m gpu -x -q -k "OR_FLAT or ORW2 or SEED" > gpurun_out/r5f_pytest2.log 2>&1; tail -3 gpurun_out/r5f_pytest2.log
