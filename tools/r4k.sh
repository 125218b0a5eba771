cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_positional.py tests/test_gpu_parity.py tests/test_gpu_flat.py -m gpu -q -k "C5 or C2 or positional or phrase or flat or random" > gpurun_out/r04k_pytest.log 2>&1; tail -3 gpurun_out/r04k_pytest.log
bash tools/ab_run.sh "--op PHRASE --topk 10 --steps 6 --warmup 1 --no-other-configs" default pipe1 pipe2 pipe4
bash tools/ab_run.sh "--no-other-configs" default pipe1 pipe2 pipe4
bash tools/ab_run.sh "--op AND --terms 2 --steps 6 --no-other-configs" default pipe1
