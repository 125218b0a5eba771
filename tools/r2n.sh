cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_queries" 2>&1 | tail -15
XGM_OR_SEED_SCALE=8 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_queries" 2>&1 | tail -3
XGM_NO_DENSE=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_queries" 2>&1 | tail -3
