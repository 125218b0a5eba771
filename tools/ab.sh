run() { echo -n "$1: "; env $1 timeout 120 python bench.py --op OR --terms 5 --topk 100 --steps 10 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['kernel_ms'],3))"; }
run XGM_X=1
XGM_LIB_PATH=$GRAFT_REPO_ROOT/xapiand_amd/csrc/ab/libxgm_timers.so timeout 300 python tools/phase.py --op OR --terms 5 --topk 100 --no-latency 2>&1 | grep PHASES
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "random_queries or other_stripe or edge_cases or batch_equals" 2>&1 | tail -3
