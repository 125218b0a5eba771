run() { echo -n "$1: "; env $1 timeout 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-latency 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['kernel_ms'],4), round(d['ms_per_step'],4))"; }
for v in w4 w5 w4 w5; do run XGM_LIB_PATH=$GRAFT_REPO_ROOT/xapiand_amd/csrc/ab/libxgm_$v.so; done
