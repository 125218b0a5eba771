run() { echo -n "$1 [$2]: "; env $1 timeout 120 python bench.py --steps 12 --warmup 2 --no-cpu-baseline $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['kernel_ms'],3), round(d['p50_latency_us'],1), round(d['roofline']['frac'],3))"; }
run XGM_X=0 ""
run XGM_NO_AND_KERNEL=1 ""
run XGM_X=0 "--stripe-bits 12"
run XGM_X=0 "--stripe-bits 11"
