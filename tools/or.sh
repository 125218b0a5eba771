# GPU evidence for the disjunction kernel: parity, then C3 throughput and diagnostics
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/t.log
run() { echo -n "$1 $2: "; env $1 timeout 300 python bench.py --op OR --terms 5 --topk 100 --steps 8 --warmup 2 --no-cpu-baseline $2 2>gpurun_out/or.err | tee gpurun_out/or_last.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['roofline']['kernel'], round(d['roofline']['kernel_ms'],3), round(d['p50_latency_us'],1), round(d['roofline']['frac'],3))" || tail -3 gpurun_out/or.err; }
run XGM_X=1
cp gpurun_out/or_last.json gpurun_out/bench_or5.json
run XGM_NO_BOUND_SUM=1
timeout 300 python tools/units.py --op OR --terms 5 --topk 100 2>&1 | grep "UNITS" | grep -v "concurrency per\|slowest"
timeout 300 python tools/phase.py --op OR --terms 5 --topk 100 2>&1 | grep "PHASES"
