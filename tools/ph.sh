# GPU evidence: parity, then C5 (PHRASE) and C3 (OR) throughput
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/t.log
run() { echo -n "$1 $2: "; env $1 timeout 300 python bench.py $2 --steps 8 --warmup 2 --no-cpu-baseline 2>gpurun_out/b.err | tee gpurun_out/last.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['roofline']['kernel'], round(d['roofline']['kernel_ms'],3), round(d['p50_latency_us'],1), round(d['roofline']['frac'],3))" || tail -3 gpurun_out/b.err; }
run XGM_X=1 "--op PHRASE --topk 10"
cp gpurun_out/last.json gpurun_out/bench_phrase.json
run XGM_X=1 "--op OR --terms 5 --topk 100"
cp gpurun_out/last.json gpurun_out/bench_or5.json

timeout 300 python tools/phase.py --op OR --terms 5 --topk 100 2>&1 | grep "PHASES"
