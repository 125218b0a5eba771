# GPU evidence: parity, then C2 / C3 / C5 throughput
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/t.log
run() { echo -n "$1 $2: "; env $1 timeout 300 python bench.py $2 --steps 40 --warmup 4 --no-cpu-baseline 2>gpurun_out/b.err | tee gpurun_out/last.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['roofline']['kernel'], round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), round(d['p50_latency_us'],1), round(d['roofline']['frac'],3))" || tail -3 gpurun_out/b.err; }
run XGM_X=1 "--op AND"
run XGM_X=1 "--op OR --terms 5 --topk 100"
run XGM_X=1 "--op PHRASE --topk 10"
bash tools/trace.sh 2>&1 | tail -4
