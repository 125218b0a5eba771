run() { echo -n "$1: "; env $1 timeout 120 python bench.py --steps 12 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['kernel_ms'],3), round(d['p50_latency_us'],1))"; }
run XGM_X=1
run XGM_LIB_PATH=$GRAFT_REPO_ROOT/xapiand_amd/csrc/libxgm_w3.so
timeout 300 python bench.py --op PHRASE --topk 10 --steps 6 --warmup 1 --no-cpu-baseline > gpurun_out/bench_phrase.json 2>gpurun_out/phrase.err; python -c "import json;d=json.load(open('gpurun_out/bench_phrase.json'));print('PHRASE',round(d['value']),d['roofline']['kernel_ms'],d['p50_latency_us'])"; tail -2 gpurun_out/phrase.err
