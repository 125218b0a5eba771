# Round-2 run E: positional on probe path + server mode: tests, C5 bench, C2 bench short with server leg
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout 1500 python -m pytest tests/test_gpu_positional.py tests/test_gpu_parity.py tests/test_gpu_mixed.py tests/test_gpu_hook_b1.py tests/test_gpu_variants.py "tests/test_gpu_configs.py::test_config_scale_parity[C5_phrase_top10]" "tests/test_gpu_configs.py::test_config_scale_parity[C2_and3_top10]" -m gpu -q > gpurun_out/${tag}_pytest.log 2>&1; tail -25 gpurun_out/${tag}_pytest.log | cut -c1-250
timeout 400 python bench.py --op PHRASE --topk 10 --steps 10 --warmup 2 --ref-docs 0 --cpu-seconds 3 --threads 0 > gpurun_out/${tag}_bench_phrase.json 2>gpurun_out/${tag}_phrase.err; tail -2 gpurun_out/${tag}_phrase.err
timeout 400 python bench.py --no-cpu-baseline --threads 64 > gpurun_out/${tag}_bench_c2.json 2>gpurun_out/${tag}_c2.err; tail -2 gpurun_out/${tag}_c2.err
python - <<PY
import json
for n in ('phrase','c2'):
    try:
        d=json.load(open('gpurun_out/${tag}_bench_%s.json'%n)); r=d['roofline']
        print(n,round(d['value']),r['kernel'],r['kernel_ms'],d['p50_latency_us'],'frac',r['frac'],r['basis'],'alg',r['algorithmic']['frac'],r['model_counts'],d.get('server_mode'))
    except Exception as e: print(n,'failed',e)
PY
