# Round-2 run F: two-sided operators after the P3c rework
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mixed.py "tests/test_gpu_configs.py::test_config_scale_parity[and_not_2x2_top10]" "tests/test_gpu_configs.py::test_config_scale_parity[and_maybe_2x2_top10]" "tests/test_gpu_configs.py::test_config_scale_parity[filter_2x1_top10]" -m gpu -q > gpurun_out/${tag}_pytest.log 2>&1; tail -12 gpurun_out/${tag}_pytest.log | cut -c1-250
for a in "AND_NOT --terms 4 --required 2" "AND_MAYBE --terms 4 --required 2"; do
  n=$(echo $a | cut -d" " -f1 | tr A-Z a-z)
  timeout 300 python bench.py --op $a --steps 20 --warmup 2 --no-cpu-baseline --threads 0 > gpurun_out/${tag}_bench_$n.json 2>gpurun_out/${tag}_$n.err; tail -2 gpurun_out/${tag}_$n.err
done
python - <<PY
import json
for n in ('and_not','and_maybe'):
    try:
        d=json.load(open('gpurun_out/${tag}_bench_%s.json'%n)); r=d['roofline']
        print(n,round(d['value']),r['kernel'],r['kernel_ms'],d['p50_latency_us'],'frac',r['frac'],r['basis'],'alg',r['algorithmic']['frac'],r['model_counts'])
    except Exception as e: print(n,'failed',e)
PY
