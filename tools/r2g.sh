# Round-2 run G: nested trees on device + hook; sided bench after the cost-model fix
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout 1500 python -m pytest tests/test_gpu_trees.py tests/test_gpu_parity.py tests/test_gpu_hook_b1.py tests/test_gpu_mixed.py tests/test_gpu_builder.py -m gpu -q > gpurun_out/${tag}_pytest.log 2>&1; tail -30 gpurun_out/${tag}_pytest.log | cut -c1-300
for a in "AND_NOT --terms 4 --required 2" "AND_MAYBE --terms 4 --required 2"; do
  n=$(echo $a | cut -d" " -f1 | tr A-Z a-z)
  timeout 300 python bench.py --op $a --steps 20 --warmup 2 --no-cpu-baseline --threads 0 > gpurun_out/${tag}_bench_$n.json 2>gpurun_out/${tag}_$n.err; tail -2 gpurun_out/${tag}_$n.err
done
python - <<PY
import json
for n in ('and_not','and_maybe'):
    try:
        d=json.load(open('gpurun_out/${tag}_bench_%s.json'%n)); r=d['roofline']
        print(n,round(d['value']),r['kernel'],r['kernel_ms'],d['p50_latency_us'],'frac',r['frac'],r['basis'],'alg',r['algorithmic']['frac'])
    except Exception as e: print(n,'failed',e)
PY
