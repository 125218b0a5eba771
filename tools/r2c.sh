# Round-2 run C: B1 hook + mixed-batch tests, AND-2 comparison for the sided operators, reference-index build timing in /dev/shm
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout 900 python -m pytest tests/test_gpu_hook_b1.py tests/test_gpu_hook.py tests/test_gpu_mixed.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/${tag}_pytest.log 2>&1; tail -15 gpurun_out/${tag}_pytest.log
timeout 300 python bench.py --op AND --terms 2 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench_and2.json 2>gpurun_out/${tag}_and2.err; tail -2 gpurun_out/${tag}_and2.err
python - <<PY
import json
d=json.load(open('gpurun_out/${tag}_bench_and2.json')); r=d['roofline']
print('and2',round(d['value']),r['kernel'],r['kernel_ms'],d['p50_latency_us'],'frac',r['frac'],r['basis'],'alg',r['algorithmic']['frac'],r['model_counts'])
PY
df -h /dev/shm | tail -1
( time python tools/ref_index.py /dev/shm/xgm_ref_t 10000000 --nopos --procs 128 ) > gpurun_out/${tag}_refidx.txt 2>&1; tail -5 gpurun_out/${tag}_refidx.txt; rm -rf /dev/shm/xgm_ref_t*
