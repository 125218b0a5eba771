# Round-2 evidence run A: GPU tests (incl. config-scale parity + multi-rank), C2 bench with the reference leg, counter calibration,
# PMC passes for the OR and PHRASE kernels, C3 / C5 / sided bench lines with parity
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
df -h /tmp . | tail -2 > gpurun_out/${tag}_df.txt; nproc >> gpurun_out/${tag}_df.txt; free -g | head -2 >> gpurun_out/${tag}_df.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/${tag}_pytest.log 2>&1; tail -3 gpurun_out/${tag}_pytest.log
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 600 gpurun_out/${tag}_bench.err; python - <<PY
import json
try:
    d=json.load(open('gpurun_out/${tag}_bench.json')); r=d['roofline']; c=d.get('cpu_baseline',{})
    print('AND3',round(d['value']),d['ms_per_step'],r['kernel_ms'],d['p50_latency_us'],'plan_us',d['plan_us_per_query'],'frac',r['frac'],r['basis'],'model',r['model_frac'],'alg',r['algorithmic']['frac'])
    print('cpu',c.get('kind'),c.get('value'),c.get('all_cores'),c.get('port_over_reference'),c.get('index_build'))
except Exception as e: print('bench parse failed',e)
PY
bash tools/calib.sh $tag > gpurun_out/${tag}_calib.out 2>&1; tail -9 gpurun_out/${tag}_calib.out
