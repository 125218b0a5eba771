# Round evidence, part A (the essentials first): full GPU test suite, headline bench with the CPU baseline legs, kernel stats of C2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/${tag}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${tag}_pytest_gpu.log
[ -n "$XGM_FINAL_TRIM" ] && exit 0        # short of GPU minutes: the headline line comes from tools/final.sh alone
timeout 1700 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 300 gpurun_out/${tag}_bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/${tag}_bench.json')); r=d['roofline']; c=d.get('cpu_baseline',{})
print('C2',round(d['value']),'kernel_ms',r['kernel_ms'],'p50',d['p50_latency_us'],'frac',r['frac'],r['basis'],'model',r['model_frac'],'parity',c.get('parity_checked_queries'),c.get('kind'),c.get('value'))
print(json.dumps(d.get('server_mode')))
PY
