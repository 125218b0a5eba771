# round 4, call d: the whole GPU suite + the default bench line with small reference indexes (a dry run of every leg)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=r04d
timeout 1200 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/${tag}_pytest_gpu.log 2>&1; tail -12 gpurun_out/${tag}_pytest_gpu.log
timeout 900 python bench.py --ref-docs 400000 --hook-pos-docs 200000 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 1500 gpurun_out/${tag}_bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/${tag}_bench.json')); r=d['roofline']
print('C2', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'kernel_ms', r['kernel_ms'], 'p50', d['p50_latency_us'], 'frac', r['frac'], r['basis'], 'verified', d['last_batch_on_host_equals_synchronous_search'])
for k,v in (d.get('other_configs') or {}).items():
    if 'error' in v: print(k, 'ERROR', v['error']); continue
    print(k, round(v['value']), 'ms/step', round(v['ms_per_step'],3), 'kernel_ms', v['roofline']['kernel_ms'], 'frac', v['roofline']['frac'], v['roofline']['basis'], 'parity', v['parity_checked_queries'], 'p50', v['p50_latency_us'], 'verified', v['last_batch_on_host_equals_synchronous_search'], 'cpu', v['cpu_baseline'] and v['cpu_baseline']['value'])
print('server', json.dumps(d.get('server_mode')))
c=d.get('cpu_baseline',{}); print('cpu', {k:v for k,v in c.items() if k not in ('sample','port_full_size','index_build')})
print('hook_parity', json.dumps(d.get('hook_parity'))[:3000])
PY
