timeout 900 python bench.py --no-hook-parity --ref-docs 0 --no-cpu-baseline > gpurun_out/r6_sub_lat2.json 2>gpurun_out/r6_sub.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_sub_lat2.json').read().strip().splitlines()[-1])
print(round(d['value']), d['host_ms_per_batch'], 'count', round(d['exact_bounds_mode'].get('value',0)))
for n in ('C3','C5'):
    o=d['other_configs'][n]
    print('  ',n, round(o['value']), o['ms_per_batch'], o.get('host_ms_per_batch'), o['roofline']['kernel_ms'])
PY
