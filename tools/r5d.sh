# round 5, call D: xgm_orw2_kernel as ONE class (flat-array instantiation for every eligible query), unit-count sweep, parity, the byte-compatible legs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OR="--no-other-configs --op OR --terms 5 --topk 100 --steps 8"
{
XGM_NO_ORW2=1 bash tools/ab_run.sh "$OR" default | sed 's/^default/old-kernel/'
bash tools/ab_run.sh "$OR" default
XGM_ORW_UNITS=12288 bash tools/ab_run.sh "$OR" default | sed 's/^default/units12288/'
XGM_ORW_UNITS=6144 bash tools/ab_run.sh "$OR" default | sed 's/^default/units6144/'
XGM_ORW_UNITS=16384 bash tools/ab_run.sh "$OR" default | sed 's/^default/units16384/'
bash tools/ab_run.sh "$OR" default
} > gpurun_out/r5d_ab.txt 2>&1
cat gpurun_out/r5d_ab.txt
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_all.py -m gpu -x -q -k "not hook" > gpurun_out/r5d_pytest.log 2>&1; tail -4 gpurun_out/r5d_pytest.log
timeout 600 python bench.py --no-cpu-baseline --threads 0 --no-latency --steps 5 > gpurun_out/r5d_bench_nocpu.json 2> gpurun_out/r5d_bench_nocpu.err; tail -c 300 gpurun_out/r5d_bench_nocpu.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r5d_bench_nocpu.json'))
    print('C2', round(d['value']), d['roofline']['kernel_ms'])
    for n,o in d['other_configs'].items():
        if 'error' in o: print(n, o); continue
        print(n, round(o['value']), o['roofline']['kernel'], o['roofline']['kernel_ms'], 'timed rows', o.get('timed_batch_rows_checked_against_oracle'), 'model', o['roofline'].get('model_min_bytes'))
        print('   compat:', json.dumps(o.get('reference_identical_mode') or o.get('exact_bounds_mode'))[:400])
except Exception as e: print('bench nocpu failed', e)
PY
