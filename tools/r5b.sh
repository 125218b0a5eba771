# round 5, call B: full GPU suite on the new build; 10 M-doc bench without the CPU legs (C2 + C3 / C5 incl. the byte-compatible-mode legs);
# the disjunction kernel's section timers; the whole new bench flow (reference index with positions, fixed timing, hook legs) at 1 M documents
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/r5b_pytest_gpu.log 2>&1; tail -12 gpurun_out/r5b_pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline --threads 0 --no-latency --steps 5 > gpurun_out/r5b_bench_nocpu.json 2> gpurun_out/r5b_bench_nocpu.err; tail -c 400 gpurun_out/r5b_bench_nocpu.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r5b_bench_nocpu.json'))
    print('C2', round(d['value']), d['roofline']['kernel_ms'])
    for n,o in d['other_configs'].items():
        if 'error' in o: print(n, o); continue
        print(n, round(o['value']), o['roofline']['kernel_ms'], 'timed rows', o.get('timed_batch_rows_checked_against_oracle'))
        print('   compat:', json.dumps(o.get('reference_identical_mode') or o.get('exact_bounds_mode'))[:700])
except Exception as e: print('bench nocpu failed', e)
PY
XGM_LIB_PATH=$PWD/xapiand_amd/csrc/ab/libxgm_ortim.so timeout 300 python tools/phase.py --op OR --terms 5 --topk 100 --no-other-configs --threads 0 --no-latency 2>&1 | grep "ORW PHASES" > gpurun_out/r5b_orw_phases.txt; cat gpurun_out/r5b_orw_phases.txt
timeout 900 python bench.py --docs-per-gpu 1000000 --steps 3 --threads 0 --no-latency --ref-seconds 4 --cpu-seconds 3 > gpurun_out/r5b_bench_1m.json 2> gpurun_out/r5b_bench_1m.err; tail -c 600 gpurun_out/r5b_bench_1m.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r5b_bench_1m.json'))
    c=d['cpu_baseline']; print('cpu_baseline', {k:c[k] for k in c if k in ('kind','value','cores','p50_ms','all_cores','cpu_model','docs','port_over_reference','parity_checked_queries','timed_batch_rows_checked_against_oracle')})
    print('sample:', c.get('sample','')[:400])
    for n,o in d['other_configs'].items(): print(n, json.dumps(o.get('cpu_baseline'))[:900])
    print('hook_parity', json.dumps(d.get('hook_parity'))[:2500])
except Exception as e: print('bench 1m failed', e)
PY
