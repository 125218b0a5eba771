# round 4, call b: the new tests alone + the timed loop's delivery modes (no CPU legs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=r04b
timeout 900 python -m pytest tests/test_gpu_all.py tests/test_gpu_stress.py tests/test_gpu_mixed.py "tests/test_gpu_hook_b1.py" -m gpu -q -x --durations=5 -k "all_matches or stress or mixed or server_mode or exact_match_count or beyond or positional_reference or single_shard" > gpurun_out/${tag}_pytest.log 2>&1; tail -5 gpurun_out/${tag}_pytest.log
B="python bench.py --no-cpu-baseline --no-other-configs --no-latency"
timeout 300 $B --threads 64 > gpurun_out/${tag}_bench_streams3.json 2> gpurun_out/${tag}_bench_streams3.err
timeout 200 $B --threads 0 --in-flight 2 > gpurun_out/${tag}_bench_streams2.json 2>/dev/null
timeout 200 $B --threads 0 --in-flight 1 > gpurun_out/${tag}_bench_streams1.json 2>/dev/null
XGM_BENCH_ONE_STREAM=1 timeout 200 $B --threads 0 > gpurun_out/${tag}_bench_one_copystream.json 2>/dev/null
XGM_BENCH_ONE_STREAM=1 XGM_COPY_ON_BATCH_STREAM=1 timeout 200 $B --threads 0 > gpurun_out/${tag}_bench_one_inorder.json 2>/dev/null
XGM_BATCHER_FLIGHTS=1 timeout 300 $B --threads 64 --steps 5 > gpurun_out/${tag}_bench_flights1.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/${tag}_bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], round(d['value']), 'ms/batch', round(d['ms_per_batch'],4), 'kernel_ms', round(r['kernel_ms'],4), 'solo', r['kernel_ms_one_in_flight'] and round(r['kernel_ms_one_in_flight'],4), 'host/batch', d['host_ms_per_batch'], 'verified', d['last_batch_on_host_equals_synchronous_search'])
        if d.get('server_mode'): print('   server', json.dumps(d['server_mode']))
    except Exception as e: print(f,'failed',e)
PY
tail -c 600 gpurun_out/${tag}_bench_streams3.err
