# round 4, call c: hook / stress tests, the positional class on the dense body (A/B), C2 with the copy stream, the batcher's flights
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=r04c
timeout 900 python -m pytest tests/test_gpu_stress.py tests/test_gpu_mixed.py tests/test_gpu_hook_b1.py tests/test_gpu_positional.py tests/test_gpu_configs.py -m gpu -q --durations=5 -k "stress or mixed or server_mode or exact_match_count or beyond or positional or single_shard or C5" > gpurun_out/${tag}_pytest.log 2>&1; tail -8 gpurun_out/${tag}_pytest.log
B="python bench.py --no-cpu-baseline --no-other-configs --no-latency --threads 0"
P="$B --op PHRASE --topk 10 --steps 6 --warmup 1"
timeout 200 $P > gpurun_out/${tag}_phrase_default.json 2>gpurun_out/${tag}_phrase_default.err
XGM_NO_DENSE_PHRASE_BODY=1 timeout 200 $P > gpurun_out/${tag}_phrase_old.json 2>/dev/null
XGM_DENSE_KERNEL=1 timeout 200 $P > gpurun_out/${tag}_phrase_alone6.json 2>/dev/null
XGM_DENSE_KERNEL=1 XGM_LIB_PATH=$PWD/xapiand_amd/csrc/ab/libxgm_dpw4.so timeout 200 $P > gpurun_out/${tag}_phrase_alone4.json 2>/dev/null
timeout 200 $B > gpurun_out/${tag}_c2_default.json 2>/dev/null
for f in 1 2 3; do XGM_BATCHER_FLIGHTS=$f timeout 200 python bench.py --no-cpu-baseline --no-other-configs --no-latency --threads 64 --steps 4 > gpurun_out/${tag}_c2_flights$f.json 2>/dev/null; done
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/${tag}_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], round(d['value']), 'ms/batch', round(d['ms_per_batch'],4), r['kernel'], 'kernel_ms', round(r['kernel_ms'],4), 'solo', r['kernel_ms_one_in_flight'] and round(r['kernel_ms_one_in_flight'],4), 'host/batch', d['host_ms_per_batch'], 'verified', d['last_batch_on_host_equals_synchronous_search'], 'launches/batch', d['host_us_per_batch']['launches_per_batch'])
        if d.get('server_mode'): print('   server', json.dumps(d['server_mode']['batched']))
    except Exception as e: print(f,'failed',e)
PY
tail -c 400 gpurun_out/${tag}_phrase_default.err
bash tools/trace.sh --no-other-configs --threads 0 2>&1 | tail -22
