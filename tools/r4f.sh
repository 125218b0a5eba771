# round 4, call f: flat-led conjunctions + the fast positional test of the dense body: parity at 10 M documents, then what they buy
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=r04f
timeout 900 python -m pytest tests/test_gpu_flat.py tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_positional.py tests/test_gpu_stress.py tests/test_gpu_variants.py -m gpu -q --durations=5 -k "flat or C2 or C5 or and_maybe or and_not or filter or random or batch_equals or positional or stress or XGM_NO_FLAT or XGM_NO_DENSE_PHRASE" > gpurun_out/${tag}_pytest.log 2>&1; tail -8 gpurun_out/${tag}_pytest.log
B="python bench.py --no-cpu-baseline --no-other-configs --no-latency --threads 0"
timeout 200 $B > gpurun_out/${tag}_c2.json 2>gpurun_out/${tag}_c2.err
XGM_NO_FLAT=1 timeout 200 $B > gpurun_out/${tag}_c2_noflat.json 2>/dev/null
P="$B --op PHRASE --topk 10 --steps 6 --warmup 1"
timeout 200 $P > gpurun_out/${tag}_phrase.json 2>/dev/null
XGM_PHRASE_UNIT_STRIPES=0 timeout 200 $P > gpurun_out/${tag}_phrase_units0.json 2>/dev/null
XGM_PHRASE_UNIT_STRIPES=4 timeout 200 $P > gpurun_out/${tag}_phrase_units4.json 2>/dev/null
for a in "AND --terms 2" "FILTER --terms 3 --required 2"; do n=$(echo $a | tr -d '-' | tr ' ' '_' | tr A-Z a-z); timeout 200 $B --op $a --steps 6 > gpurun_out/${tag}_$n.json 2>/dev/null; done
timeout 200 python bench.py --no-cpu-baseline --no-other-configs --no-latency --threads 64 --steps 4 > gpurun_out/${tag}_c2_server.json 2>/dev/null
XGM_UNITS_PER_QUERY=0 timeout 200 python bench.py --no-cpu-baseline --no-other-configs --no-latency --threads 64 --steps 4 > gpurun_out/${tag}_c2_server_units0.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/${tag}_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], round(d['value']), 'ms/batch', round(d['ms_per_batch'],4), r['kernel'], 'kernel_ms', round(r['kernel_ms'],4), 'host/batch', d['host_ms_per_batch'], 'verified', d['last_batch_on_host_equals_synchronous_search'], 'model GB', r['model_min_bytes'] and round(r['model_min_bytes']/1e9,3), 'index GB', round(d['index']['device_bytes']/1e9,2), 'build s', round(d['index']['build_seconds'],2))
        if d.get('server_mode'): print('   server', json.dumps(d['server_mode']['batched']))
    except Exception as e: print(f,'failed',e)
PY
tail -c 300 gpurun_out/${tag}_c2.err
timeout 300 python tools/qcost.py 2>/dev/null | grep QCOST | head -14
timeout 300 python tools/qcost.py --op PHRASE --topk 10 2>/dev/null | grep QCOST | head -14
