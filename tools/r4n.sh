cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=r04n
timeout 900 python -m pytest tests/test_gpu_flat.py tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_positional.py tests/test_gpu_mixed.py tests/test_gpu_variants.py tests/test_gpu_stress.py -m gpu -q --durations=5 -k "flat or C2 or C5 or phrase or positional or mixed or random or stress or XGM_NO_FLAT or XGM_NO_DENSE_PHRASE or XGM_NO_PHRASEW or XGM_NO_FUSED" > gpurun_out/${tag}_pytest.log 2>&1; tail -6 gpurun_out/${tag}_pytest.log
B="python bench.py --no-cpu-baseline --no-other-configs --no-latency --threads 0"
P="$B --op PHRASE --topk 10 --steps 6 --warmup 1"
for e in "" "XGM_NO_FLAT_PHRASE=1"; do env $e timeout 200 $P 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('PHRASE [$e]', round(d['value']), round(r['kernel_ms'],4), 'verified', d['last_batch_on_host_equals_synchronous_search'], 'index GB', round(d['index']['device_bytes']/1e9,2))"; done
timeout 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('C2', round(d['value']), round(r['kernel_ms'],4), 'index GB', round(d['index']['device_bytes']/1e9,2), 'build', round(d['index']['build_seconds'],2))"
timeout 300 python tools/qcost.py --op PHRASE --topk 10 2>/dev/null | grep QCOST | head -8
