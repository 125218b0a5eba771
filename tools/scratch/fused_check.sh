cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
b() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --threads 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('qps', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(r['kernel_ms'],4), 'p50', round(d.get('p50_latency_us') or 0,1), 'p99', round(d.get('p99_latency_us') or 0,1))"; }
echo "C2 fused:"; b
echo "C2 merge launch:"; XGM_NO_FUSED_MERGE=1 b
echo "C2 fused:"; b
echo "C5 fused:"; b --op PHRASE --topk 10 --steps 10
echo "C5 merge launch:"; XGM_NO_FUSED_MERGE=1 b --op PHRASE --topk 10 --steps 10
echo "AND-2 fused:"; b --op AND --terms 2
timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_positional.py tests/test_gpu_mixed.py tests/test_gpu_trees.py -m gpu -q -x -k "not C3_or5" 2>&1 | tail -4
