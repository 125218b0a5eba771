for rep in 1 2; do
for v in "" "XGM_ORDER_BY_BODY=1"; do
  env $v timeout 300 python bench.py --steps 20 --warmup 3 --no-other-configs --no-hook-parity --ref-docs 0 --no-cpu-baseline --threads 0 --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('C2 [$v]', 'qps', round(d['value']), 'kernel_ms', round(r['kernel_ms'],4))"
  env $v timeout 300 python bench.py --op PHRASE --topk 10 --steps 10 --warmup 2 --no-other-configs --no-hook-parity --ref-docs 0 --no-cpu-baseline --threads 0 --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('C5i [$v]', 'qps', round(d['value']), 'kernel_ms', round(r['kernel_ms'],4))"
done; done
timeout 1700 python -m pytest tests/ -x -q -m gpu > gpurun_out/r6_full_pytest.log 2>&1
tail -3 gpurun_out/r6_full_pytest.log | cut -c1-300
