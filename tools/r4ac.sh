cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # op terms topk env...
  op=$1; terms=$2; topk=$3; shift 3
  env "$@" timeout 200 python bench.py --no-cpu-baseline --no-other-configs --no-latency --threads 0 --op $op --terms $terms --topk $topk --steps 6 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$op $terms [$*]', round(d['value']), round(r['kernel_ms'],4), 'host/batch', d['host_ms_per_batch'])"
}
run PHRASE 3 10 X=1
run AND 3 10 X=1
timeout 600 python -m pytest tests/test_gpu_positional.py tests/test_gpu_parity.py -q -m gpu 2>&1 | tail -3
