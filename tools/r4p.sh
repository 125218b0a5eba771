cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # op terms topk env...
  op=$1; terms=$2; topk=$3; shift 3
  env "$@" timeout 200 python bench.py --no-cpu-baseline --no-other-configs --no-latency --threads 0 --op $op --terms $terms --topk $topk --steps 6 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$op [$*]', round(d['value']), round(r['kernel_ms'],4), 'host/batch', d['host_ms_per_batch'])"
}
for e in "X=1" "XGM_UNITS_PER_QUERY=64 XGM_TARGET_UNITS=16384" "XGM_UNITS_PER_QUERY=80 XGM_TARGET_UNITS=20480" "XGM_UNITS_PER_QUERY=96 XGM_TARGET_UNITS=24576" "XGM_UNITS_PER_QUERY=128 XGM_TARGET_UNITS=32768"; do run PHRASE 3 10 $e; done
for e in "X=1" "XGM_UNITS_PER_QUERY=64 XGM_TARGET_UNITS=16384" "XGM_UNITS_PER_QUERY=96 XGM_TARGET_UNITS=24576"; do run AND 3 10 $e; done
