# A/B sweep of environment switches over one bench configuration: tools/sweep.sh "<bench args>" VAR=val[,VAR=val] ...   (one MI355X; prints queries/s and kernel ms)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
args=$1; shift
for v in "$@"; do
  env $(echo $v | tr ',' ' ') timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-latency --threads 0 --steps ${SWEEP_STEPS:-3} --warmup 1 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', 'qps', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'kernel_ms', round(r['kernel_ms'],4), r['kernel'])"
done
