cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P="python bench.py --no-cpu-baseline --no-other-configs --no-latency --threads 0 --op PHRASE --topk 10 --steps 6 --warmup 1"
for e in "XGM_TARGET_UNITS=12288" "XGM_TARGET_UNITS=24576" "XGM_TARGET_UNITS=49152" "XGM_PHRASE_UNIT_STRIPES=1" "XGM_PHRASE_UNIT_STRIPES=2" "XGM_TARGET_UNITS=24576 XGM_PHRASE_CAND_COST=1.0" "XGM_TARGET_UNITS=49152 XGM_PHRASE_CAND_COST=1.0"; do env $e timeout 200 $P 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('PHRASE [$e]', round(d['value']), round(r['kernel_ms'],4), 'host/batch', d['host_ms_per_batch'])"; done
