cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in 0.25 0.5 1.0 2.0; do XGM_PHRASE_CAND_COST=$c python bench.py --op PHRASE --topk 10 --steps 6 --warmup 1 --no-other-configs --no-cpu-baseline --threads 0 --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cand_cost $c', round(d['value']), round(d['roofline']['kernel_ms'],4))"; done
for u in 12288 16384 24576; do XGM_TARGET_UNITS=$u python bench.py --op PHRASE --topk 10 --steps 6 --warmup 1 --no-other-configs --no-cpu-baseline --threads 0 --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('target_units $u', round(d['value']), round(d['roofline']['kernel_ms'],4))"; done
