cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
srv() {
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-latency --threads 64 --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['server_mode']['batched']; print('server [$*]', round(s['value']), 'p50', round(s['p50_us']), 'p99', round(s['p99_us']), 'mean_batch', round(s['mean_batch'],1), '| C2', round(d['value']))"
}
run() { # op terms topk env...
  op=$1; terms=$2; topk=$3; shift 3
  env "$@" timeout 200 python bench.py --no-cpu-baseline --no-other-configs --no-latency --threads 0 --op $op --terms $terms --topk $topk --steps 6 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$op $terms [$*]', round(d['value']), round(r['kernel_ms'],4), 'host/batch', d['host_ms_per_batch'])"
}
srv X=1
srv XGM_BATCHER_LINGER_US=0
srv XGM_BATCHER_LINGER_US=80
srv XGM_BATCHER_COPY_STREAMS=1
srv XGM_BATCHER_LINGER_US=40 XGM_BATCHER_FLIGHTS=3
run PHRASE 3 10 X=1
run PHRASE 3 10 XGM_PHRASE_T3_COST=2
run PHRASE 3 10 XGM_PHRASE_T3_COST=4
run PHRASE 3 10 XGM_PHRASE_CAND_COST=0.5
