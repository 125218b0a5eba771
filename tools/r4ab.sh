cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in 8 24 48 96; do for fl in 1 2; do
XGM_BENCH_BATCH=$b timeout 200 python bench.py --no-cpu-baseline --no-other-configs --no-latency --threads 0 --steps 6 --warmup 2 --in-flight $fl --batches-per-step 64 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('batch $b in-flight $fl:', round(d['value']), 'q/s  ms/batch', round(d['ms_per_batch'],4), 'kernel_ms', round(r['kernel_ms'],4), 'host/batch', d['host_ms_per_batch'])"
done; done
