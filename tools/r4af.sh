cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for fl in 3 2 4 6 3; do
timeout 200 python bench.py --no-cpu-baseline --no-other-configs --no-latency --threads 0 --steps 20 --warmup 3 --in-flight $fl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('in-flight $fl:', round(d['value']), 'ms/batch', round(d['ms_per_batch'],4), 'kernel_ms', round(r['kernel_ms'],4), 'host/batch', d['host_ms_per_batch'])"
done
