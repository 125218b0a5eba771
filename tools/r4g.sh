# round 4, call g: the dispatcher's batches (unit floor, copies in stream order, flights) + the occupancy profile of the C2 launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=r04g
S="python bench.py --no-cpu-baseline --no-other-configs --no-latency --threads 64 --steps 3"
timeout 200 $S > gpurun_out/${tag}_srv_default.json 2>/dev/null
XGM_UNITS_FLOOR=1024 timeout 200 $S > gpurun_out/${tag}_srv_floor1024.json 2>/dev/null
XGM_UNITS_FLOOR=2048 timeout 200 $S > gpurun_out/${tag}_srv_floor2048.json 2>/dev/null
XGM_UNITS_FLOOR=1024 XGM_UNITS_PER_QUERY=96 timeout 200 $S > gpurun_out/${tag}_srv_floor1024_upq96.json 2>/dev/null
XGM_BATCHER_COPY_STREAMS=1 timeout 200 $S > gpurun_out/${tag}_srv_copystreams.json 2>/dev/null
XGM_BATCHER_FLIGHTS=2 timeout 200 $S > gpurun_out/${tag}_srv_flights2.json 2>/dev/null
XGM_BATCHER_FLIGHTS=2 XGM_UNITS_FLOOR=1024 timeout 200 $S > gpurun_out/${tag}_srv_flights2_floor1024.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/${tag}_srv_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d['value']), json.dumps(d['server_mode']['batched']))
    except Exception as e: print(f,'failed',e)
PY
timeout 300 python tools/units.py --no-other-configs --no-latency --threads 0 2>/dev/null | grep UNITS
XGM_TARGET_UNITS=16384 timeout 300 python tools/units.py --no-other-configs --no-latency --threads 0 2>/dev/null | grep UNITS
XGM_TARGET_UNITS=24576 timeout 300 python tools/units.py --no-other-configs --no-latency --threads 0 2>/dev/null | grep UNITS
for u in 16384 24576; do XGM_TARGET_UNITS=$u timeout 200 python bench.py --no-cpu-baseline --no-other-configs --no-latency --threads 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('target_units $u', round(d['value']), d['ms_per_batch'], d['roofline']['kernel_ms'])"; done
