cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
run() { name=$1; shift
  timeout 300 python bench.py "$@" --steps 30 --warmup 3 --no-cpu-baseline --no-latency --threads 0 > gpurun_out/${tag}_$name.json 2>gpurun_out/${tag}_$name.err
  python -c "import json;d=json.load(open('gpurun_out/${tag}_$name.json'));print('$name',round(d['value']),d['roofline']['kernel_ms'],d['roofline'].get('frac'),d.get('parity_checked_queries'))"; }
run and3
run and2 --op AND --terms 2
run andnot --op AND_NOT --terms 4 --required 2
run andmaybe --op AND_MAYBE --terms 4 --required 2
XGM_UNITS_TAG=and3 python tools/units.py --no-latency --threads 0 2>&1 | grep "UNITS n"
XGM_UNITS_TAG=and2 python tools/units.py --no-latency --threads 0 --op AND --terms 2 2>&1 | grep "UNITS n"
