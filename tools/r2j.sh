# Round-2 run J: pipelined all-dense conjunction path: parity subset + the conjunction-class bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_mixed.py -x -q -m gpu > gpurun_out/${tag}_pytest.log 2>&1
tail -3 gpurun_out/${tag}_pytest.log
run() { name=$1; shift
  timeout 300 python bench.py "$@" --steps 30 --warmup 3 --no-cpu-baseline --no-latency --threads 0 > gpurun_out/${tag}_$name.json 2>gpurun_out/${tag}_$name.err
  python -c "import json;d=json.load(open('gpurun_out/${tag}_$name.json'));print('$name',round(d['value']),d['roofline']['kernel_ms'],d['roofline'].get('frac'),d.get('parity_checked_queries'))"; }
run and3
run and2 --op AND --terms 2
run andnot --op AND_NOT --terms 4 --required 2
run andmaybe --op AND_MAYBE --terms 4 --required 2
run filter --op FILTER --terms 3 --required 2
run phrase --op PHRASE --topk 10
