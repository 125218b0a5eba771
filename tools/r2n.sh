cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout 900 python -m pytest tests/test_gpu_positional.py tests/test_gpu_mixed.py -m gpu -q -x 2>&1 | tail -2
run() { name=$1; shift
  timeout 300 python bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline --no-latency --threads 0 > gpurun_out/${tag}_$name.json 2>gpurun_out/${tag}_$name.err
  python -c "import json;d=json.load(open('gpurun_out/${tag}_$name.json'));r=d['roofline'];print('$name',round(d['value']),r['kernel_ms'],'positions',r['model_counts']['positions'],'dl',r['model_counts']['doclen_gathers'])" || tail -3 gpurun_out/${tag}_$name.err; }
run phrase --op PHRASE --topk 10
