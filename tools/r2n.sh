cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
XGM_LIB_PATH=$GRAFT_REPO_ROOT/xapiand_amd/csrc/ab/libxgm_timers.so timeout 300 python tools/phase.py --op OR --terms 5 --topk 100 --no-latency --threads 0 2>&1 | grep "ORW PHASES" | tee gpurun_out/${tag}_phases.txt
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/${tag}_pytest_gpu.log 2>&1; tail -5 gpurun_out/${tag}_pytest_gpu.log
run() { name=$1; shift
  timeout 300 python bench.py "$@" --steps 20 --warmup 3 --no-cpu-baseline --no-latency --threads 0 > gpurun_out/${tag}_$name.json 2>gpurun_out/${tag}_$name.err
  python -c "import json;d=json.load(open('gpurun_out/${tag}_$name.json'));r=d['roofline'];print('$name',round(d['value']),r['kernel_ms'],r.get('model_frac'))" || tail -3 gpurun_out/${tag}_$name.err; }
run and3
run phrase --op PHRASE --topk 10
run or2 --op OR --terms 2 --topk 10
run or8 --op OR --terms 8 --topk 100
