cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_variants.py -m gpu -q -x -k "random_queries or edge_cases or other_stripe or OR or or5 or c3 or disj or SEED or PRUNE or PHASE_A or BOUND_SUM or DENSE" 2>&1 | tail -3
run() { name=$1; shift
  timeout 300 python bench.py "$@" --steps 20 --warmup 3 --no-cpu-baseline --no-latency --threads 0 > gpurun_out/${tag}_$name.json 2>gpurun_out/${tag}_$name.err
  python -c "import json;d=json.load(open('gpurun_out/${tag}_$name.json'));r=d['roofline'];print('$name',round(d['value']),r['kernel_ms'],'weighed',r['model_counts']['doclen_gathers'],'model_frac',r['model_frac'])" || tail -3 gpurun_out/${tag}_$name.err; }
run or5 --op OR --terms 5 --topk 100
run or3 --op OR --terms 3 --topk 10
XGM_LIB_PATH=$GRAFT_REPO_ROOT/xapiand_amd/csrc/ab/libxgm_timers.so timeout 300 python tools/phase.py --op OR --terms 5 --topk 100 --no-latency --threads 0 2>&1 | grep "ORW PHASES"
