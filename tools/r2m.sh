# disjunction kernel v2 (seeded threshold + wdf>=2 planes): tests, C3 with parity, A/B of the group size and of the seed, other classes for reference
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/${tag}_pytest_gpu.log 2>&1; tail -5 gpurun_out/${tag}_pytest_gpu.log
timeout 400 python bench.py --op OR --terms 5 --topk 100 --steps 20 --warmup 2 --ref-docs 0 --cpu-seconds 4 --threads 0 > gpurun_out/${tag}_bench_or5.json 2>gpurun_out/${tag}_or5.err
python -c "
import json;d=json.load(open('gpurun_out/${tag}_bench_or5.json'));r=d['roofline'];print('C3 parity run',round(d['value']),r['kernel_ms'],'frac',r.get('frac'),'model',r.get('model_frac'),'alg',r['algorithmic']['frac'],'parity',d['cpu_baseline'].get('parity_checked_queries'),'p50',d.get('p50_latency_us'))" || tail -5 gpurun_out/${tag}_or5.err
run() { name=$1; shift
  timeout 300 python bench.py "$@" --steps 20 --warmup 3 --no-cpu-baseline --no-latency --threads 0 > gpurun_out/${tag}_$name.json 2>gpurun_out/${tag}_$name.err
  python -c "import json;d=json.load(open('gpurun_out/${tag}_$name.json'));print('$name',round(d['value']),d['roofline']['kernel_ms'],d['roofline'].get('frac'),d['roofline'].get('model_frac'))" || tail -3 gpurun_out/${tag}_$name.err; }
run or5_g5 --op OR --terms 5 --topk 100
XGM_LIB_PATH=$GRAFT_REPO_ROOT/xapiand_amd/csrc/ab/libxgm_g3.so run or5_g3 --op OR --terms 5 --topk 100
XGM_LIB_PATH=$GRAFT_REPO_ROOT/xapiand_amd/csrc/ab/libxgm_g2.so run or5_g2 --op OR --terms 5 --topk 100
XGM_OR_SEED_SCALE=0 run or5_noseed --op OR --terms 5 --topk 100
XGM_OR_SEED_SCALE=0.8 run or5_seed08 --op OR --terms 5 --topk 100
XGM_OR_SEED_SCALE=1.15 run or5_seed115 --op OR --terms 5 --topk 100
run or5_top10 --op OR --terms 5 --topk 10
run or3_top10 --op OR --terms 3 --topk 10
run phrase --op PHRASE --topk 10
run andnot --op AND_NOT --terms 4 --required 2
run andmaybe --op AND_MAYBE --terms 4 --required 2
run and3
