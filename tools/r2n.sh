cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_mixed.py -m gpu -q -x 2>&1 | tail -1
run() { name=$1; shift
  timeout 300 python bench.py "$@" --steps 20 --warmup 3 --no-cpu-baseline --no-latency --threads 0 > gpurun_out/${tag}_$name.json 2>gpurun_out/${tag}_$name.err
  python -c "import json;d=json.load(open('gpurun_out/${tag}_$name.json'));r=d['roofline'];print('$name',round(d['value']),'ms/step',round(d['ms_per_step'],3),'kernel',round(r['kernel_ms'],3))" || tail -3 gpurun_out/${tag}_$name.err; }
run and3
run phrase --op PHRASE --topk 10
run or5 --op OR --terms 5 --topk 100
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python bench.py --steps 20 --no-cpu-baseline --no-latency --threads 0 > gpurun_out/${tag}_prof.log 2>&1
grep "xgm_merge_kernel\|xgm_andw_kernel" gpurun_out/${tag}_prof/*/*kernel_stats.csv | awk -F'",' '{print substr($1,1,90), $2}' | cut -c1-200
