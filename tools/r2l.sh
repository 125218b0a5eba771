# HEAD check after the session restart: GPU tests, headline bench, kernel stats, C3 / C5 quick lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/${tag}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${tag}_pytest_gpu.log
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 300 gpurun_out/${tag}_bench.err
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_and3 -- python bench.py --steps 20 --no-cpu-baseline --no-latency --threads 0 > gpurun_out/${tag}_prof_and3.log 2>&1
run() { name=$1; shift
  timeout 300 python bench.py "$@" --steps 20 --warmup 3 --no-cpu-baseline --no-latency --threads 0 > gpurun_out/${tag}_$name.json 2>gpurun_out/${tag}_$name.err
  python -c "import json;d=json.load(open('gpurun_out/${tag}_$name.json'));print('$name',round(d['value']),d['roofline']['kernel_ms'],d['roofline'].get('frac'))"; }
run or5 --op OR --terms 5 --topk 100
run phrase --op PHRASE --topk 10
python -c "
import json;d=json.load(open('gpurun_out/${tag}_bench.json'));r=d['roofline'];print('C2',round(d['value']),r['kernel_ms'],r['frac'],r['basis'],d['p50_latency_us'],json.dumps(d.get('server_mode')));print(json.dumps({k:v for k,v in d['cpu_baseline'].items() if k!='sample'})[:800])"
find gpurun_out/${tag}_prof_and3 -name "*kernel_stats.csv" | head -1 | xargs head -8
