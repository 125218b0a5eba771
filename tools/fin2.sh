cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/${tag}_pytest.log 2>&1; tail -1 gpurun_out/${tag}_pytest.log
timeout 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; python -c "import json;d=json.load(open('gpurun_out/${tag}_bench.json'));print('AND3',round(d['value']),d['ms_per_step'],d['roofline']['kernel_ms'],d['p50_latency_us'],d['roofline']['frac'],d['roofline']['traffic'],d['cpu_baseline']['value'])"
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python bench.py --steps 20 --no-cpu-baseline --no-latency > gpurun_out/${tag}_prof.log 2>&1
timeout 300 python bench.py --op OR --terms 5 --topk 100 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench_or5.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/${tag}_bench_or5.json'));print('OR5',round(d['value']),d['ms_per_step'],d['roofline']['kernel_ms'],d['p50_latency_us'],d['roofline']['frac'])"
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_or5 -- python bench.py --op OR --terms 5 --topk 100 --steps 10 --no-cpu-baseline --no-latency > gpurun_out/${tag}_prof_or5.log 2>&1
find gpurun_out/${tag}_prof* -name "*kernel_stats.csv" | head
