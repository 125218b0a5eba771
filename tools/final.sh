# Round-end evidence run: tests, headline bench (+cpu baseline), rocprofv3 stats, HBM traffic PMC, C3/C5 side numbers
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest.log 2>&1; tail -1 gpurun_out/${tag}_pytest.log
# HBM traffic first (bench.py reads profiles/traffic.json when present; this run refreshes the numbers)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "xgm_andw" --output-format csv -d gpurun_out/${tag}_pmc_$c -- python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-latency > gpurun_out/${tag}_pmc_$c.log 2>&1
done
python tools/pmc_parse.py gpurun_out/${tag}_pmc_FETCH_SIZE gpurun_out/${tag}_pmc_WRITE_SIZE | tee gpurun_out/${tag}_pmc.txt
timeout 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; python -c "import json;d=json.load(open('gpurun_out/${tag}_bench.json'));print('AND3',round(d['value']),d['ms_per_step'],d['roofline']['kernel_ms'],d['p50_latency_us'],d['roofline']['frac'],d['cpu_baseline']['value'])"
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python bench.py --steps 20 --no-cpu-baseline --no-latency > gpurun_out/${tag}_prof.log 2>&1
timeout 300 python bench.py --op OR --terms 5 --topk 100 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench_or5.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/${tag}_bench_or5.json'));print('OR5',round(d['value']),d['roofline']['kernel_ms'],d['p50_latency_us'],d['roofline']['frac'])"
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_or5 -- python bench.py --op OR --terms 5 --topk 100 --steps 10 --no-cpu-baseline --no-latency > gpurun_out/${tag}_prof_or5.log 2>&1
timeout 300 python bench.py --op PHRASE --topk 10 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench_phrase.json 2>gpurun_out/${tag}_phrase.err; python -c "import json;d=json.load(open('gpurun_out/${tag}_bench_phrase.json'));print('PHRASE',round(d['value']),d['roofline']['kernel_ms'],d['p50_latency_us'],d['roofline']['frac'])"; tail -2 gpurun_out/${tag}_phrase.err
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_phrase -- python bench.py --op PHRASE --topk 10 --steps 6 --no-cpu-baseline --no-latency > gpurun_out/${tag}_prof_phrase.log 2>&1
for a in "AND_NOT --terms 4 --required 2" "AND_MAYBE --terms 4 --required 2" "FILTER --terms 3 --required 2"; do
  n=$(echo $a | cut -d" " -f1 | tr A-Z a-z)
  timeout 300 python bench.py --op $a --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench_$n.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/${tag}_bench_$n.json'));print('$n',round(d['value']),d['roofline']['kernel'],d['roofline']['kernel_ms'],d['p50_latency_us'],d['roofline']['frac'])"
done
find gpurun_out/${tag}_prof* -name "*kernel_stats.csv" | head
