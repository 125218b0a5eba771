# Round-2 evidence run B: B1 hook tests, C2 bench with the FULL-SIZE reference leg, calibration (with the pair kernel), PMC passes
# for the OR and PHRASE kernels, C3 / C5 / sided bench lines (parity on, reference leg off)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout 900 python -m pytest tests/test_gpu_hook_b1.py tests/test_gpu_hook.py -m gpu -q > gpurun_out/${tag}_pytest_hook.log 2>&1; tail -15 gpurun_out/${tag}_pytest_hook.log
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 600 gpurun_out/${tag}_bench.err; python - <<PY
import json
try:
    d=json.load(open('gpurun_out/${tag}_bench.json')); r=d['roofline']; c=d.get('cpu_baseline',{})
    print('AND3',round(d['value']),d['ms_per_step'],r['kernel_ms'],d['p50_latency_us'],'plan_us',d['plan_us_per_query'],'frac',r['frac'],r['basis'],'model',r['model_frac'],'alg',r['algorithmic']['frac'])
    print(r['model_counts'])
    print('cpu',c.get('kind'),c.get('docs'),c.get('value'),c.get('all_cores'),c.get('port_over_reference'),c.get('index_build'))
except Exception as e: print('bench parse failed',e)
PY
bash tools/calib.sh $tag > gpurun_out/${tag}_calib.out 2>&1; tail -11 gpurun_out/${tag}_calib.out
for w in "or5 orw --op OR --terms 5 --topk 100" "phrase andw --op PHRASE --topk 10" "and3 andw"; do
  set -- $w; n=$1; rx=$2; shift 2
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "xgm_$rx" --output-format csv -d gpurun_out/${tag}_pmc_${n}_$c -- python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-latency $* > gpurun_out/${tag}_pmc_${n}_$c.log 2>&1
  done
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-include-regex "xgm_$rx" --output-format csv -d gpurun_out/${tag}_pmc_${n}_SQ -- python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-latency $* > gpurun_out/${tag}_pmc_${n}_SQ.log 2>&1
  python tools/pmc_parse.py gpurun_out/${tag}_pmc_${n}_FETCH_SIZE gpurun_out/${tag}_pmc_${n}_WRITE_SIZE gpurun_out/${tag}_pmc_${n}_SQ > gpurun_out/${tag}_pmc_${n}.txt; grep -c PMC gpurun_out/${tag}_pmc_${n}.txt
done
timeout 300 python bench.py --op OR --terms 5 --topk 100 --steps 20 --warmup 2 --ref-docs 0 --cpu-seconds 4 > gpurun_out/${tag}_bench_or5.json 2>gpurun_out/${tag}_or5.err; tail -2 gpurun_out/${tag}_or5.err
timeout 400 python bench.py --op PHRASE --topk 10 --steps 10 --warmup 2 --ref-docs 0 --cpu-seconds 4 > gpurun_out/${tag}_bench_phrase.json 2>gpurun_out/${tag}_phrase.err; tail -2 gpurun_out/${tag}_phrase.err
for a in "AND_NOT --terms 4 --required 2" "AND_MAYBE --terms 4 --required 2"; do
  n=$(echo $a | cut -d" " -f1 | tr A-Z a-z)
  timeout 300 python bench.py --op $a --steps 20 --warmup 2 --ref-docs 0 --cpu-seconds 3 > gpurun_out/${tag}_bench_$n.json 2>gpurun_out/${tag}_$n.err; tail -2 gpurun_out/${tag}_$n.err
done
python - <<PY
import json
for n in ('or5','phrase','and_not','and_maybe'):
    try:
        d=json.load(open('gpurun_out/${tag}_bench_%s.json'%n)); r=d['roofline']
        print(n,round(d['value']),r['kernel'],r['kernel_ms'],d['p50_latency_us'],'frac',r['frac'],r['basis'],'alg',r['algorithmic']['frac'],'parity',d['cpu_baseline']['parity_checked_queries'],r['model_counts'])
    except Exception as e: print(n,'failed',e)
PY
