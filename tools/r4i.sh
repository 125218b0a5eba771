cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=r04i
timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_positional.py tests/test_gpu_parity.py -m gpu -q -k "C5 or C2 or positional or phrase" > gpurun_out/${tag}_pytest.log 2>&1; tail -3 gpurun_out/${tag}_pytest.log
B="python bench.py --no-cpu-baseline --no-other-configs --no-latency --threads 0"
P="$B --op PHRASE --topk 10 --steps 6 --warmup 1"
timeout 200 $P > gpurun_out/${tag}_phrase.json 2>/dev/null
timeout 200 $B > gpurun_out/${tag}_c2.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "xgm_andw" --output-format csv -d gpurun_out/${tag}_pmc_$c -- $P --steps 2 > /dev/null 2>&1; done
python tools/pmc_parse.py gpurun_out/${tag}_pmc_FETCH_SIZE gpurun_out/${tag}_pmc_WRITE_SIZE | grep -v TALLY
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/${tag}_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], round(d['value']), 'ms/batch', round(d['ms_per_batch'],4), 'kernel_ms', round(r['kernel_ms'],4), 'verified', d['last_batch_on_host_equals_synchronous_search'])
    except Exception as e: print(f,'failed',e)
PY
