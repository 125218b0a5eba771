cd $GRAFT_REPO_ROOT
for v in "" "XGM_NO_OR_FUSED_MERGE=1"; do
  env $v timeout -k 5 400 python bench.py --op OR --terms 5 --topk 100 --steps 8 --warmup 2 --no-cpu-baseline --no-latency --threads 0 --no-hook-parity --ref-docs 0 > gpurun_out/r6_c3.json 2> gpurun_out/r6_c3.err
  python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/r6_c3.json").read().strip().splitlines()[-1])
    print("$v", "qps", round(r["value"]), "ms/batch", round(r["ms_per_batch"], 4), "kernel", r["roofline"]["kernel"], "kernel_ms", r["roofline"]["kernel_ms"], "verified", r["last_batch_on_host_equals_synchronous_search"])
except Exception as e:
    print("failed", e); print(open("gpurun_out/r6_c3.err").read()[-1500:])
PY
done
python -m pytest tests/test_gpu_variants.py -m gpu -x -q -k "OR_FUSED or SEED_SCALE" 2>&1 | tail -2
python -m pytest tests/test_gpu_stress.py tests/test_gpu_parity.py tests/test_gpu_hook_b1.py -m gpu -x -q 2>&1 | tail -2
