cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_frozen_batch.py -m gpu -x -q 2>&1 | tail -3
for mode in none count; do
  timeout -k 5 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-latency --threads 0 --no-hook-parity --no-other-configs --ref-docs 0 --replay $mode > gpurun_out/r6_c2_$mode.json 2> gpurun_out/r6_c2_$mode.err
  python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/r6_c2_$mode.json").read().strip().splitlines()[-1])
    print("$mode", "qps", round(r["value"]), "ms/batch", round(r["ms_per_batch"], 4), "kernel", r["roofline"]["kernel"], "kernel_ms", r["roofline"]["kernel_ms"], "verified", r["last_batch_on_host_equals_synchronous_search"])
except Exception as e:
    print("$mode", "failed", e); print(open("gpurun_out/r6_c2_$mode.err").read()[-1500:])
PY
done
