# round 6: the reference-identical batch mode of C5 on the MI355X — the new GPU test, then C5 in both modes (queries/s, kernel ms)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_frozen_batch.py -m gpu -x -q > gpurun_out/r6_c5_test.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_c5_test.log
tail -3 gpurun_out/r6_c5_test.log
for mode in none frozen; do
  timeout -k 5 400 python bench.py --op PHRASE --topk 10 --steps 8 --warmup 2 --no-cpu-baseline --no-latency --threads 0 --no-hook-parity --ref-docs 0 --replay $mode > gpurun_out/r6_c5_$mode.json 2> gpurun_out/r6_c5_$mode.err
  python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/r6_c5_$mode.json").read().strip().splitlines()[-1])
    print("$mode", "qps", round(r["value"]), "ms/batch", round(r["ms_per_batch"], 4), "kernel", r["roofline"]["kernel"], "kernel_ms", r["roofline"]["kernel_ms"], "solo", r["roofline"]["kernel_ms_one_in_flight"])
except Exception as e:
    print("$mode", "failed", e); print(open("gpurun_out/r6_c5_$mode.err").read()[-1500:])
PY
done
