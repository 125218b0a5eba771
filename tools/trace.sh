cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-latency $* > gpurun_out/trace.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "xgm_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(("match" if "merge" not in r["Kernel_Name"] else "merge"), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# last 12 steps = 24 events before the alg-bytes passes; just print a window in the middle
mid = [e for e in ev]
for i in range(6, min(len(mid), 26)):
    k, s, e = mid[i]
    gap = s - mid[i - 1][2]
    print("%-6s dur %8.1f us   gap-before %7.1f us" % (k, (e - s) / 1e3, gap / 1e3))
PY
