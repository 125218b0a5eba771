cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_all.py tests/test_gpu_variants.py -m gpu -x -q -k "replay or all_matches" > gpurun_out/r5h_pytest.log 2>&1; tail -2 gpurun_out/r5h_pytest.log
timeout 300 python tools/replay_prof.py OR 16 2>&1 | tail -1 | sed 's/^/parallel: /'
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5h_replay_or -- python tools/replay_prof.py OR 16 > gpurun_out/r5h_replay_or.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r5h_replay_or/**/*kernel_stats.csv", recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows:
    if "replay" in r["Name"] or "match_sorted" in r["Name"] or "xgm_all" in r["Name"]:
        print(r["Name"][:60].replace("(anonymous namespace)::", ""), "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1))
PY
