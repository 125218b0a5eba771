# round 5, call C: the new disjunction kernel (xgm_orw2_kernel) on the MI355X: A/B against the old one and between occupancy variants; parity at
# config size; kernel split of xgm_search_replay
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OR="--no-other-configs --op OR --terms 5 --topk 100 --steps 8"
{
XGM_NO_ORW2=1 bash tools/ab_run.sh "$OR" default | sed 's/^default/old-kernel/'
bash tools/ab_run.sh "$OR" default o2d4 o2s2
XGM_NO_ORW2=1 bash tools/ab_run.sh "$OR" default | sed 's/^default/old-kernel/'
bash tools/ab_run.sh "$OR" default
} > gpurun_out/r5c_ab.txt 2>&1
cat gpurun_out/r5c_ab.txt
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_mixed.py -m gpu -x -q -k "not hook" > gpurun_out/r5c_pytest.log 2>&1; tail -4 gpurun_out/r5c_pytest.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5c_replay_or -- python tools/replay_prof.py OR 16 > gpurun_out/r5c_replay_or.log 2>&1; tail -2 gpurun_out/r5c_replay_or.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5c_replay_ph -- python tools/replay_prof.py PHRASE 48 > gpurun_out/r5c_replay_ph.log 2>&1; tail -2 gpurun_out/r5c_replay_ph.log
python - <<'PY'
import csv, glob
for tag in ("or", "ph"):
    f = glob.glob("gpurun_out/r5c_replay_%s/**/*kernel_stats.csv" % tag, recursive=True)
    if not f: print(tag, "no stats"); continue
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:8]:
        print(tag, r["Name"][:70].replace("(anonymous namespace)::", ""), "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "total_ms", round(float(r["TotalDurationNs"]) / 1e6, 1))
PY
