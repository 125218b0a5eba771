# usage: tools/kstat.sh <tag> [bench args]: mean duration per xgm kernel of a bench run (rocprofv3 --kernel-trace --stats), one line each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
rm -rf gpurun_out/ks_$tag
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$tag -- python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-latency --threads 0 $* > gpurun_out/ks_$tag.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/ks_$tag/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "xgm_" in r["Name"] and int(r["Calls"]) >= 10:
        print("$tag", r["Name"][:60].replace("(anonymous namespace)::", ""), "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "min", round(float(r["MinNs"]) / 1e3, 1), "max", round(float(r["MaxNs"]) / 1e3, 1))
PY
