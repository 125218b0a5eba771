# usage: tools/prof.sh <tag>: bench line + rocprofv3 kernel stats into gpurun_out/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout 300 python bench.py > gpurun_out/bench_${tag}.json 2> gpurun_out/bench_${tag}.err
tail -c 900 gpurun_out/bench_${tag}.json
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python bench.py --steps 20 --no-cpu-baseline > gpurun_out/prof_${tag}.log 2>&1
find gpurun_out/prof_${tag} -name "*kernel_stats.csv" | head -2
f=$(find gpurun_out/prof_${tag} -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-200
