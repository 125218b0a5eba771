# usage: tools/pmc1.sh <tag> "<counters>" <kernel regex> [bench args]: per-kernel mean of a few PMC counters (one rocprofv3 --pmc pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; ctr=$2; rx=$3; shift 3
rm -rf gpurun_out/pmc1_$tag
timeout -k 5 240 rocprofv3 --kernel-trace --pmc $ctr --kernel-include-regex "$rx" --output-format csv -d gpurun_out/pmc1_$tag -- python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-latency --threads 0 $* > gpurun_out/pmc1_$tag.log 2>&1
python tools/pmc_parse.py gpurun_out/pmc1_$tag | sed "s/^/$tag /"
