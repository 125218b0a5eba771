# usage: tools/ab_run.sh "<bench args>" <lib name|default> [...]: one bench line per A/B library (xapiand_amd/csrc/ab/libxgm_<name>.so)
cd $GRAFT_REPO_ROOT
args=$1; shift
for n in "$@"; do
  if [ "$n" = default ]; then unset XGM_LIB_PATH; else export XGM_LIB_PATH=$PWD/xapiand_amd/csrc/ab/libxgm_$n.so; fi
  timeout 300 python bench.py $args --no-cpu-baseline --threads 0 --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$n', 'qps', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(r['kernel_ms'],4), r['kernel'])"
done
