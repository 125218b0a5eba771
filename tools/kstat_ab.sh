# usage: tools/kstat_ab.sh "<bench args>" <lib|default> ...: tools/kstat.sh per A/B library
cd $GRAFT_REPO_ROOT
args=$1; shift
for n in "$@"; do
  if [ "$n" = default ]; then unset XGM_LIB_PATH; else export XGM_LIB_PATH=$PWD/xapiand_amd/csrc/ab/libxgm_$n.so; fi
  bash tools/kstat.sh $n $args | grep -v merge
done
