# usage: tools/ab_build.sh <name> <extra hipcc flags...>: A/B build of libxgm into xapiand_amd/csrc/ab/libxgm_<name>.so
set -e
name=$1; shift
cd /root/repo/xapiand_amd/csrc
objs=""
for f in xgm_api.cc xgm_plan.cc xgm_segment_build.cc xgm_glass.cc xgm_kernels.hip xgm_dense_and.hip xgm_or.hip xgm_synth.hip xgm_dense.hip xgm_all.hip xgm_replay.hip xgm_frozen.hip xgm_count.hip; do
  o=/tmp/ab_${name}_${f%.*}.o
  case "$f" in
    xgm_or.hip|xgm_kernels.hip|xgm_dense_and.hip) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -x hip "$@" -c $f -o $o 2>/dev/null ;;
    *) o=${f%.*}.o ;;
  esac
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libxgm_${name}.so $objs
echo built ab/libxgm_${name}.so
