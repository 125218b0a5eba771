# Round-2 run I: disjunction kernel occupancy variants (A/B builds under xapiand_amd/csrc/ab/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
for v in base orwb orwc orwd; do
  if [ $v = base ]; then unset XGM_LIB_PATH; else export XGM_LIB_PATH=$GRAFT_REPO_ROOT/xapiand_amd/csrc/ab/libxgm_$v.so; fi
  timeout 300 python bench.py --op OR --terms 5 --topk 100 --steps 20 --warmup 2 --no-cpu-baseline --no-latency --threads 0 > gpurun_out/${tag}_or_$v.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/${tag}_or_$v.json'));print('or5 $v',round(d['value']),d['roofline']['kernel_ms'])"
done
unset XGM_LIB_PATH
