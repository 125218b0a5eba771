# Round-2 run H: cost-model sweep for the positional class
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
for c in 0.03 0.1 0.25 0.6; do
  XGM_PHRASE_CAND_COST=$c timeout 300 python bench.py --op PHRASE --topk 10 --steps 10 --warmup 2 --no-cpu-baseline --threads 0 --no-latency > gpurun_out/${tag}_phrase_$c.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/${tag}_phrase_$c.json'));print('phrase cand_cost $c',round(d['value']),d['roofline']['kernel_ms'])"
done
