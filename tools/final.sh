# Round evidence run: usage tools/final.sh <tag>   (one MI355X, ~12 min)
#   full GPU test suite; headline bench (C2: reference leg, server leg); rocprofv3 kernel stats; counter calibration; PMC passes
#   (FETCH_SIZE, WRITE_SIZE, SQ) for the conjunction, disjunction and positional kernels → traffic entries; C3 / C5 / sided lines with parity
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
# (GPU tests and the headline line: tools/final_a.sh)
# (counter calibration: tools/calib.sh, kept from the first pass of this round — profiles/r02_fetch_calib.txt)
: > gpurun_out/${tag}_traffic_entries.jsonl
# (round 6: + C5 in its credited, reference-identical batch mode — xgm_andw_list_kernel — and C2 with ProtoMSet's count — xgm_andw_all_kernel)
for w in "and3 andw_kernel" "or5 orw --op OR --terms 5 --topk 100" "phrase andw_kernel --op PHRASE --topk 10" "phrasef andw_list --op PHRASE --topk 10 --replay frozen" "and3count andw_all --replay count"; do
  set -- $w; n=$1; rx=$2; shift 2
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "xgm_$rx" --output-format csv -d gpurun_out/${tag}_pmc_${n}_$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-hook-parity --ref-docs 0 --no-latency --threads 0 $* > gpurun_out/${tag}_pmc_${n}_$c.log 2>&1
  done
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-include-regex "xgm_$rx" --output-format csv -d gpurun_out/${tag}_pmc_${n}_SQ -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-hook-parity --ref-docs 0 --no-latency --threads 0 $* > gpurun_out/${tag}_pmc_${n}_SQ.log 2>&1
  python tools/pmc_parse.py gpurun_out/${tag}_pmc_${n}_FETCH_SIZE gpurun_out/${tag}_pmc_${n}_WRITE_SIZE gpurun_out/${tag}_pmc_${n}_SQ > gpurun_out/${tag}_pmc_${n}.txt
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-hook-parity --ref-docs 0 --no-latency --threads 0 $* > gpurun_out/${tag}_model_${n}.json 2>/dev/null
  python tools/traffic.py gpurun_out/${tag}_pmc_${n}.txt gpurun_out/${tag}_model_${n}.json xgm_$rx >> gpurun_out/${tag}_traffic_entries.jsonl
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_${n} -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --no-hook-parity --ref-docs 0 --no-latency --threads 0 $* > gpurun_out/${tag}_prof_${n}.log 2>&1
done
python - <<PY
import json
ents=[json.loads(l) for l in open('gpurun_out/${tag}_traffic_entries.jsonl')]
json.dump({"entries":ents}, open('profiles/traffic.json','w'), indent=1)      # bench.py below quotes these
json.dump({"entries":ents}, open('gpurun_out/${tag}_traffic.json','w'), indent=1)
for e in ents: print('traffic',e['op'],e['kernel'],round(e['hbm_bytes_per_launch']/1e9,3),'GB/launch')
PY
# (XGM_FINAL_NO_BENCH=1: the headline line is taken afterwards by tools/final_a.sh, with profiles/traffic.json = gpurun_out/${tag}_traffic.json in place)
if [ -z "$XGM_FINAL_NO_BENCH" ]; then cp gpurun_out/${tag}_bench.json gpurun_out/${tag}_bench_first.json 2>/dev/null; timeout 1700 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 300 gpurun_out/${tag}_bench.err; fi
# (C3 / C5: sub-legs of the default line since round 4 — other_configs — with parity and PMC-stamped rooflines)
# the merge launch instead of the last-unit merge, for the record (same build, same box)
XGM_NO_FUSED_MERGE=1 timeout 300 python bench.py --no-cpu-baseline --no-other-configs --threads 0 > gpurun_out/${tag}_bench_merge_launch.json 2>/dev/null
if [ -z "$XGM_FINAL_TRIM" ]; then
for a in "AND_NOT --terms 4 --required 2" "AND_MAYBE --terms 4 --required 2" "FILTER --terms 3 --required 2" "AND --terms 2"; do
  n=$(echo $a | tr -d '-' | tr ' ' '_' | tr A-Z a-z)
  timeout 300 python bench.py --op $a --steps 5 --warmup 1 --ref-docs 0 --cpu-seconds 3 --threads 0 > gpurun_out/${tag}_bench_$n.json 2>gpurun_out/${tag}_$n.err
done
# the N > 1 code path (C4's protocol: shards on ranks, all-gather, merge) on this box's one GPU: two ranks share it, gloo carries the
# exchange — a functional line (the ranks contend for one GPU), not a scaling figure; the driver measures N = 2, 4, 8 on its node
XGM_BENCH_BACKEND=gloo XGM_BENCH_SHARE_GPU=1 timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-latency --threads 0 > gpurun_out/${tag}_n2_shared_gpu.json 2> gpurun_out/${tag}_n2_shared_gpu.err; tail -c 400 gpurun_out/${tag}_n2_shared_gpu.json
fi
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/${tag}_bench*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']; c=d.get('cpu_baseline',{})
        print(f.split('/')[-1],round(d['value']),r['kernel'],'kernel_ms',round(r['kernel_ms'],3),'p50',d['p50_latency_us'],'frac',round(r['frac'],3),r['basis'],'model',r['model_frac'] and round(r['model_frac'],3),'alg',round(r['algorithmic']['frac'],3),'parity',c.get('parity_checked_queries'),'cpu',c.get('kind'),c.get('value'))
    except Exception as e: print(f,'failed',e)
try:
    d=json.load(open('gpurun_out/${tag}_bench.json')); print(json.dumps(d.get('server_mode'))); print(json.dumps({k:v for k,v in d['cpu_baseline'].items() if k!='sample'})[:1500])
except Exception as e: print('no headline line in this call', e)
PY
find gpurun_out/${tag}_prof_* -name "*kernel_stats.csv" | head
