# the N > 1 code path on this box's one GPU: ranks share it, gloo carries the exchange — a functional line (parity, cpu_baseline, fields), not a scaling figure
cd $GRAFT_REPO_ROOT
n=${1:-2}; docs=${2:-2000000}; per=${3:-200000}
XGM_BENCH_BACKEND=gloo XGM_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $n --steps 3 --warmup 1 --no-latency --threads 0 --docs-per-gpu $docs --ref-docs-per-shard $per --ref-seconds 6 > gpurun_out/r6_n${n}_shared_gpu.json 2> gpurun_out/r6_n${n}_shared_gpu.err
tail -c 500 gpurun_out/r6_n${n}_shared_gpu.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r6_n${n}_shared_gpu.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "n_gpus", "n_ranks_seen", "parity_checked_queries", "code_path")})
    print(d["config"]["workload"])
    c = d.get("cpu_baseline", {})
    print({k: c.get(k) for k in ("kind", "value", "cores", "shards", "docs_per_shard", "index_build_seconds", "error")}, c.get("all_cores"))
    print("roofline", {k: d["roofline"].get(k) for k in ("frac", "basis", "kernel", "kernel_ms")})
except Exception as e:
    print("failed", e)
PY
