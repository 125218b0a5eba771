#!/usr/bin/env python
"""Benchmark of the match/rank hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1 needs one process per GPU: when started as a plain `python bench.py --gpus N` the script re-executes
itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (the form
the driver uses directly), so both invocations work.

Workload (N = 1, BASELINE.json configs[1], "C2"): 10 M-doc / 1 M-term Zipf synthetic index resident in HBM,
3-term conjunctive BM25 queries, top-10.  One "step" = one pass of the hot path over one batch of 256 queries
AS THE MATCHER HOOK RECEIVES THEM (terms, operator, first/maxitems, BM25 parameters, merged statistics):
xgm_get_mset_batch_device = dictionary lookups + BM25Weight::init + leaf ordering (xgm_plan_query) → decode →
intersect → BM25 → top-k → merge.  N > 1 is configs[3] scaled ("C4"): the corpus has 10 M × N documents sharded
N ways exactly like the reference (global doc g → shard (g-1) % N, src/xapian/backends/multi.h:38-73), every rank
searches its shard with the MERGED collection statistics, then one RCCL all-gather of the per-shard top-k records
+ a device-side merge (xgm_merge_shards_device).  Weak scaling: per-GPU work is fixed.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and `cpu_baseline`:

roofline.achieved / frac  — bytes the dominant kernel MOVED per launch ÷ its HIP-event duration ÷ 8 TB/s.  The
    bytes are the HBM traffic from the committed PMC passes (profiles/traffic.json, `basis: "pmc"`) when they exist
    for this very workload, else the kernel's own request tallies at memory-sector granularity (`basis: "model"`,
    measured live: xgm_last_batch_traffic).  roofline.algorithmic carries SURVEY §8(d)'s format-independent
    figure (Σ df·8 + S·4 + P·4 + k·16 per query) ÷ the same duration: the probe containers answer "is d in t, with
    which wdf" in one byte, so that figure can exceed what any kernel that READ those bytes could reach — it is a
    speed-up over the posting-scan algorithm, not evidence of HBM efficiency, and is reported as such.
cpu_baseline — the REAL reference (oracle/_ref/xapian_ref = the vendored Xapian's Enquire::get_mset) timed on this
    box's host cores on a glass index of the same corpus built here with the reference's own WritableDatabase
    (1/10 of the configuration's documents: indexing is the reference's bottleneck, see tools/ref_index.py), the
    oracle port timed on the identical index (→ port_over_reference) and the port at the full 10 M documents.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK = 8.0e12          # B/s, /opt/skills/guides/MI355X_MICROARCH.md
SECTOR = 64                # bytes one random gather costs at the memory side (tools/fetch_calib.hip, profiles/)
CORPUS_SEED = 0x5EED0001
QUERY_SEED = 0x5EED0002
BATCH = 256


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--docs-per-gpu", type=int, default=10_000_000)
    ap.add_argument("--vocab", type=int, default=1_000_000)
    ap.add_argument("--op", default="AND")
    ap.add_argument("--terms", type=int, default=3)
    ap.add_argument("--required", type=int, default=1, help="AND_NOT / AND_MAYBE / FILTER: terms of the left-hand AND")
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--stripe-bits", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--ref-docs", type=int, default=-1, help="documents of the reference glass index built on this box (0: skip the reference leg; "
                    "default: 1/5 of the configuration's documents — indexing 10 M documents through the reference's WritableDatabase takes "
                    "~3.5 min even on 128 cores; `--ref-docs 10000000` reproduces profiles/r02_reference_full.json)")
    ap.add_argument("--no-latency", action="store_true", help="skip the one-query-in-flight leg (profiling runs)")
    ap.add_argument("--threads", type=int, default=64, help="server leg: T host threads, each with one xgm_get_mset_batch(nq=1) in flight (0: skip)")
    return ap.parse_args()


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: become N ranks under torch.distributed.run."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def workload_name(args, world, n_docs_global, k):
    """BASELINE.json config the run corresponds to (C2 is the one the metric is quoted on)."""
    shape = {"AND": "%d-term conjunctive" % args.terms, "OR": "%d-term disjunctive" % args.terms, "PHRASE": "2-3-term phrase (positions)"}.get(
        args.op, "%s (%d-term AND, %d on the right)" % (args.op, args.required, args.terms - args.required))
    if world > 1:
        return "C4 (weak-scaled): %dM-doc index sharded %d ways, %s BM25 top-%d, RCCL top-k all-gather" % (n_docs_global // 1000000, world, shape, k)
    cfg = {"AND": "C2", "OR": "C3", "PHRASE": "C5"}.get(args.op, "next (SURVEY 8f.2)")
    return "%s: %dM-doc / %dM-term Zipf index, %s BM25 top-%d, 1 MI355X" % (cfg, args.docs_per_gpu // 1000000, args.vocab // 1000000, shape, k)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_spawn(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    import helpers as H        # query pool shared with tests/test_gpu_configs.py (PHRASE: the corpus restated in Python)
    from xapiand_amd import Database, Query, _lib
    from xapiand_amd.distributed import ShardedSearcher

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if os.environ.get("XGM_BENCH_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # XGM_BENCH_BACKEND=gloo + XGM_BENCH_SHARE_GPU=1: the N > 1 code path on a box with ONE GPU (RCCL refuses two ranks per device) — a
        # functional check of this script, not a measurement
        backend = os.environ.get("XGM_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # ---- index: this rank's shard, generated + inverted + block-encoded on the GPU ----------------
    n_docs_global = args.docs_per_gpu * world
    t0 = time.time()
    db = Database.synthetic(CORPUS_SEED, n_docs_global, args.vocab, n_shards=world, shard=rank, device=local_rank,
                            stripe_bits=args.stripe_bits)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    info = db.info()
    # One explicit (non-null) HIP stream for searches and collectives: the xgm_*_device calls are then asynchronous,
    # so the host plans batch i+1 while the GPU runs batch i (ShardedSearcher binds the index to torch's current stream).
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)

    # ---- queries + merged statistics (Enquire::add_prepared_mset: Σ over shards): one all-reduce ----
    pool = H.bench_pool(args.op, args.terms, args.required, n_docs_global, args.vocab, seed=QUERY_SEED)
    k = args.topk
    searcher = ShardedSearcher(db, rank, world, dev)
    sided = args.op in ("AND_NOT", "AND_MAYBE", "FILTER")
    qobjs = [Query(q["op"], q["terms"], n_required=args.required if sided else 0) for q in pool]
    descs, gstats = searcher.describe(qobjs, 0, k)                 # what the hook is handed per get_mset
    plans = searcher.prepare(qobjs, 0, k)                          # bookkeeping only (algorithmic bytes, parity leg)
    L = _lib.lib()
    n_timed = len(pool) - 100
    n_batches = n_timed // BATCH

    def batch_of(lo):
        d = (_lib.QueryDesc * BATCH)(*[descs[(lo + i) % len(pool)] for i in range(BATCH)])
        g = (_lib.GlobalStats * BATCH)(*[gstats[(lo + i) % len(pool)] for i in range(BATCH)])
        return d, g
    batches = [batch_of(100 + b * BATCH) for b in range(n_batches)]
    warm = batch_of(0)
    timed_plans = plans[100:]

    def step(b):
        # plan + search (+ RCCL all-gather of top-k + xgm_merge_shards_device when sharded)
        return searcher.run_descs(b[0], b[1], BATCH, k)

    for _ in range(args.warmup):
        step(warm)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    db.set_profiling(1)
    L.xgm_debug_host_ns((C.c_uint64 * 8)())           # reset the host-side section timers
    t0 = time.perf_counter()
    host_s = 0.0                                   # time the host spends inside the (asynchronous) calls: plan + enqueue
    for s in range(args.steps):
        h0 = time.perf_counter()
        step(batches[s % n_batches])
        host_s += time.perf_counter() - h0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    hn = (C.c_uint64 * 8)()
    L.xgm_debug_host_ns(hn)
    n_launch = max(1, int(hn[3]))
    host_sections = {"plan_queries": round(hn[0] / 1e3 / args.steps, 1), "plan_batch": round(hn[1] / 1e3 / n_launch, 1),
                     "stage_enqueue_launch": round(hn[2] / 1e3 / n_launch, 1), "staging_memcpy": round(hn[4] / 1e3 / n_launch, 1), "upload_enqueue": round(hn[5] / 1e3 / n_launch, 1), "match_launch": round(hn[6] / 1e3 / n_launch, 1), "merge_launch": round(hn[7] / 1e3 / n_launch, 1), "launches_per_step": round(n_launch / args.steps, 2)}
    kernel_ms = db.last_kernel_ms()            # mean match-kernel duration over the timed steps (HIP events)
    kernel_name = db.last_kernel_name()
    db.set_profiling(0)
    cdev = dev if (world == 1 or dist.get_backend() == "nccl") else torch.device("cpu")      # where the small collectives of this script live
    t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    qps = args.steps * BATCH / elapsed             # queries answered over the WHOLE (sharded) index per second

    # ---- per-launch byte counts, outside the timed region ------------------------------------------------------
    # algorithmic: Σ_t df_t·8 + S·4 + P·4 + k·16 per query; model: the kernel's own request tallies
    db.set_profiling(0 if os.environ.get("XGM_BENCH_NO_TALLY") else 2)   # the tallying instantiation of the wave kernels (tools/units.py switches it off: it wants the product kernel's own timeline)
    alg_bytes, model_sector, model_useful, tallies = [], [], [], []
    for b in range(n_batches):
        step(batches[b])
        torch.cuda.synchronize()
        h = searcher._buffers(BATCH, k)["hdrs"].cpu().numpy().view(np.uint8).reshape(BATCH, 32)   # this shard's own header
        matches = h[:, 8:16].copy().view(np.uint64).reshape(BATCH) & np.uint64((1 << 63) - 1)      # (bit 63: lower bound only, include/xgm.h)
        post = sum(L.xgm_query_postings_bytes(db._h, C.byref(p)) for p in timed_plans[b * BATCH:(b + 1) * BATCH])
        tl = (C.c_uint64 * 10)()
        have_tally = L.xgm_last_batch_traffic(db._h, tl, 10) == 0 and kernel_name in ("xgm_andw_kernel", "xgm_orw_kernel", "xgm_dense_kernel")
        bmpw, probes, blkw, hdrs_, dls, aux, cands, npos, probes_raw, dls_raw = [int(x) for x in tl]
        alg_bytes.append(post + int(matches.sum()) * 4 + npos * 4 + BATCH * k * 16)
        if have_tally:
            stream_b = 4 * bmpw + 4 * blkw + 12 * hdrs_ + 4 * aux + 16 * cands + 4 * npos
            model_sector.append(stream_b + SECTOR * (probes + dls))
            model_useful.append(stream_b + probes_raw + 4 * dls_raw)
            tallies.append(dict(bitmap_words=bmpw, probe_sectors=probes, probes=probes_raw, payload_words=blkw, block_headers=hdrs_,
                                doclen_sectors=dls, doclen_gathers=dls_raw, aux_words=aux, candidates_out=cands, positions=npos))
    db.set_profiling(0)
    used = [s % n_batches for s in range(args.steps)]
    bytes_per_launch = float(np.mean([alg_bytes[i] for i in used]))
    kt = kernel_ms * 1e-3 if kernel_ms and kernel_ms > 0 else None
    alg_rate = bytes_per_launch / kt if kt else 0.0
    model_bytes = float(np.mean([model_sector[i] for i in used])) if model_sector else None
    useful_bytes = float(np.mean([model_useful[i] for i in used])) if model_useful else None

    # ---- planning cost (inside every timed step): host microseconds per query ------------------------------------
    plan_us = L.xgm_debug_plan_us(db._h, batches[0][0], batches[0][1], BATCH, 20)

    # ---- latency mode: one query in flight, host-timed around plan + search incl. H2D/D2H ---------------------
    lat = []
    if (rank == 0 or world > 1) and not args.no_latency:
        one_hits = (_lib.Hit * k)()
        one_hdr = _lib.ResultHdr()
        db.set_stream(0)
        for i in range(min(1000, n_timed)):                 # 1 000 queries (the first 20 warm up)
            d1 = (_lib.QueryDesc * 1)(descs[100 + i])
            g1 = (_lib.GlobalStats * 1)(gstats[100 + i])
            a = time.perf_counter()
            _lib.check(L.xgm_get_mset_batch(db._h, d1, g1, 1, k, one_hits, C.byref(one_hdr)))
            if i >= 20:
                lat.append(time.perf_counter() - a)
    lat.sort()

    # ---- server mode: T host threads, each with ONE query in flight (Xapiand's http_client_pool shape) ----------
    server = None
    if rank == 0 and world == 1 and args.threads > 0:
        server = server_leg(db, descs, gstats, k, args.threads, n_timed)

    # ---- HBM traffic of the dominant kernel: from the committed PMC passes (tools/final.sh → profiles/traffic.json),
    # which cannot be collected from inside this process; only quoted when measured on this very workload
    traffic, traffic_note = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    sha_path = os.path.join(ROOT, "xapiand_amd", "csrc", "libxgm.so.sha")
    lib_sha = open(sha_path).read().strip() if os.path.exists(sha_path) else None
    stale_traffic = False
    if os.path.exists(tpath):
        for tr in json.load(open(tpath)).get("entries", []):
            if tr.get("lib_sha") != lib_sha:
                stale_traffic = True          # counters of another build of the library: not this kernel's traffic
                continue
            if (tr["kernel"] == kernel_name and tr["op"] == args.op and tr["docs_per_gpu"] == args.docs_per_gpu and tr["top_k"] == k and
                    tr["terms"] == (args.terms if args.op != "PHRASE" else 0) and tr["batch"] == BATCH and tr.get("required", 1) == (args.required if sided else 1)):
                traffic, traffic_note = tr["hbm_bytes_per_launch"], tr["note"]

    # ---- N > 1: every rank's own match kernel against its own GPU's roofline (per-GPU bytes use the shard's own df) ------
    per_rank = None
    if world > 1:
        mine = torch.tensor([kernel_ms if kernel_ms and kernel_ms > 0 else 0.0, float(traffic or model_bytes or bytes_per_launch)], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = []
        for r, v in enumerate(allr):
            ms, by = float(v[0].item()), float(v[1].item())
            per_rank.append({"rank": r, "kernel_ms": ms, "bytes_per_launch": by, "achieved": (by / (ms * 1e-3) / 1e9) if ms > 0 else None,
                             "frac": (by / (ms * 1e-3) / HBM_PEAK) if ms > 0 else None})

    result = None
    if rank == 0:
        basis = "pmc" if traffic else ("model" if model_bytes else "algorithmic")
        moved = traffic if traffic else (model_bytes if model_bytes else bytes_per_launch)
        achieved = moved / kt if kt else 0.0
        result = {
            "metric": "queries/sec + p50 latency, %dM-doc synthetic index, %s, top-%d" % (
                args.docs_per_gpu // 1000000, {"AND": "%d-term AND" % args.terms, "OR": "%d-term OR" % args.terms,
                                               "PHRASE": "2-3-term PHRASE"}.get(args.op, "%d-term %s" % (args.terms, args.op)), k),
            "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 postings + f64 BM25", "data": "synthetic",
            "config": {"workload": workload_name(args, world, n_docs_global, k),
                       "docs_per_gpu": args.docs_per_gpu, "docs_total": n_docs_global, "vocab": args.vocab, "op": args.op,
                       "terms_per_query": args.terms, "top_k": k, "batch": BATCH, "parallelism": "shard%d" % world,
                       "corpus_seed": hex(CORPUS_SEED), "query_seed": hex(QUERY_SEED),
                       "step": "xgm_get_mset_batch_device: plan (lookups, BM25 init, leaf order) + match + merge, 256 queries"},
            "host_ms_per_step": round(1e3 * host_s / args.steps, 4), "host_us_per_step": host_sections, "p50_latency_us": lat[len(lat) // 2] * 1e6 if lat else None,
            "p99_latency_us": lat[int(len(lat) * 0.99)] * 1e6 if lat else None,
            "plan_us_per_query": plan_us,
            "index": {"postings": info.n_postings, "blocks": info.n_blocks, "payload_bytes": info.payload_bytes,
                      "device_bytes": info.device_bytes, "bytes_per_posting": info.device_bytes / max(1, info.n_postings),
                      "build_seconds": build_s},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "basis": basis, "traffic": traffic, "traffic_note": traffic_note,
                         "traffic_lib_sha": lib_sha if traffic else None, "traffic_entries_of_another_build_ignored": stale_traffic,
                         "kernel_ms": kernel_ms,
                         "model_min_bytes": model_bytes, "model_useful_bytes": useful_bytes,
                         "model_frac": (model_bytes / kt / HBM_PEAK) if (model_bytes and kt) else None,
                         "model_note": "requests tallied by the kernel itself (xgm_last_batch_traffic): streamed words at 4 B + %d B per DISTINCT "
                                       "memory sector a round of one-byte container probes / doclen gathers touches" % SECTOR,
                         "model_counts": tallies[0] if tallies else None,
                         "algorithmic": {"bytes_per_launch": bytes_per_launch, "achieved": alg_rate / 1e9, "frac": alg_rate / HBM_PEAK,
                                         "note": "SURVEY 8(d): sum df*8 + S*4 + P*4 + k*16 per query; not bytes this design reads"}},
        }
        if per_rank:
            ok = [p for p in per_rank if p["achieved"]]
            result["roofline"]["per_rank"] = per_rank
            result["roofline"]["aggregate"] = {"achieved": sum(p["achieved"] for p in ok), "peak": HBM_PEAK / 1e9 * world, "unit": "GB/s",
                                               "frac": (sum(p["frac"] for p in ok) / len(ok)) if ok else None,
                                               "note": "sum over the ranks of bytes moved per launch / that rank's kernel time; frac = mean of the per-rank fractions"}
        if server:
            result["server_mode"] = server

    # ---- CPU baseline: the real reference + the oracle port, on this box's host cores -------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(db, pool[100:], args, k, timed_plans)
    if rank == 0:
        print(json.dumps(result), flush=True)
    db.close()
    if world > 1:
        dist.destroy_process_group()


def server_leg(db, descs, gstats, k, n_threads, n_timed):
    """Xapiand's shape of load (reference src/manager.cc:161): T host threads, each answering ONE query at a time through
    xgm_get_mset_batch(nq = 1) (plan + search), without any caller-side batching — native threads
    (xgm_debug_concurrent_searches), first with every call launching on its own, then with the index's opt-in
    micro-batching queue (xgm_index_set_batching) that lets concurrent single-query calls share launches."""
    from xapiand_amd import _lib
    L = _lib.lib()
    db.set_stream(0)
    per = max(8, min(200, n_timed // max(1, n_threads) * 4))
    d0 = C.cast(C.byref(descs, 100 * C.sizeof(_lib.QueryDesc)), C.POINTER(_lib.QueryDesc))
    g0 = C.cast(C.byref(gstats, 100 * C.sizeof(_lib.GlobalStats)), C.POINTER(_lib.GlobalStats))
    out = {"threads": n_threads, "queries_per_thread": per}
    for name, max_batch in (("unbatched", 0), ("batched", 256)):
        _lib.check(L.xgm_index_set_batching(db._h, max_batch))
        lat = (C.c_double * (n_threads * per))()
        L.xgm_debug_concurrent_searches(db._h, d0, g0, n_timed, n_threads, 8, k, lat)            # warm-up
        wall = L.xgm_debug_concurrent_searches(db._h, d0, g0, n_timed, n_threads, per, k, lat)
        v = sorted(lat)
        out[name] = {"value": len(v) / wall if wall > 0 else None, "unit": "queries/s", "p50_us": v[len(v) // 2], "p99_us": v[int(len(v) * 0.99)]}
        if max_batch:
            info = (C.c_uint64 * 3)()
            L.xgm_debug_batching_info(db._h, info)
            out[name]["mean_batch"] = info[1] / max(1, info[0])
    _lib.check(L.xgm_index_set_batching(db._h, 0))
    return out


def time_port(ora, sample, op, k, seconds, n_required, checker=None):
    """oracle/xgm_oracle.cc (the reference algorithm restated) on 1 thread over `sample`, repeated for ~seconds."""
    import helpers as H
    done, spent, passes, lat = 0, 0.0, 0, []
    while spent < seconds and passes < 50:
        for qi, q in enumerate(sample):
            a = time.perf_counter()
            hits, _ = H.oracle_search(ora, op, q["terms"], 0, k, n_required=n_required)
            dt = time.perf_counter() - a
            spent += dt
            lat.append(dt)
            done += 1
            if passes == 0 and checker:
                checker(qi, hits)
        passes += 1
    lat.sort()
    return dict(value=done / spent, unit="queries/s", cores=1, p50_ms=lat[len(lat) // 2] * 1e3, seconds=spent, passes=passes)


def port_all_cores(ora, sample, op, k, seconds, n_required=0):
    """The same port on every host core at once (oracle/xgm_oracle.cc::xgo_search_many, one C++ thread per core)."""
    import helpers as H
    n_threads = max(1, os.cpu_count() or 1)
    flat = [t.encode() for q in sample for t in q["terms"]]
    n_terms = (C.c_uint32 * len(sample))(*[len(q["terms"]) for q in sample])
    terms = (C.c_char_p * len(flat))(*flat)
    lens = (C.c_uint32 * len(flat))(*[len(t) for t in flat])
    done, wall = C.c_uint64(), C.c_double()
    ol = H.olib()
    ol.xgo_search_many.restype = C.c_int
    ol.xgo_search_many.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_char_p), C.POINTER(C.c_uint32),
                                   C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    opcode = H.OPS[op] | ((n_required or 1) << 8 if op in H.SIDED else 0)
    rc = ol.xgo_search_many(ora.oracle_index(), opcode, len(sample), n_terms, terms, lens, 0, k, n_threads, float(seconds),
                            C.byref(done), C.byref(wall))
    assert rc == 0
    return {"value": done.value / wall.value, "unit": "queries/s", "cores": n_threads, "seconds": wall.value}


def reference_leg(args, sample, k, n_required, full, ora_full):
    """The real reference on this box: build a glass index of the first --ref-docs documents of the corpus with
    the reference's own WritableDatabase (parallel slices + Database::compact, tools/ref_index.py), time
    Enquire::get_mset on 1 thread and on every core (one Database handle per thread) with
    `oracle/_ref/xapian_ref time`, then time the port on the identical postings."""
    import shutil
    import tempfile
    import helpers as H
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ref_index
    from xapiand_amd import Database
    ref_docs = args.ref_docs if args.ref_docs >= 0 else (args.docs_per_gpu // 10 if args.op == "PHRASE" else args.docs_per_gpu // 5)
    if not H.have_xapian_ref() or ref_docs <= 0:
        return None
    # the index lives in memory-backed storage when there is one: indexing through WritableDatabase is write-heavy
    tmp = tempfile.mkdtemp(prefix="xgm_ref_", dir="/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None)
    try:
        dbdir = os.path.join(tmp, "glass")
        binfo = ref_index.build(dbdir, ref_docs, nopos=args.op != "PHRASE", vocab=args.vocab)
        qfile = os.path.join(tmp, "q.txt")
        H.write_queries(qfile, [dict(q, first=0, maxitems=k) for q in sample])
        cores = max(1, os.cpu_count() or 1)
        one = json.loads(H.xapian_ref("time", qfile, 1, 2, dbdir))
        # enough repeats that every thread answers a few dozen queries
        rep = max(2, (cores * 24 + len(sample) - 1) // len(sample))
        many = json.loads(H.xapian_ref("time", qfile, cores, rep, dbdir))
        # the port on the same postings: the same corpus generated on the GPU at the reference index's size
        small = None
        if ref_docs == full["docs"]:
            ora, port = ora_full, full                           # the full-size index and timing of cpu_baseline()
        else:
            small = Database.synthetic(CORPUS_SEED, ref_docs, args.vocab, device=0, with_positions=args.op == "PHRASE")
            ora = H.DeviceOracle(small, [t for q in sample for t in q["terms"]], positions=args.op == "PHRASE")
            ora.warm()
            port = time_port(ora, sample, args.op, k, min(5.0, args.cpu_seconds), n_required)
        # parity of port and reference on the sampled queries, on this index
        ref_out = os.path.join(tmp, "ref_out.txt")
        H.xapian_ref("query", qfile, ref_out, dbdir)
        ref_res = H.parse_ref_output(ref_out)
        checked = 0
        if args.op != "PHRASE":                                  # (PHRASE top-k: the reference's stale-weight quirk)
            want = H.oracle_search_batch(ora, sample, 0, k)
            for (rows, _), rr in zip(want, ref_res):
                assert [(d, w) for d, w, _ in rows] == [(d, w) for d, w, _ in rr["hits"]], "port/reference parity failure on the reference index"
                checked += 1
        if small is not None:
            ora.close()
            small.close()
        return {"kind": "reference", "value": one["qps"], "unit": "queries/s", "cores": 1, "p50_ms": one["p50_us"] / 1e3,
                "all_cores": {"value": many["qps"], "unit": "queries/s", "cores": cores, "p50_ms": many["p50_us"] / 1e3},
                "docs": ref_docs, "index_build": binfo, "port_same_index": {kk: vv for kk, vv in port.items() if kk != "all_cores"},
                "port_over_reference": port["value"] / one["qps"], "port_vs_reference_parity_checked": checked}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(db, pool_q, args, k, timed_plans):
    """`value` is the REAL reference when oracle/_ref/xapian_ref is present (kind "reference": Enquire::get_mset of the
    vendored Xapian on a --ref-docs glass index built on this box), else the port (kind "port").  Always also: the
    port at the configuration's full size on the postings copied back from HBM, 1 thread and all cores, with the GPU
    answers checked against it on every sampled query."""
    import helpers as H
    from xapiand_amd import _lib
    L = _lib.lib()
    sample = pool_q[:128]
    n_required = args.required if args.op in H.SIDED else 0
    ora = H.DeviceOracle(db, [t for q in sample for t in q["terms"]], positions=args.op == "PHRASE")
    ora.warm()
    one_hits = (_lib.Hit * k)()
    one_hdr = _lib.ResultHdr()
    checked = [0]

    def checker(qi, hits):          # parity of the GPU answer on every sampled query, outside the timing
        _lib.check(L.xgm_search(db._h, C.byref(timed_plans[qi]), one_hits, C.byref(one_hdr)))
        got = [(one_hits[j].docid, one_hits[j].weight) for j in range(one_hdr.n_hits)]
        assert got == [(d, w) for d, w, _ in hits], "GPU/CPU parity failure on bench query %d" % qi
        checked[0] += 1
    db.set_stream(0)
    full = time_port(ora, sample, args.op, k, args.cpu_seconds, n_required, checker)
    full["all_cores"] = port_all_cores(ora, sample, args.op, k, min(5.0, args.cpu_seconds), n_required)
    full["docs"] = db.info().doccount
    sample_txt = ("first %d queries of the timed pool; port = oracle/xgm_oracle.cc (the reference's glass-chunk / MultiAnd / BM25 / "
                  "ProtoMSet algorithm restated) on the same %d-doc postings copied back from HBM, %.1f s of CPU work" % (len(sample), full["docs"], full["seconds"]))
    ref = None
    try:
        ref = reference_leg(args, sample, k, n_required, full, ora)
    except Exception as e:       # the reference leg is best effort (disk space, missing binary): say why it is absent
        ref = None
        sample_txt += "; reference leg failed: %r" % (e,)
    ora.close()
    if ref:
        out = dict(ref)
        out["port_full_size"] = full
        if ref["docs"] != full["docs"]:
            out["reference_full_size_estimate"] = {"value": full["value"] / ref["port_over_reference"], "unit": "queries/s", "cores": 1,
                                                   "note": "port at %d docs / port_over_reference" % full["docs"]}
            rpath = os.path.join(ROOT, "profiles", "r02_reference_full.json")
            if args.op == "AND" and args.terms == 3 and k == 10 and full["docs"] == 10_000_000 and os.path.exists(rpath):
                # the same leg run ONCE at the full size on this box type (too long for every default run)
                r = json.load(open(rpath))
                out["reference_full_size_measured"] = {"value": r["one_thread"]["value"], "unit": "queries/s", "cores": 1, "all_cores": r["all_cores"],
                                                       "docs": r["docs"], "source": "profiles/r02_reference_full.json (python bench.py --ref-docs 10000000)"}
        out["sample"] = ("Enquire::get_mset of the vendored Xapian (oracle/_ref/xapian_ref time), glass index of the first %d documents of the same "
                         "corpus built on this box (%.0f s on %d cores + %.0f s compact), same queries; " % (ref["docs"], ref["index_build"]["build_s"],
                                                                                                       ref["index_build"]["procs"], ref["index_build"]["compact_s"])) + sample_txt
        out["parity_checked_queries"] = checked[0]
        return out
    out = dict(full)
    out.update(kind="port", sample=sample_txt, parity_checked_queries=checked[0])
    return out


if __name__ == "__main__":
    main()
