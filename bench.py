#!/usr/bin/env python
"""Benchmark of the match/rank hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

Workload (N = 1, BASELINE.json configs[1], "C2"): 10 M-doc / 1 M-term Zipf synthetic index resident
in HBM, 3-term conjunctive BM25 queries, top-10.  One "step" = one pass of the hot path over one
batch of 256 planned queries (xgm_search_batch_device: decode → intersect → BM25 → top-k → merge).
N > 1 is configs[3] scaled ("C4"): the corpus has 10 M × N documents sharded N ways exactly like the
reference (global doc g → shard (g-1) % N, src/xapian/backends/multi.h:38-73), every rank searches
its shard with the MERGED collection statistics, then one RCCL all-gather of the per-shard top-k
records + a device-side merge (xgm_merge_shards_device).  Weak scaling: per-GPU work is fixed.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (algorithmic bytes
of the dominant kernel ÷ its HIP-event-measured duration vs 8 TB/s) and `cpu_baseline` (the CPU
oracle port of the reference algorithm timed on a bounded sample of the same queries, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from xapiand_amd import Database, Query, _lib  # noqa: E402
from xapiand_amd.distributed import ShardedSearcher  # noqa: E402

HBM_PEAK = 8.0e12          # B/s, /opt/skills/guides/MI355X_MICROARCH.md
CORPUS_SEED = 0x5EED0001
QUERY_SEED = 0x5EED0002
BATCH = 256


def gen_queries(n, n_terms, lo, hi, seed):
    import math
    import random
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        ranks = set()
        while len(ranks) < n_terms:
            ranks.add(max(1, int(round(math.exp(rng.uniform(math.log(lo), math.log(hi)))))))
        ranks = list(ranks)
        rng.shuffle(ranks)
        out.append(["t%d" % r for r in ranks])
    return out


def workload_name(args, world, n_docs_global, k):
    """BASELINE.json config the run corresponds to (C2 is the one the metric is quoted on)."""
    shape = {"AND": "%d-term conjunctive" % args.terms, "OR": "%d-term disjunctive" % args.terms, "PHRASE": "2-3-term phrase (positions)"}.get(
        args.op, "%s (%d-term AND, %d on the right)" % (args.op, args.required, args.terms - args.required))
    if world > 1:
        return "C4 (weak-scaled): %dM-doc index sharded %d ways, %s BM25 top-%d, RCCL top-k all-gather" % (n_docs_global // 1000000, world, shape, k)
    cfg = {"AND": "C2", "OR": "C3", "PHRASE": "C5"}.get(args.op, "next (SURVEY 8f.2)")
    return "%s: %dM-doc / %dM-term Zipf index, %s BM25 top-%d, 1 MI355X" % (cfg, args.docs_per_gpu // 1000000, args.vocab // 1000000, shape, k)


def main():
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
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-latency", action="store_true", help="skip the one-query-in-flight leg (profiling runs)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- index: this rank's shard, generated + inverted + block-encoded on the GPU ----------------
    n_docs_global = args.docs_per_gpu * world
    t0 = time.time()
    db = Database.synthetic(CORPUS_SEED, n_docs_global, args.vocab, n_shards=world, shard=rank, device=local_rank,
                            stripe_bits=args.stripe_bits)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    info = db.info()
    # One explicit (non-null) HIP stream for searches and collectives: xgm_search_batch_device is then
    # asynchronous, so the host plans batch i+1 while the GPU runs batch i (a null stream handle means
    # "the library's own stream, synchronised before returning").
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    db.set_stream(stream.cuda_stream)

    # ---- queries + merged statistics (Enquire::add_prepared_mset: Σ over shards): one all-reduce ----
    if args.op == "PHRASE":
        # C5: 2-3-grams that occur in a random document (tests/helpers.py restates the corpus in Python)
        import helpers as H
        pool = [q["terms"] for q in H.gen_phrase_queries(1100, n_docs_global, args.vocab, seed=QUERY_SEED)]
    else:
        pool = gen_queries(1100, args.terms, 8, 4096, QUERY_SEED)
    k = args.topk
    searcher = ShardedSearcher(db, rank, world, dev)
    sided = args.op in ("AND_NOT", "AND_MAYBE", "FILTER")
    plans = searcher.prepare([Query(args.op, terms, n_required=args.required if sided else 0) for terms in pool], 0, k)
    warm_plans, timed_plans = plans[:100], plans[100:]
    n_batches = len(timed_plans) // BATCH
    batches = [(_lib.Query * BATCH)(*timed_plans[i * BATCH:(i + 1) * BATCH]) for i in range(n_batches)]
    warm_batch = (_lib.Query * BATCH)(*(warm_plans * 3)[:BATCH])
    L = _lib.lib()
    state = {}

    def step(batch):
        # xgm_search_batch_device (+ RCCL all-gather of top-k + xgm_merge_shards_device when sharded)
        state["hits"], state["hdrs"] = searcher.run_batch(batch, BATCH, k)

    for _ in range(args.warmup):
        step(warm_batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    db.set_profiling(True)
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(batches[s % n_batches])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = db.last_kernel_ms()            # mean match-kernel duration over the timed steps (HIP events)
    kernel_name = db.last_kernel_name()
    db.set_profiling(False)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    qps = args.steps * BATCH / elapsed             # queries answered over the WHOLE (sharded) index per second

    # ---- algorithmic bytes per launch: Σ_t df_t·8 + S·4 + k·16 per query ----------
    alg_bytes = []
    for b in range(n_batches):
        step(batches[b])
        torch.cuda.synchronize()
        h = searcher._buffers(BATCH, k)["hdrs"].cpu().numpy().view(np.uint8).reshape(BATCH, 32)   # this shard's own header
        matches = h[:, 8:16].copy().view(np.uint64).reshape(BATCH)
        post = sum(L.xgm_query_postings_bytes(db._h, C.byref(p)) for p in timed_plans[b * BATCH:(b + 1) * BATCH])
        alg_bytes.append(post + int(matches.sum()) * 4 + BATCH * k * 16)
    used = [alg_bytes[s % n_batches] for s in range(args.steps)]
    bytes_per_launch = float(np.mean(used))
    achieved = bytes_per_launch / (kernel_ms * 1e-3) if kernel_ms and kernel_ms > 0 else 0.0

    # ---- latency mode: one query in flight, host-timed around xgm_search (incl. H2D/D2H) -----------
    lat = []
    if (rank == 0 or world > 1) and not args.no_latency:
        one_hits = (_lib.Hit * k)()
        one_hdr = _lib.ResultHdr()
        db.set_stream(0)
        for i, p in enumerate(timed_plans[:1000]):        # 1 000 queries (the first 20 warm up)
            a = time.perf_counter()
            _lib.check(L.xgm_search(db._h, C.byref(p), one_hits, C.byref(one_hdr)))
            if i >= 20:
                lat.append(time.perf_counter() - a)
    lat.sort()

    # ---- HBM traffic of the dominant kernel: from the committed PMC passes (tools/final.sh → profiles/traffic.json),
    # which cannot be collected from inside this process; only quoted when measured on this very workload
    traffic, traffic_note = None, None
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")
    if os.path.exists(tpath):
        for t in json.load(open(tpath)).get("entries", []):
            if (t["kernel"] == kernel_name and t["op"] == args.op and t["docs_per_gpu"] == args.docs_per_gpu and t["top_k"] == k and
                    t["terms"] == (args.terms if args.op != "PHRASE" else 0) and t["batch"] == BATCH):
                traffic, traffic_note = t["hbm_bytes_per_launch"], t["note"]

    result = None
    if rank == 0:
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
                       "corpus_seed": hex(CORPUS_SEED), "query_seed": hex(QUERY_SEED)},
            "p50_latency_us": lat[len(lat) // 2] * 1e6 if lat else None,
            "p99_latency_us": lat[int(len(lat) * 0.99)] * 1e6 if lat else None,
            "index": {"postings": info.n_postings, "blocks": info.n_blocks, "payload_bytes": info.payload_bytes,
                      "device_bytes": info.device_bytes, "build_seconds": build_s},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_note": traffic_note,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "kernel_ms": kernel_ms},
        }

    # ---- CPU baseline: the oracle port of the reference algorithm on a bounded sample --------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(db, pool[100:], args, k, timed_plans)
    if rank == 0:
        print(json.dumps(result), flush=True)
    db.close()
    if world > 1:
        dist.destroy_process_group()


def cpu_all_cores(shim, sample, op, k, seconds, n_required=0):
    """The same port on every host core at once: oracle/xgm_oracle.cc::xgo_search_many runs one C++ thread per
    core (queries striped across threads, the index is read-only once every list is built): aggregate queries/s."""
    import helpers as H
    n_threads = max(1, os.cpu_count() or 1)
    flat = [t.encode() for q in sample for t in q]
    n_terms = (C.c_uint32 * len(sample))(*[len(q) for q in sample])
    terms = (C.c_char_p * len(flat))(*flat)
    lens = (C.c_uint32 * len(flat))(*[len(t) for t in flat])
    done, wall = C.c_uint64(), C.c_double()
    ol = H.olib()
    ol.xgo_search_many.restype = C.c_int
    ol.xgo_search_many.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_char_p), C.POINTER(C.c_uint32),
                                   C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    opcode = H.OPS[op] | ((n_required or 1) << 8 if op in H.SIDED else 0)
    rc = ol.xgo_search_many(shim.oracle_index(), opcode, len(sample), n_terms, terms, lens, 0, k, n_threads, float(seconds),
                            C.byref(done), C.byref(wall))
    assert rc == 0
    return {"value": done.value / wall.value, "unit": "queries/s", "cores": n_threads, "seconds": wall.value}


def cpu_baseline(db, term_lists, args, k, timed_plans):
    """Times oracle/libxgm_oracle.so (glass-format varint chunks, MultiAnd leapfrog, doclen-list
    lookups, fp64 BM25, ProtoMSet heap — the reference's algorithm restated, 1 thread) on the first
    queries of the timed pool, over the SAME 10M-doc postings (copied back from HBM), and checks the
    GPU results against it."""
    import helpers as H
    L = _lib.lib()
    sample = term_lists[:128]
    terms = sorted({t.encode() for q in sample for t in q})
    info = db.info()
    doclen = np.zeros(info.lastdocid + 1, dtype=np.uint32)
    _lib.check(min(0, L.xgm_debug_read_doclen(db._h, doclen.ctypes.data_as(C.POINTER(C.c_uint32)), doclen.size)))
    dids, wdfs, dfs = [], [], []
    for t in terms:
        tid, tf = C.c_uint32(), C.c_uint32()
        _lib.check(L.xgm_lookup_term(db._h, t, len(t), C.byref(tid), C.byref(tf), None, None))
        d = np.zeros(tf.value, dtype=np.uint32)
        w = np.zeros(tf.value, dtype=np.uint32)
        if tf.value:
            n = L.xgm_debug_decode_term_device(db._h, tid.value, d.ctypes.data_as(C.POINTER(C.c_uint32)),
                                               w.ctypes.data_as(C.POINTER(C.c_uint32)), tf.value)
            assert n == tf.value, L.xgm_last_error()
        dids.append(d); wdfs.append(w); dfs.append(tf.value)
    keep = [i for i, n in enumerate(dfs) if n > 0]
    terms = [terms[i] for i in keep]
    did = np.concatenate([dids[i] for i in keep]) if keep else np.zeros(0, dtype=np.uint32)
    wdf = np.concatenate([wdfs[i] for i in keep]) if keep else np.zeros(0, dtype=np.uint32)
    df = np.array([dfs[i] for i in keep], dtype=np.uint32)
    tlen = np.array([len(t) for t in terms], dtype=np.uint32)
    tarr = (C.c_char_p * len(terms))(*terms)
    u32p = C.POINTER(C.c_uint32)
    ol = H.olib()
    oidx = ol.xgo_index_from_raw(len(terms), info.lastdocid, info.doccount, info.total_length, doclen.ctypes.data_as(u32p),
                                 C.cast(tarr, C.POINTER(C.c_char_p)), tlen.ctypes.data_as(u32p), df.ctypes.data_as(u32p),
                                 did.ctypes.data_as(u32p), wdf.ctypes.data_as(u32p), None, None)
    for t in terms:
        ol.xgo_index_warm(oidx, t, len(t))          # glass chunk encoding is index-build work: not timed

    class Shim:            # what helpers.oracle_search needs
        def oracle_index(self):
            return oidx
    shim = Shim()
    n_required = args.required if args.op in H.SIDED else 0
    done, spent, checked, passes = 0, 0.0, 0, 0
    one_hits = (_lib.Hit * k)()
    one_hdr = _lib.ResultHdr()
    lat = []
    while spent < args.cpu_seconds and passes < 50:
        for qi, q in enumerate(sample):
            a = time.perf_counter()
            hits, _ = H.oracle_search(shim, args.op, q, 0, k, n_required=n_required)
            dt = time.perf_counter() - a
            spent += dt
            lat.append(dt)
            done += 1
            if passes == 0:          # parity of the GPU answer on every sampled query, outside the timing
                _lib.check(L.xgm_search(db._h, C.byref(timed_plans[qi]), one_hits, C.byref(one_hdr)))
                got = [(one_hits[j].docid, one_hits[j].weight) for j in range(one_hdr.n_hits)]
                assert got == [(d, w) for d, w, _ in hits], "GPU/CPU parity failure on bench query %d" % qi
                checked += 1
        passes += 1
    many = cpu_all_cores(shim, sample, args.op, k, min(5.0, args.cpu_seconds), n_required)
    ol.xgo_index_free(oidx)
    lat.sort()
    return {"value": done / spent, "unit": "queries/s", "cores": 1, "kind": "port", "all_cores": many,
            "sample": "first %d queries of the timed pool x %d passes (same 10M-doc postings, copied back from HBM), %.1f s of CPU work; "
                      "oracle/xgm_oracle.cc = the reference's glass-chunk/MultiAnd/BM25/ProtoMSet algorithm, 1 thread" % (len(sample), passes, spent),
            "p50_ms": lat[len(lat) // 2] * 1e3, "parity_checked_queries": checked}


if __name__ == "__main__":
    main()
