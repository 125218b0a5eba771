#!/usr/bin/env python
"""Benchmark of the match/rank hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1 needs one process per GPU: when started as a plain `python bench.py --gpus N` the script re-executes
itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (the form
the driver uses directly), so both invocations work.

Workload (N = 1, BASELINE.json configs[1], "C2"): 10 M-doc / 1 M-term Zipf synthetic index resident in HBM,
3-term conjunctive BM25 queries, top-10.  One "step" = one pass of the hot path over --batches-per-step (16) batches of
256 queries each (4 096 queries; 20 steps ≈ 130 ms of GPU work) AS THE MATCHER HOOK RECEIVES THEM (terms, operator,
first/maxitems, BM25 parameters, merged statistics): xgm_get_mset_batch_begin = dictionary lookups + BM25Weight::init +
leaf ordering (xgm_plan_query) → decode → intersect → BM25 → top-k → merge → the hits and headers of EVERY batch copied
to pinned HOST memory (xgm_batch_end hands them out); two batches are in flight (--in-flight), so planning, match and
download overlap (round 4: rounds 1-3 left the results in HBM).  The default run also times C3 (5-term OR, top-100) and
C5 (2-3-term PHRASE, top-10) the same way (`other_configs`), each checked against the oracle on 128 queries.  N > 1 is configs[3] scaled ("C4"): the corpus has 10 M × N documents sharded
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
BATCH = int(os.environ.get("XGM_BENCH_BATCH", "256"))          # (diagnostics only: the metric is quoted on batches of 256)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batches-per-step", type=int, default=16, help="batches of 256 queries one step submits")
    ap.add_argument("--in-flight", type=int, default=2, help="batches in flight (xgm_get_mset_batch_begin ... xgm_batch_end); measured on the MI355X, round 5: 2 beats 3 by 4-7 %% on C2 (the host needs 0.15 ms per batch, the kernel 0.35: one batch queued behind the running one is enough, a third only adds queue traffic)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C3 / C5 sub-legs of the default (C2) line")
    ap.add_argument("--no-hook-parity", action="store_true", help="skip the hook-on == hook-off leg on the reference's own glass index")
    ap.add_argument("--ref-no-positions", action="store_true", help="build the reference index of the headline run WITHOUT positions (faster; C5's reference baseline and hook-parity leg are then skipped)")
    ap.add_argument("--ref-seconds", type=float, default=12.0, help="time box of the reference's all-core leg")
    ap.add_argument("--ref-build-budget", type=float, default=1000.0, help="seconds the reference's index build may take (normally ~520 on 128 cores) before the leg is given up and the port reported (0 = no limit)")
    ap.add_argument("--docs-per-gpu", type=int, default=10_000_000)
    ap.add_argument("--vocab", type=int, default=1_000_000)
    ap.add_argument("--op", default="AND")
    ap.add_argument("--terms", type=int, default=3)
    ap.add_argument("--required", type=int, default=1, help="AND_NOT / AND_MAYBE / FILTER: terms of the left-hand AND")
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--replay", choices=("none", "frozen", "count", "frozen+count"), default="none",
                    help="frozen (PHRASE): every query carries XGM_REPLAY_BATCH_FROZEN — the REFERENCE's own page (SelectPostList's frozen weight) at batch throughput; "
                         "count: XGM_REPLAY_BATCH_COUNT — ProtoMSet's known_matching_docs (the exact HTTP total) with every page")
    ap.add_argument("--stripe-bits", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--ref-docs-per-shard", type=int, default=1_000_000, help="N > 1: documents per shard of the reference's shard indexes (the CPU baseline of C4: a bounded sample)")
    ap.add_argument("--ref-docs", type=int, default=-1, help="documents of the reference glass index built on this box (0: skip the reference leg; "
                    "default: the configuration's own size for the headline workload — indexing 10 M documents through the reference's "
                    "WritableDatabase takes ~3.5 min on the 256-core host —, 1/5 of it for the other operators)")
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
        import torch.distributed as dist
        exchange = "RCCL top-k all-gather" if dist.is_initialized() and dist.get_backend() == "nccl" else "gloo top-k all-gather staged through host memory (functional run: not RCCL)"
        return "C4 (weak-scaled): %dM-doc index sharded %d ways, %s BM25 top-%d, %s" % (n_docs_global // 1000000, world, shape, k, exchange)
    cfg = {"AND": "C2", "OR": "C3", "PHRASE": "C5"}.get(args.op, "next (SURVEY 8f.2)")
    return "%s: %dM-doc / %dM-term Zipf index, %s BM25 top-%d, 1 MI355X" % (cfg, args.docs_per_gpu // 1000000, args.vocab // 1000000, shape, k)


class Leg:
    """One workload on the resident index: its query pool as the hook would receive it, cut into batches."""

    def __init__(self, searcher, op, terms, required, k, n_docs_global, vocab, n_batches, replay=0):
        import helpers as H
        from xapiand_amd import Query, _lib
        self.op, self.terms, self.required, self.k, self.replay = op, terms, required, k, replay
        self.sided = op in ("AND_NOT", "AND_MAYBE", "FILTER")
        self.pool = H.bench_pool(op, terms, required, n_docs_global, vocab, n=100 + n_batches * BATCH, seed=QUERY_SEED, maxitems=k)
        qobjs = [Query(q["op"], q["terms"], n_required=required if self.sided else 0) for q in self.pool]
        self.descs, self.gstats = searcher.describe(qobjs, 0, k)       # what the hook is handed per get_mset
        self.plans = searcher.prepare(qobjs, 0, k)                     # bookkeeping only (algorithmic bytes, parity leg)
        if replay:                                                     # XGM_REPLAY_BATCH_* bits: the reference's own collation inside the batch
            for i in range(len(self.pool)):
                self.descs[i].replay = replay
                self.plans[i].replay = replay
        self._keep = searcher._keep                                   # (the descriptions' term bytes)
        n = len(self.pool)

        def batch_of(lo):
            d = (_lib.QueryDesc * BATCH)(*[self.descs[(lo + i) % n] for i in range(BATCH)])
            g = (_lib.GlobalStats * BATCH)(*[self.gstats[(lo + i) % n] for i in range(BATCH)])
            return d, g
        self.batches = [batch_of(100 + b * BATCH) for b in range(n_batches)]
        self.warm = batch_of(0)
        self.timed_plans = self.plans[100:]
        self.timed_pool = self.pool[100:]


def run_steps(db, searcher, leg, steps, bps, depth, world, keep_last=False):
    """`steps` steps of `bps` batches each.  N = 1: through xgm_get_mset_batch_begin / xgm_batch_end with up to `depth` batches in flight,
    every batch's hits and headers delivered to pinned host memory and touched there.  N > 1: the sharded protocol per batch
    (ShardedSearcher.run_descs: search, RCCL all-gather, device merge) and an asynchronous copy of the merged result to pinned host
    memory, double-buffered.  Returns (host seconds spent inside the calls, rows delivered, bytes of the last batch on the host)."""
    import collections
    import torch
    from xapiand_amd import _lib
    L = _lib.lib()
    k, nb = leg.k, len(leg.batches)
    host_s, delivered, last = 0.0, 0, None
    if world == 1:
        inflight = collections.deque()
        hp, dp = C.POINTER(_lib.Hit)(), C.POINTER(_lib.ResultHdr)()

        def finish(f, keep):
            _lib.check(L.xgm_batch_end(f, C.byref(hp), C.byref(dp)))
            n = dp[0].n_hits + dp[BATCH - 1].n_hits                   # the rows are host memory: read them
            snap = None
            if keep:
                snap = (C.string_at(hp, BATCH * k * 16), C.string_at(dp, BATCH * 32))
            L.xgm_batch_release(f)
            return n, snap
        for s in range(steps):
            for j in range(bps):
                bi = (s * bps + j) % nb
                d, g = leg.batches[bi]
                h0 = time.perf_counter()
                f = C.c_void_p()
                _lib.check(L.xgm_get_mset_batch_begin(db._h, d, g, BATCH, k, C.byref(f)))
                host_s += time.perf_counter() - h0
                inflight.append((f, bi))
                if len(inflight) >= depth:
                    f0, b0 = inflight.popleft()
                    n, snap = finish(f0, keep_last and b0 == 0 and s == steps - 1)      # (batch 0 of the LAST step: the oracle answers its first 128 queries)
                    delivered += n
                    last = snap or last
        while inflight:
            f0, b0 = inflight.popleft()
            n, snap = finish(f0, keep_last and b0 == 0)
            delivered += n
            last = snap or last
        return host_s, delivered, last
    # sharded: results of batch i on the host while batch i + 1 runs
    slots = getattr(searcher, "_host_slots", None)
    if slots is None:
        slots = searcher._host_slots = [dict(hits=torch.empty((BATCH, k, 2), dtype=torch.float64).pin_memory(), hdrs=torch.empty((BATCH, 4), dtype=torch.float64).pin_memory(),
                                             ev=torch.cuda.Event()) for _ in range(2)]
    pending = [False, False]
    i = 0
    for s in range(steps):
        for j in range(bps):
            d, g = leg.batches[(s * bps + j) % nb]
            sl = slots[i & 1]
            if pending[i & 1]:
                sl["ev"].synchronize()
                delivered += int(sl["hdrs"][0, 0].item() != -1.0)
            h0 = time.perf_counter()
            oh, od = searcher.run_descs(d, g, BATCH, k, slot=i & 1)        # (two sets of device buffers: batch i + 1 is enqueued while batch i's copy is in flight)
            sl["hits"].copy_(oh, non_blocking=True)
            sl["hdrs"].copy_(od, non_blocking=True)
            sl["ev"].record()
            host_s += time.perf_counter() - h0
            pending[i & 1] = True
            i += 1
    for b in range(2):
        if pending[b]:
            slots[b]["ev"].synchronize()
    last = None
    if keep_last and i:
        # the LAST timed batch as it reached the host: (pool batch index, merged hits [BATCH][k][2] f64 views of xgm_hit, headers)
        sl = slots[(i - 1) & 1]
        last = ((steps * bps - 1) % nb, sl["hits"].clone(), sl["hdrs"].clone())
    return host_s, delivered, last


def measure(db, searcher, leg, args, world, rank, dev, steps, warmup):
    """Timed region + per-launch byte counts of one workload.  Returns a dict (rank 0 uses it; every rank takes part in the collectives)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from xapiand_amd import _lib
    L = _lib.lib()
    k, bps, depth = leg.k, args.batches_per_step, max(1, args.in_flight)
    # the batches' kernels run back to back on ONE stream, uploads and downloads on the in-flight batches' own streams (measured, round 4:
    # 603 k queries/s; with every batch on a stream of its own the kernels overlap and slow each other down: 539 k — XGM_BENCH_MULTI_STREAM=1)
    bind_one_stream = os.environ.get("XGM_BENCH_MULTI_STREAM") is None
    if world == 1:
        db.set_stream(torch.cuda.current_stream(dev).cuda_stream if bind_one_stream else 0)
    # (this script's own garbage: by the time the sub-legs run the process holds millions of Python objects — query pools, oracle answers — and one
    #  generation-2 collection inside a timed region of ~100 ms costs tens of ms: measured, the C5 sub-leg of the default line ran at 283 k
    #  queries/s where the same leg alone gives 466 k.  The collector is off from the warm-up to the end of the timed region; no product code is Python)
    import gc
    gc.collect()
    gc.disable()
    run_steps(db, searcher, leg, warmup, bps, depth, world)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    db.set_profiling(1)
    L.xgm_debug_host_ns((C.c_uint64 * 8)())           # reset the host-side section timers
    t0 = time.perf_counter()
    host_s, delivered, last = run_steps(db, searcher, leg, steps, bps, depth, world, keep_last=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    hn = (C.c_uint64 * 8)()
    L.xgm_debug_host_ns(hn)
    n_launch = max(1, int(hn[3]))
    n_batches_run = steps * bps
    host_sections = {"plan_queries": round(hn[0] / 1e3 / n_batches_run, 1), "plan_batch": round(hn[1] / 1e3 / n_launch, 1),
                     "stage_enqueue_launch": round(hn[2] / 1e3 / n_launch, 1), "staging_memcpy": round(hn[4] / 1e3 / n_launch, 1),
                     "upload_enqueue": round(hn[5] / 1e3 / n_launch, 1), "match_launch": round(hn[6] / 1e3 / n_launch, 1),
                     "merge_launch": round(hn[7] / 1e3 / n_launch, 1), "launches_per_batch": round(n_launch / n_batches_run, 2), "unit": "us per batch"}
    kernel_ms = db.last_kernel_ms()            # mean match-kernel duration over the timed launches (HIP events on the launch streams)
    kernel_name = db.last_kernel_name()
    db.set_profiling(0)
    cdev = dev if (world == 1 or dist.get_backend() == "nccl") else torch.device("cpu")      # where the small collectives of this script live
    t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # the last delivered batch against a synchronous search of the same batch: the rows on the host are the answer
    verified = None
    if world == 1 and last is not None:
        d, g = leg.batches[0]                                     # (the snapshot is batch 0 of the last timed step)
        hits = (_lib.Hit * (BATCH * k))()
        hdrs = (_lib.ResultHdr * BATCH)()
        _lib.check(L.xgm_get_mset_batch(db._h, d, g, BATCH, k, hits, hdrs))
        import struct
        hb, db_ = bytes(memoryview(hits)), bytes(memoryview(hdrs))
        verified = True
        for q in range(BATCH):
            n = hdrs[q].n_hits
            verified = verified and hb[q * k * 16:(q * k + n) * 16] == last[0][q * k * 16:(q * k + n) * 16]
            a, b = struct.unpack_from("<IIQdd", db_, q * 32), struct.unpack_from("<IIQdd", last[1], q * 32)
            # (a positional query that prunes by weight reports a LOWER BOUND of its match count — how many candidates were tested depends
            #  on when the query-wide threshold rose: timing, not semantics; include/xgm.h)
            lower = (a[2] >> 63) or (b[2] >> 63)
            verified = verified and a[:2] == b[:2] and a[3:] == b[3:] and (lower or a[2] == b[2])
    # ---- kernel duration WITHOUT a neighbour: one batch in flight (the timed region overlaps consecutive batches on the chip) ----
    kernel_ms_solo = None
    if world == 1 and depth > 1:
        db.set_profiling(1)
        run_steps(db, searcher, leg, max(1, min(4, steps)), bps, 1, world)
        torch.cuda.synchronize()
        kernel_ms_solo = db.last_kernel_ms()
        db.set_profiling(0)

    # ---- per-launch byte counts, outside the timed region ------------------------------------------------------
    # algorithmic: Σ_t df_t·8 + S·4 + P·4 + k·16 per query; model: the kernel's own request tallies
    db.set_profiling(0 if os.environ.get("XGM_BENCH_NO_TALLY") else 2)   # the tallying instantiation of the wave kernels
    if world == 1:
        db.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    alg_bytes, model_sector, model_useful, tallies = [], [], [], []
    for b in range(min(len(leg.batches), 4)):
        if leg.replay and world == 1:
            # (the replay bits are honoured by the host-delivering entry points: the tallied launch must be the one that was timed)
            th_, td_ = (_lib.Hit * (BATCH * k))(), (_lib.ResultHdr * BATCH)()
            _lib.check(L.xgm_get_mset_batch(db._h, leg.batches[b][0], leg.batches[b][1], BATCH, k, th_, td_))
            matches = np.array([td_[q].matches_exact & ((1 << 63) - 1) for q in range(BATCH)], dtype=np.uint64)
        else:
            searcher.run_descs(leg.batches[b][0], leg.batches[b][1], BATCH, k)
            torch.cuda.synchronize()
            h = searcher._buffers(BATCH, k)["hdrs"].cpu().numpy().view(np.uint8).reshape(BATCH, 32)   # this shard's own header
            matches = h[:, 8:16].copy().view(np.uint64).reshape(BATCH) & np.uint64((1 << 63) - 1)      # (bit 63: lower bound only, include/xgm.h)
        post = sum(L.xgm_query_postings_bytes(db._h, C.byref(p)) for p in leg.timed_plans[b * BATCH:(b + 1) * BATCH])
        tl = (C.c_uint64 * 10)()
        have_tally = L.xgm_last_batch_traffic(db._h, tl, 10) == 0 and kernel_name in ("xgm_andw_kernel", "xgm_orw_kernel", "xgm_orw2_kernel", "xgm_dense_kernel", "xgm_andw_list_kernel")
        bmpw, probes, blkw, hdrs_, dls, aux, cands, npos, probes_raw, dls_raw = [int(x) for x in tl]
        alg_bytes.append(post + int(matches.sum()) * 4 + npos * 4 + BATCH * k * 16)
        if have_tally:
            stream_b = 4 * bmpw + 4 * blkw + 12 * hdrs_ + 4 * aux + 16 * cands + 4 * npos
            model_sector.append(stream_b + SECTOR * (probes + dls))
            model_useful.append(stream_b + probes_raw + 4 * dls_raw)
            tallies.append(dict(bitmap_words=bmpw, probe_sectors=probes, probes=probes_raw, payload_words=blkw, block_headers=hdrs_,
                                doclen_sectors=dls, doclen_gathers=dls_raw, aux_words=aux, candidates_out=cands, positions=npos))
    db.set_profiling(0)
    bytes_per_launch = float(np.mean(alg_bytes))
    model_bytes = float(np.mean(model_sector)) if model_sector else None
    useful_bytes = float(np.mean(model_useful)) if model_useful else None

    # ---- HBM traffic of the dominant kernel: from the committed PMC passes (tools/final.sh → profiles/traffic.json),
    # which cannot be collected from inside this process; only quoted when measured on this very workload AND this very build
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
            if (tr["kernel"] == kernel_name and tr["op"] == leg.op and tr["docs_per_gpu"] == args.docs_per_gpu and tr["top_k"] == k and
                    tr["terms"] == (leg.terms if leg.op != "PHRASE" else 0) and tr["batch"] == BATCH and tr.get("required", 1) == (leg.required if leg.sided else 1) and
                    tr.get("replay_bits", 0) == leg.replay):
                traffic, traffic_note = tr["hbm_bytes_per_launch"], tr["note"]

    # ---- N > 1: every rank's own match kernel against its own GPU's roofline, with ITS OWN bytes (the shard's own tallies) ------
    per_rank = None
    if world > 1:
        mine = torch.tensor([kernel_ms if kernel_ms and kernel_ms > 0 else 0.0, float(model_bytes or bytes_per_launch)], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = []
        for r, v in enumerate(allr):
            ms, by = float(v[0].item()), float(v[1].item())
            per_rank.append({"rank": r, "kernel_ms": ms, "bytes_per_launch": by, "basis": "model (this rank's own request tallies)",
                             "achieved": (by / (ms * 1e-3) / 1e9) if ms > 0 else None, "frac": (by / (ms * 1e-3) / HBM_PEAK) if ms > 0 else None})

    n_queries = steps * bps * BATCH
    kt = kernel_ms * 1e-3 if kernel_ms and kernel_ms > 0 else None
    basis = "pmc" if traffic else ("model" if model_bytes else "algorithmic")
    moved = traffic if traffic else (model_bytes if model_bytes else bytes_per_launch)
    achieved = moved / kt if kt else 0.0
    alg_rate = bytes_per_launch / kt if kt else 0.0
    kts = kernel_ms_solo * 1e-3 if kernel_ms_solo and kernel_ms_solo > 0 else None
    roofline = {"bound": "hbm", "kernel": kernel_name, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK, "basis": basis, "traffic": traffic, "traffic_note": traffic_note,
                "traffic_lib_sha": lib_sha if traffic else None, "traffic_entries_of_another_build_ignored": stale_traffic,
                "kernel_ms": kernel_ms,
                "kernel_ms_note": "mean HIP-event duration of the match kernel over the timed region (up to %d batches in flight; their match kernels run "
                                  "back to back on one stream, uploads and downloads on streams of their own); kernel_ms_one_in_flight: the same with "
                                  "one batch in flight" % depth if world == 1 else None,
                "kernel_ms_one_in_flight": kernel_ms_solo,
                "frac_one_in_flight": (moved / kts / HBM_PEAK) if kts else None,
                "whole_timed_region": {"bytes": moved * n_batches_run, "seconds": elapsed, "achieved": moved * n_batches_run / elapsed / 1e9,
                                       "frac": moved * n_batches_run / elapsed / HBM_PEAK,
                                       "note": "every byte the match launches moved / wall time of the timed region (planning, uploads, downloads, gaps included)"},
                "model_min_bytes": model_bytes, "model_useful_bytes": useful_bytes,
                "model_frac": (model_bytes / kt / HBM_PEAK) if (model_bytes and kt) else None,
                "model_note": "requests tallied by the kernel itself (xgm_last_batch_traffic): streamed words at 4 B + %d B per DISTINCT "
                              "memory sector a round of one-byte container probes / doclen gathers touches" % SECTOR,
                "model_counts": tallies[0] if tallies else None,
                "algorithmic": {"bytes_per_launch": bytes_per_launch, "achieved": alg_rate / 1e9, "frac": alg_rate / HBM_PEAK,
                                "note": "SURVEY 8(d): sum df*8 + S*4 + P*4 + k*16 per query; not bytes this design reads"}}
    if per_rank:
        ok = [p for p in per_rank if p["achieved"]]
        roofline["per_rank"] = per_rank
        roofline["aggregate"] = {"achieved": sum(p["achieved"] for p in ok), "peak": HBM_PEAK / 1e9 * world, "unit": "GB/s",
                                 "frac": (sum(p["frac"] for p in ok) / len(ok)) if ok else None,
                                 "note": "sum over the ranks of bytes moved per launch / that rank's kernel time; frac = mean of the per-rank fractions"}
    return {"value": n_queries / elapsed, "unit": "queries/s", "ms_per_step": elapsed / steps * 1e3, "ms_per_batch": elapsed / n_batches_run * 1e3,
            "queries_per_step": bps * BATCH, "steps": steps, "elapsed_s": elapsed,
            "host_ms_per_batch": round(1e3 * host_s / n_batches_run, 4), "host_us_per_batch": host_sections,
            "hits_delivered_to_host": True, "rows_read_on_host": delivered, "last_batch_on_host_equals_synchronous_search": verified,
            "_snapshot": last, "roofline": roofline}


def snapshot_rows(snap, k, n):
    """Rows 0..n of a delivered batch (the bytes xgm_batch_end handed out inside the timed region) → [(hits, (n_hits, subqs, matches, max_attained))]."""
    import struct
    out = []
    for q in range(n):
        nh, ms, mx, ma, _mp = struct.unpack_from("<IIQdd", snap[1], q * 32)
        hits = [struct.unpack_from("<IId", snap[0], (q * k + j) * 16) for j in range(nh)]
        out.append(([(d, w) for d, _, w in hits], (nh, ms, mx, ma)))
    return out


def parity_vs_port(db, leg, n=128, snapshot=None, ora=None):
    """The GPU's answers to the first n queries of the timed pool against the oracle port run on the very postings the device holds
    (copied back from HBM): docids, weight bit patterns, match counts — through xgm_search_batch AND, when `snapshot` is given, the very
    rows the TIMED entry point (xgm_get_mset_batch_begin ... xgm_batch_end) delivered to the host for batch 0 of the last timed step.
    Returns (queries checked, DeviceOracle, rows of the timed batch checked) — outside any timing."""
    import helpers as H
    from xapiand_amd import _lib
    L = _lib.lib()
    sample = leg.timed_pool[:n]
    if ora is None:
        ora = H.DeviceOracle(db, [t for q in sample for t in q["terms"]], positions=leg.op == "PHRASE")
        ora.warm()
    want = H.oracle_search_batch(ora, sample, 0, leg.k, reference_select_bug=bool(leg.replay))      # (replay bits: the oracle in reference mode — SelectPostList's frozen weight restated)
    db.set_stream(0)
    k = leg.k
    qs = (_lib.Query * n)(*leg.timed_plans[:n])
    hits = (_lib.Hit * (n * k))()
    hdrs = (_lib.ResultHdr * n)()
    _lib.check(L.xgm_search_batch(db._h, qs, n, k, hits, hdrs))
    for qi, (rows, oh) in enumerate(want):
        got = [(hits[qi * k + j].docid, hits[qi * k + j].weight) for j in range(hdrs[qi].n_hits)]
        assert got == [(d, w) for d, w, _ in rows], "GPU/CPU parity failure on %s bench query %d" % (leg.op, qi)
        if not leg.replay:           # (the oracle's reference mode reports what ITS walk counted before SelectPostList shut the loop, not the match count)
            H.check_matches(hdrs[qi].matches_exact, oh["matches"], len(got), (leg.op, qi))
    timed_rows = 0
    if snapshot is not None:
        for qi, ((rows, oh), (got, hd)) in enumerate(zip(want, snapshot_rows(snapshot, k, min(n, BATCH)))):
            assert got == [(d, w) for d, w, _ in rows], "timed batch / oracle parity failure on %s bench query %d" % (leg.op, qi)
            if not leg.replay:
                H.check_matches(hd[2], oh["matches"], len(got), (leg.op, qi, "timed batch"))
            timed_rows += 1
    return len(want), ora, timed_rows


def byte_compatible_leg(db, leg, ora, n=256, n_check=128):
    """The byte-compatible modes as the MATCHER HOOK issues them — one query per call, one call per query: xgm_search_batch_known(nq = 1) with
    the plan's XGM_REPLAY_BATCH_* bits (round 6; round 5: xgm_search, then xgm_search_replay behind it):
      PHRASE  FROZEN | COUNT: the page, weights and known_matching_docs of the reference incl. SelectPostList's frozen weight
              (selectpostlist.cc:28-55), listed and replayed on the device — checked against the oracle's reference mode (pinned to the compiled reference);
      OR      COUNT: known_matching_docs behind the HTTP total (protomset.h:340-400), counted by xgm_search_replay when the call is collected —
              checked against the host restatement (xgm_known_matching_docs, pinned to the compiled reference) over xgm_search_all's list on a few queries.
    One query in flight, then 8 host threads.  Returns a dict with queries/s."""
    import helpers as H
    from xapiand_amd import _lib
    L = _lib.lib()
    k = leg.k
    positional = leg.op == "PHRASE"
    bits = (_lib.XGM_REPLAY_BATCH_FROZEN | _lib.XGM_REPLAY_BATCH_COUNT) if positional else _lib.XGM_REPLAY_BATCH_COUNT
    db.set_stream(0)
    n = min(n, len(leg.timed_plans))
    plans = (_lib.Query * n)()
    for i in range(n):
        C.memmove(C.byref(plans[i]), C.byref(leg.timed_plans[i]), C.sizeof(_lib.Query))
        plans[i].replay = bits
    hits = (_lib.Hit * k)()
    hdr = _lib.ResultHdr()
    known = C.c_uint64()
    lat, answers, full_pages = [], [], 0
    t0 = time.perf_counter()
    for i in range(n):
        a = time.perf_counter()
        _lib.check(L.xgm_search_batch_known(db._h, C.byref(plans[i]), 1, k, hits, C.byref(hdr), C.byref(known)))
        lat.append(time.perf_counter() - a)
        full_pages += hdr.n_hits == k
        answers.append(([(hits[j].docid, hits[j].weight) for j in range(hdr.n_hits)], known.value))
    wall = time.perf_counter() - t0
    lat.sort()
    out = {"value": n / wall, "unit": "queries/s", "queries": n, "in_flight": 1, "p50_us": lat[len(lat) // 2] * 1e6, "p99_us": lat[int(len(lat) * 0.99)] * 1e6,
           "full_pages": full_pages,
           "what": ("xgm_search_batch_known(nq = 1, FROZEN | COUNT) per query: the reference's own top-%d incl. its frozen weight + its known_matching_docs" % k) if positional else
                   ("xgm_search_batch_known(nq = 1, COUNT) per query: the page + the reference's known_matching_docs (exact HTTP total)")}
    # the same calls from 8 host threads at once (Xapiand calls get_mset from every HTTP worker thread), with the index's dispatcher on: the
    # single-query calls meet in shared launches; rows compared with the one-in-flight answers above
    import threading
    n_thr = 8
    errors = []

    def worker(t):
        try:
            h_, r_, kn_ = (_lib.Hit * k)(), _lib.ResultHdr(), C.c_uint64()
            for i in range(t, n, n_thr):
                _lib.check(L.xgm_search_batch_known(db._h, C.byref(plans[i]), 1, k, h_, C.byref(r_), C.byref(kn_)))
                if ([(h_[j].docid, h_[j].weight) for j in range(r_.n_hits)], kn_.value) != answers[i]:
                    errors.append(i)
        except Exception as e:            # noqa: BLE001  (reported below: a thread must not die silently)
            errors.append(repr(e))

    _lib.check(L.xgm_index_set_batching(db._h, 64))
    thr = [threading.Thread(target=worker, args=(t,)) for t in range(n_thr)]
    t0 = time.perf_counter()
    for t in thr:
        t.start()
    for t in thr:
        t.join()
    wall_c = time.perf_counter() - t0
    _lib.check(L.xgm_index_set_batching(db._h, 0))
    assert not errors, "concurrent byte-compatible searches differ from the sequential ones: %r" % errors[:4]
    out["concurrent"] = {"value": n / wall_c, "unit": "queries/s", "host_threads": n_thr, "queries": n, "dispatcher": "xgm_index_set_batching(64)",
                         "rows_equal_to_one_in_flight": True, "note": "Python threads: the GIL bounds this figure; hook_parity's threaded legs run native threads"}
    if positional:
        sample = leg.timed_pool[:min(n_check, n)]
        want = H.oracle_search_batch(ora, sample, 0, k, reference_select_bug=True)
        for qi, (rows, _) in enumerate(want):
            assert answers[qi][0] == [(d, w) for d, w, _ in rows], "reference-mode parity failure on C5 bench query %d" % qi
        out["parity_checked_queries"] = len(want)
        out["parity_against"] = "oracle/xgm_oracle.cc in reference mode (SelectPostList's stale cached weight restated; pinned to the compiled reference)"
    else:
        L.xgm_known_matching_docs.restype = C.c_uint64
        L.xgm_known_matching_docs.argtypes = [C.POINTER(C.c_double), C.c_uint64, C.c_uint32, C.c_uint32]
        checked = 0
        hdr2 = _lib.ResultHdr()
        for qi in range(min(6, n)):
            p = leg.timed_plans[qi]
            cap = max(1, p.est_max)
            allh = (_lib.Hit * cap)()
            nm = C.c_uint64()
            _lib.check(L.xgm_search_all(db._h, C.byref(p), allh, cap, C.byref(nm), C.byref(hdr2)))
            w = (C.c_double * nm.value)(*[allh[j].weight for j in range(nm.value)])
            assert L.xgm_known_matching_docs(w, nm.value, k, p.check_at_least) == answers[qi][1], "known_matching_docs differs on %s bench query %d" % (leg.op, qi)
            checked += 1
        out["known_matching_docs_checked_queries"] = checked
    return out


def latency_leg(db, leg, n_timed):
    """One query in flight: host-timed around plan + search incl. H2D / D2H (1 000 queries, the first 20 warm up)."""
    from xapiand_amd import _lib
    L = _lib.lib()
    k = leg.k
    one_hits = (_lib.Hit * k)()
    one_hdr = _lib.ResultHdr()
    db.set_stream(0)
    lat = []
    for i in range(min(1000, n_timed)):
        d1 = (_lib.QueryDesc * 1)(leg.descs[100 + i])
        g1 = (_lib.GlobalStats * 1)(leg.gstats[100 + i])
        a = time.perf_counter()
        _lib.check(L.xgm_get_mset_batch(db._h, d1, g1, 1, k, one_hits, C.byref(one_hdr)))
        if i >= 20:
            lat.append(time.perf_counter() - a)
    lat.sort()
    return lat


def sharded_parity(db, leg, world, rank, k, last, n=32):
    """N > 1: the first n queries of the LAST TIMED batch — the merged hits as they reached the host inside the timed region — against Xapiand's
    per-shard protocol run on the ORACLE: every rank answers its own shard (the postings copied back from its GPU) with the MERGED statistics
    (Enquire::add_prepared_mset, api/enquire.cc:385-394), the per-shard top k are gathered, docids unsharded ((local - 1) * n + shard + 1,
    mset.cc:367-373) and merged by (weight desc, docid asc) (Matcher::merge_mset, matcher.cc:653-781).  Returns the queries checked (rank 0)."""
    import struct
    import torch.distributed as dist
    import helpers as H
    b_last, hits_t, hdrs_t = last
    idx0 = 100 + b_last * BATCH
    sample = [leg.pool[(idx0 + i) % len(leg.pool)] for i in range(n)]
    ora = H.DeviceOracle(db, [t for q in sample for t in q["terms"]])
    mine = []
    for i, q in enumerate(sample):
        g = leg.gstats[(idx0 + i) % len(leg.pool)]
        gs = dict(total_length=g.total_length, collection_size=g.collection_size, has_positions=bool(g.full_db_has_positions),
                  termfreq=[g.termfreq[t] for t in range(len(q["terms"]))])
        rows, _ = H.oracle_search(ora, q["op"], q["terms"], 0, k, global_stats=gs)
        mine.append([(d, w) for d, w, _ in rows])
    ora.close()
    allr = [None] * world
    dist.all_gather_object(allr, mine)
    if rank != 0:
        return 0
    raw = hits_t.numpy().tobytes()
    hraw = hdrs_t.numpy().tobytes()
    for qi in range(n):
        want = sorted(((d - 1) * world + s + 1, w) for s in range(world) for d, w in allr[s][qi])
        want = sorted(want, key=lambda r: (-r[1], r[0]))[:k]
        nh = struct.unpack_from("<I", hraw, qi * 32)[0]
        got = [struct.unpack_from("<IId", raw, (qi * k + j) * 16) for j in range(nh)]
        assert [(d, w) for d, _, w in got] == want, "sharded parity failure on query %d of the last timed batch: %r vs %r" % (qi, got[:3], want[:3])
    return n


def sharded_cpu_baseline(args, leg, world, k):
    """N > 1 (C4): the REAL reference running Xapiand's per-shard protocol (prepare_mset on every shard, add_prepared_mset, get_mset per shard,
    unshard_docids, merge_mset — src/database/handler.cc:1532-1549; oracle/ref_build/ref_driver.cc run_query over several Database handles) on
    `world` shard indexes built on this box's host cores.  A BOUNDED sample: --ref-docs-per-shard documents per shard (default 1 M; the GPUs hold
    --docs-per-gpu each), the same round-robin sharding, the same queries; 1 thread and all cores."""
    import shutil
    import tempfile
    import helpers as H
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ref_index
    if not H.have_xapian_ref():
        return {"error": "oracle/_ref/xapian_ref is not built"}
    per_shard = args.ref_docs_per_shard
    tmp = tempfile.mkdtemp(prefix="xgm_ref_sh_", dir="/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None)
    try:
        t0 = time.time()
        dirs = []
        for sh in range(world):
            d = os.path.join(tmp, "shard%d" % sh)
            ref_index.build(d, per_shard * world, nopos=True, vocab=args.vocab, n_shards=world, shard=sh, budget_s=args.ref_build_budget or None)
            dirs.append(d)
        build_s = time.time() - t0
        qf1, qfp = os.path.join(tmp, "q1.txt"), os.path.join(tmp, "qp.txt")
        H.write_queries(qf1, [dict(q, first=0, maxitems=k) for q in leg.timed_pool[:64]])
        H.write_queries(qfp, [dict(q, first=0, maxitems=k) for q in leg.timed_pool])
        cores = max(1, os.cpu_count() or 1)
        one = json.loads(H.xapian_ref("time", qf1, 1, 2, "--seconds", 20, *dirs))
        runs = []
        for t in sorted({cores, max(1, cores // 2), max(1, cores // 4)}, reverse=True):
            runs.append(json.loads(H.xapian_ref("time", qfp, t, 64, "--seconds", max(3.0, args.ref_seconds / 3), *dirs)))
        best = max(runs, key=lambda r: r["qps"])
        return {"kind": "reference", "value": one["qps"], "unit": "queries/s", "cores": 1, "p50_ms": one["p50_us"] / 1e3, "queries": one["queries"],
                "all_cores": {"value": best["qps"], "unit": "queries/s", "cores": physical_cores(), "hardware_threads": cores, "threads": best["threads"], "p50_ms": best["p50_us"] / 1e3,
                              "thread_counts_tried": [{"threads": r["threads"], "value": r["qps"]} for r in runs]},
                "cpu_model": cpu_model(), "shards": world, "docs_per_shard": per_shard, "docs_total": per_shard * world, "index_build_seconds": build_s,
                "index_storage": storage_of(tmp),
                "sample": ("Enquire::get_mset of the vendored Xapian through Xapiand's per-shard protocol (prepare_mset / add_prepared_mset / get_mset / unshard_docids / "
                           "merge_mset, handler.cc:1532-1549; oracle/_ref/xapian_ref time over %d Database handles) on %d round-robin shard indexes of %d documents each "
                           "(a bounded sample: the GPUs hold %d each), without positions, same queries: 1 thread over the first 64 queries of the timed pool, all cores over the pool"
                           % (world, world, per_shard, args.docs_per_gpu))}
    except Exception as e:
        return {"error": repr(e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_spawn(args)

    import torch
    import torch.distributed as dist

    import helpers as H        # query pool shared with tests/test_gpu_configs.py (PHRASE: the corpus restated in Python)
    from xapiand_amd import Database, _lib
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
    # one explicit (non-null) HIP stream for searches and collectives (ShardedSearcher binds the index to torch's current stream per call;
    # the N = 1 timed region binds it once: the match kernels of the batches in flight run back to back there)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)

    k = args.topk
    searcher = ShardedSearcher(db, rank, world, dev)
    L = _lib.lib()
    n_pool_batches = max(4, args.batches_per_step)
    leg = Leg(searcher, args.op, args.terms, args.required, k, n_docs_global, args.vocab, n_pool_batches,
              replay=((_lib.XGM_REPLAY_BATCH_FROZEN if ("frozen" in args.replay and args.op == "PHRASE") else 0) |
                      (_lib.XGM_REPLAY_BATCH_COUNT if "count" in args.replay else 0)))
    m = measure(db, searcher, leg, args, world, rank, dev, args.steps, args.warmup)
    n_timed = len(leg.timed_pool)

    # ---- N > 1: one timed batch against Xapiand's protocol on the oracle (every rank takes part) --------------------
    sharded_checked = None
    if world > 1 and m.get("_snapshot") is not None:
        sharded_checked = sharded_parity(db, leg, world, rank, k, m["_snapshot"])

    # ---- every OTHER timed region of the default line next — C2 with ProtoMSet's count, C3, C5 and C5's two other modes — before the legs that
    # change the process's state (one-query latency, the server threads, the oracle on the host, the one-query-per-call modes): measured, a sub-leg
    # taken after the headline's latency leg runs 5-10 % below the same leg alone (C5 418 k against 466 k queries/s)
    headline = world == 1 and args.op == "AND" and args.terms == 3 and k == 10
    early_count, early_timed, early_errors = None, {}, {}
    if rank == 0 and headline and not args.no_other_configs:
        FROZEN, COUNT = _lib.XGM_REPLAY_BATCH_FROZEN, _lib.XGM_REPLAY_BATCH_COUNT
        try:
            lgc = Leg(searcher, args.op, args.terms, args.required, k, n_docs_global, args.vocab, n_pool_batches, replay=COUNT)
            early_count = (lgc, measure(db, searcher, lgc, args, world, rank, dev, max(10, args.steps // 2), 1))
        except Exception as e:
            early_errors["exact_bounds_mode"] = repr(e)
        for name, op, terms, kk, st in (("C3", "OR", 5, 100, max(10, args.steps // 2)), ("C5", "PHRASE", 3, 10, max(10, args.steps))):
            try:
                # C5's credited figure is the REFERENCE's answer: every query carries XGM_REPLAY_BATCH_FROZEN — the page
                # SelectPostList's frozen weight leaves (selectpostlist.cc:28-55), listed and replayed on the device inside the batch
                lg = Leg(searcher, op, terms, 1, kk, n_docs_global, args.vocab, n_pool_batches, replay=FROZEN if op == "PHRASE" else 0)
                mm = measure(db, searcher, lg, args, world, rank, dev, st, 1)
                modes = []
                if op == "PHRASE":
                    for mname, bits in (("intended_semantics_mode", 0), ("reference_identical_with_exact_count_mode", FROZEN | COUNT)):
                        lg2 = Leg(searcher, op, terms, 1, kk, n_docs_global, args.vocab, n_pool_batches, replay=bits)
                        modes.append((mname, lg2, measure(db, searcher, lg2, args, world, rank, dev, st, 1)))
                early_timed[name] = (lg, mm, modes)
            except Exception as e:
                early_errors[name] = repr(e)

    # ---- planning cost (inside every timed step): host microseconds per query ------------------------------------
    plan_us = L.xgm_debug_plan_us(db._h, leg.batches[0][0], leg.batches[0][1], BATCH, 20)

    # ---- latency mode: one query in flight, host-timed around plan + search incl. H2D/D2H ---------------------
    lat = []
    if (rank == 0 or world > 1) and not args.no_latency:
        lat = latency_leg(db, leg, n_timed)

    # ---- server mode: T host threads, each with ONE query in flight (Xapiand's http_client_pool shape) ----------
    server = None
    if rank == 0 and world == 1 and args.threads > 0:
        server = server_leg(db, leg.descs, leg.gstats, k, args.threads, n_timed)

    result = None
    if rank == 0:
        result = {
            "metric": "queries/sec + p50 latency, %dM-doc synthetic index, %s, top-%d" % (
                args.docs_per_gpu // 1000000, {"AND": "%d-term AND" % args.terms, "OR": "%d-term OR" % args.terms,
                                               "PHRASE": "2-3-term PHRASE"}.get(args.op, "%d-term %s" % (args.terms, args.op)), k),
            "value": m["value"], "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": m["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 postings + f64 BM25", "data": "synthetic",
            "config": {"workload": workload_name(args, world, n_docs_global, k),
                       "docs_per_gpu": args.docs_per_gpu, "docs_total": n_docs_global, "vocab": args.vocab, "op": args.op,
                       "terms_per_query": args.terms, "top_k": k, "batch": BATCH, "batches_per_step": args.batches_per_step, "replay_bits": leg.replay,
                       "queries_per_step": args.batches_per_step * BATCH, "batches_in_flight": args.in_flight if world == 1 else 2,
                       "parallelism": "shard%d" % world,
                       "corpus_seed": hex(CORPUS_SEED), "query_seed": hex(QUERY_SEED),
                       "corpus": "BASELINE.md §3's distribution (Zipf s = 1 over the vocabulary, document lengths uniform in [50, 150], positions 1..L, contiguous docids) drawn from a "
                                 "COUNTER HASH of (seed, document, position) instead of std::mt19937_64 — a deviation from §3's letter, chosen so that the GPU builder, the CPU oracle and "
                                 "the reference's indexer generate any document independently (tools/xgm_corpus.h); every side of every comparison uses the same generator",
                       "step": ("%d x xgm_get_mset_batch_begin (plan: lookups, BM25 init, leaf order + match + merge, 256 queries) ... xgm_batch_end: the hits and "
                                "headers of every batch delivered to pinned HOST memory, %d batches in flight" % (args.batches_per_step, args.in_flight)) if world == 1 else
                               ("%d x (xgm_get_mset_batch_device + all-gather of the shards' top-k + xgm_merge_shards_device, 256 queries) with the merged hits "
                                "copied to pinned HOST memory, double-buffered" % args.batches_per_step)},
            "ms_per_batch": m["ms_per_batch"], "host_ms_per_batch": m["host_ms_per_batch"], "host_us_per_batch": m["host_us_per_batch"],
            "hits_delivered_to_host": m["hits_delivered_to_host"], "last_batch_on_host_equals_synchronous_search": m["last_batch_on_host_equals_synchronous_search"],
            "p50_latency_us": lat[len(lat) // 2] * 1e6 if lat else None,
            "p99_latency_us": lat[int(len(lat) * 0.99)] * 1e6 if lat else None,
            "plan_us_per_query": plan_us,
            "index": {"postings": info.n_postings, "blocks": info.n_blocks, "payload_bytes": info.payload_bytes,
                      "device_bytes": info.device_bytes, "bytes_per_posting": info.device_bytes / max(1, info.n_postings),
                      "build_seconds": build_s},
            "roofline": m["roofline"],
        }
        if server:
            result["server_mode"] = server
        if world == 1:
            result["n_ranks_seen"] = 1
            result["code_path"] = ("one GPU: xgm_get_mset_batch_begin ... xgm_batch_end, %d batches in flight (no collective; the N > 1 lines add one all-gather of "
                                   "packed top-k records + a device merge per batch: their N = 1 point is THIS path, not the sharded one)" % args.in_flight)

    # ---- C2 with the reference's known_matching_docs (the exact HTTP total) for every query: XGM_REPLAY_BATCH_COUNT inside the batch ----
    if rank == 0 and headline and not args.no_other_configs:
        try:
            if early_count is None:
                raise RuntimeError(early_errors.get("exact_bounds_mode", "not measured"))
            lgc, mc = early_count
            # known_matching_docs of a few queries against the host restatement (pinned to the compiled reference) over xgm_search_all's list
            L.xgm_known_matching_docs.restype = C.c_uint64
            L.xgm_known_matching_docs.argtypes = [C.POINTER(C.c_double), C.c_uint64, C.c_uint32, C.c_uint32]
            nchk = 24
            qs_ = (_lib.Query * nchk)(*lgc.timed_plans[:nchk])
            hh, hd, kn = (_lib.Hit * (nchk * k))(), (_lib.ResultHdr * nchk)(), (C.c_uint64 * nchk)()
            db.set_stream(0)
            _lib.check(L.xgm_search_batch_known(db._h, qs_, nchk, k, hh, hd, kn))
            hdr2 = _lib.ResultHdr()
            for qi in range(nchk):
                p = lgc.timed_plans[qi]
                cap = max(1, p.est_max)
                allh = (_lib.Hit * cap)()
                nm = C.c_uint64()
                _lib.check(L.xgm_search_all(db._h, C.byref(p), allh, cap, C.byref(nm), C.byref(hdr2)))
                w = (C.c_double * max(1, nm.value))(*[allh[j].weight for j in range(nm.value)])
                assert L.xgm_known_matching_docs(w, nm.value, k, p.check_at_least) == kn[qi], "known_matching_docs differs on C2 bench query %d" % qi
                assert hd[qi].matches_exact == nm.value, "match count differs on C2 bench query %d" % qi
            result["exact_bounds_mode"] = {"value": mc["value"], "unit": "queries/s", "ms_per_batch": mc["ms_per_batch"], "kernel": mc["roofline"]["kernel"],
                                           "kernel_ms": mc["roofline"]["kernel_ms"], "known_matching_docs_checked_queries": nchk,
                                           "last_batch_on_host_equals_synchronous_search": mc["last_batch_on_host_equals_synchronous_search"],
                                           "what": "every query carries XGM_REPLAY_BATCH_COUNT: the page + ProtoMSet's known_matching_docs (the exact HTTP total), counted on the "
                                                   "device inside the batch (xgm_andw_all_kernel lists every match, xgm_count.hip replays ProtoMSet per unit)"}
        except Exception as e:
            result["exact_bounds_mode"] = {"error": repr(e)}

    # ---- the other single-GPU configurations of BASELINE.json in the same run: C3 (5-term OR, top-100), C5 (2-3-term PHRASE, top-10) ----
    if rank == 0 and headline and not args.no_other_configs:
        others = {}
        FROZEN, COUNT = _lib.XGM_REPLAY_BATCH_FROZEN, _lib.XGM_REPLAY_BATCH_COUNT
        # (their timed regions: measured above, right after the headline's)
        timed = early_timed
        for name, err in early_errors.items():
            if name != "exact_bounds_mode":
                others[name] = {"error": err}
        for name, op, terms, kk, st in (("C3", "OR", 5, 100, max(10, args.steps // 2)), ("C5", "PHRASE", 3, 10, max(10, args.steps))):
            if name not in timed:
                continue
            try:
                lg, mm, modes = timed[name]
                ll = latency_leg(db, lg, min(220, len(lg.timed_pool))) if not args.no_latency else []
                checked, ora, timed_rows = parity_vs_port(db, lg, 128, mm.pop("_snapshot", None))
                port = None
                if not args.no_cpu_baseline:
                    port = time_port(ora, lg.timed_pool[:32], op, kk, min(4.0, args.cpu_seconds), 0)
                extra_modes = {}
                if op == "PHRASE":
                    # beside it: the intended semantics (the top-k of the reference's own full ranking, positional pruning) and the reference's
                    # page WITH its exact known_matching_docs (FROZEN | COUNT: the listing units walk their whole range)
                    for mname, lg2, m2 in modes:
                        c2, _, t2 = parity_vs_port(db, lg2, 128, m2.pop("_snapshot", None), ora=ora)
                        extra_modes[mname] = {"value": m2["value"], "unit": "queries/s", "ms_per_batch": m2["ms_per_batch"], "kernel": m2["roofline"]["kernel"],
                                              "kernel_ms": m2["roofline"]["kernel_ms"], "parity_checked_queries": c2, "timed_batch_rows_checked_against_oracle": t2,
                                              "roofline_frac": m2["roofline"]["frac"], "roofline_basis": m2["roofline"]["basis"]}
                    intended = H.oracle_search_batch(ora, lg.timed_pool[:128], 0, kk)
                    refmode = H.oracle_search_batch(ora, lg.timed_pool[:128], 0, kk, reference_select_bug=True)
                    extra_modes["answers_that_differ_between_the_two_semantics"] = sum(
                        [(d, w) for d, w, _ in a[0]] != [(d, w) for d, w, _ in b[0]] for a, b in zip(intended, refmode))
                try:
                    compat = byte_compatible_leg(db, lg, ora)
                except Exception as e:
                    compat = {"error": repr(e)}
                ora.close()
                others[name] = {"workload": "%s: %dM-doc / %dM-term Zipf index, %s BM25 top-%d, 1 MI355X" % (
                                    name, args.docs_per_gpu // 1000000, args.vocab // 1000000, "5-term disjunctive" if op == "OR" else "2-3-term phrase (positions)", kk),
                                "value": mm["value"], "unit": "queries/s", "ms_per_step": mm["ms_per_step"], "steps": st, "queries_per_step": mm["queries_per_step"],
                                "ms_per_batch": mm["ms_per_batch"], "host_ms_per_batch": mm["host_ms_per_batch"], "host_us_per_batch": mm["host_us_per_batch"],
                                "hits_delivered_to_host": True,
                                "last_batch_on_host_equals_synchronous_search": mm["last_batch_on_host_equals_synchronous_search"],
                                "p50_latency_us": ll[len(ll) // 2] * 1e6 if ll else None, "p99_latency_us": ll[int(len(ll) * 0.99)] * 1e6 if ll else None,
                                "roofline": mm["roofline"], "parity_checked_queries": checked, "timed_batch_rows_checked_against_oracle": timed_rows,
                                "parity_against": ("oracle/xgm_oracle.cc in REFERENCE mode (SelectPostList's stale cached weight restated; pinned to the compiled reference)"
                                                   if op == "PHRASE" else "oracle/xgm_oracle.cc (pinned to the compiled reference)"),
                                "one_query_per_call_mode": compat,
                                "headline_mode": ("REFERENCE-IDENTICAL: every query carries XGM_REPLAY_BATCH_FROZEN — the reference's own page incl. SelectPostList's frozen weight "
                                                  "(xgm_andw_list_kernel + xgm_frozen_finish_kernel inside the batch); `intended_semantics_mode` and the mode with the exact "
                                                  "known_matching_docs are measured beside it") if op == "PHRASE" else
                                                 "the reference's own top-k (bit-identical); `one_query_per_call_mode` adds its known_matching_docs (the exact HTTP total)",
                                "cpu_baseline": dict(port, kind="port", sample="first 32 queries of the timed pool, oracle port on the postings copied back from HBM") if port else None}
                others[name].update(extra_modes)
            except Exception as e:                # a sub-leg must not take the headline line down with it: say what happened
                others[name] = {"error": repr(e)}
        result["other_configs"] = others
        # (top-level copies: the driver's record keeps top-level keys)
        for name in ("C3", "C5"):
            oc = others.get(name, {})
            if "value" in oc:
                result["%s_value" % name.lower()] = oc["value"]
                result["%s_roofline_frac" % name.lower()] = oc["roofline"]["frac"]
                result["%s_kernel_ms" % name.lower()] = oc["roofline"]["kernel_ms"]

    # ---- CPU baseline: the real reference + the oracle port, on this box's host cores -------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(db, leg.timed_pool, args, k, leg.timed_plans, result, m.get("_snapshot"))
    if rank == 0 and world > 1:
        result["parity_checked_queries"] = sharded_checked
        result["parity_against"] = "Xapiand's per-shard protocol on oracle/xgm_oracle.cc (every rank its own shard with the merged statistics; unshard + merge), the LAST TIMED batch's rows as delivered to the host"
        result["n_ranks_seen"] = dist.get_world_size()
        result["code_path"] = "sharded (xgm_get_mset_batch_device + one all-gather of packed records + xgm_merge_shards_packed_device per batch, double-buffered)"
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = sharded_cpu_baseline(args, leg, world, k)
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


HOOK_B1 = os.path.join(ROOT, "oracle", "_ref", "xapian_hook_b1")


def run_hook_b1(flags, qfile, dbdir, timeout=1500):
    r = subprocess.run([HOOK_B1] + flags + ([qfile] if qfile else []) + [dbdir], capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not lines:
        return [{"error": (r.stdout[-600:] + r.stderr[-600:]) or "no output", "returncode": r.returncode}]
    outs = [json.loads(l) for l in lines]
    bad = [l for l in r.stdout.splitlines() if l.startswith(("MISMATCH", "BOUNDS"))]
    for o in outs:
        o["returncode"] = r.returncode
    if bad:
        outs[-1]["first_differences"] = bad[:5]
    return outs


def hook_parity_leg(args, tmp, dbdir, ref_docs, pools, has_positions):
    """The boundary at (reference-index) size, through the REAL reference (VERDICT r3 #1b, r4 #1a): oracle/_ref/xapian_hook_b1 — the vendored
    Xapian with integration/matcher_hook.patch applied and integration/xgm_matcher_hook.cc linked in — exports the glass index this
    run just built with the native reader (xgm_segment_build_from_glass, timed; ONE export and load for all legs), registers it with the
    hook and answers the first 128 queries of the C2 and C3 pools with the hook OFF (CPU matcher) and ON (device) — exact match-count
    figures on (byte-compatible mode: the device counts as ProtoMSet would, xgm_search_replay) — and, the index having positions, the
    first 128 C5 queries in POSITIONAL_REFERENCE mode (the frozen-weight replay on the device) AT THE SAME SIZE.  Identical MSets
    required: docids, weight bits, percentages, matches_lower / estimated / upper, HTTP total."""
    import helpers as H
    if not os.path.exists(HOOK_B1):
        return {"skipped": "oracle/_ref/xapian_hook_b1 is not built"}
    out = {"docs": ref_docs, "shapes": {}}
    total = dict(queries=0, mismatches=0, bounds_violations=0, http_total_equal=0, answered_on_device=0, threaded_queries=0, threaded_mismatches=0)
    # every configuration twice over ONE export + load: in its byte-compatible mode (exact match-count figures; C5: the reference's page AND figures) and in
    # the hook's default mode (C5: POSITIONAL_REFERENCE's page alone) — each followed by the SAME queries from 64 native threads (Xapiand's load shape:
    # every HTTP worker its own Database handle, one get_mset at a time; the calls meet in the index's dispatcher), every answer compared
    flags = ["--threads", "64", "--thread-repeat", "4"]
    labels = {}
    for name, mode, label in (("C2", "exact-bounds", "C2"), ("C3", "exact-bounds", "C3"), ("C5", "positional-reference", "C5 (POSITIONAL_REFERENCE)"),
                              ("C2", "plain", "C2 (default mode)"), ("C3", "plain", "C3 (default mode)"), ("C5", "positional-reference-page", "C5 (POSITIONAL_REFERENCE, page only)")):
        if name == "C5" and not has_positions:
            continue
        qf = os.path.join(tmp, "hook_%s.txt" % name)
        if not os.path.exists(qf):
            H.write_queries(qf, [dict(q, first=0) for q in pools[name][:128]])
        leg_name = "%s_%s" % (name, mode)
        labels[leg_name] = label
        flags += ["--leg", "%s:%s:%s" % (leg_name, mode, qf)]
    for r in run_hook_b1(flags + ["-"], None, dbdir):
        name = r.pop("leg", "error")
        out["shapes"][labels.get(name, name)] = r
        if r.get("queries") and r.get("hook_seconds"):
            r["hook_queries_per_second"] = r["queries"] / r["hook_seconds"]
            r["cpu_matcher_queries_per_second"] = r["queries"] / r["cpu_matcher_seconds"]
        if r.get("threaded_queries") and r.get("threaded_seconds"):
            r["threaded_hook_queries_per_second"] = r["threaded_queries"] / r["threaded_seconds"]
        if "export_seconds" in r and "exporter" not in out:
            out["exporter"] = {"docs": r["docs"], "seconds": r["export_seconds"], "segment_bytes": r["segment_bytes"], "open_seconds": r["open_seconds"], "positions": has_positions,
                               "what": "xgm_segment_build_from_glass: the glass B-trees read natively (postlist.glass, position.glass), block-encoded, written; once for all legs"}
    if has_positions:
        out["docs_with_positions"] = ref_docs
    for label, r in out["shapes"].items():
        fast = "default mode" in label or "page only" in label          # (the fast modes' match-count figures are bounds, not the reference's: not part of the byte-compatible totals)
        for key in total:
            if key in r and (not fast or key.startswith("threaded")):
                total[key] += r[key]
        if fast and r.get("mismatches"):
            total["mismatches"] += r["mismatches"]
        if "error" in r:
            total["errors"] = total.get("errors", 0) + 1
    out.update(total)
    return out


def time_reference(H, qfile, threads, repeat, seconds, dbdir):
    """`xapian_ref time` (oracle/ref_build/ref_driver.cc cmd_time: one shared atomic work counter, handles opened and warmed before the start
    barrier) → its JSON."""
    return json.loads(H.xapian_ref("time", qfile, threads, repeat, "--seconds", seconds, dbdir))


def reference_all_cores(H, qfile, cores, seconds, dbdir):
    """The reference on every core — and on half and a quarter of them: one Database handle per thread, and every get_mset reads its B-tree
    blocks with pread(); on a 256-thread host the kernel side of those reads can serialise the threads (measured at 1 M documents: 256
    threads answer 10x one thread's rate, each query's own latency only 1.7x longer).  The BEST thread count is reported as the all-core
    figure (generous to the CPU side), every measurement beside it with the share of the threads' time spent inside get_mset."""
    runs = []
    for t in sorted({cores, max(1, cores // 2), max(1, cores // 4)}, reverse=True):
        r = time_reference(H, qfile, t, 64, seconds, dbdir)
        runs.append({"threads": t, "value": r["qps"], "p50_ms": r["p50_us"] / 1e3, "p99_ms": r["p99_us"] / 1e3, "queries": r["queries"], "seconds": r["wall_s"],
                     "share_of_thread_time_inside_get_mset": r["sum_latency_s"] / (t * r["wall_s"])})
    best = max(runs, key=lambda x: x["value"])
    return best, runs, r


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def physical_cores():
    """(sockets x cores per socket) from /proc/cpuinfo — os.cpu_count() counts hardware THREADS (2 per core with SMT)."""
    try:
        pairs = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
        return len(pairs) or None
    except OSError:
        return None


def storage_of(path):
    """The file system a path lives on (the reference's glass index: tmpfs = page-cache resident by construction)."""
    try:
        best = ("", "?", "?")
        for line in open("/proc/mounts"):
            dev, mnt, fstype = line.split()[:3]
            if (path == mnt or path.startswith(mnt.rstrip("/") + "/")) and len(mnt) > len(best[0]):
                best = (mnt, fstype, dev)
        return {"mount": best[0], "fstype": best[1]}
    except OSError:
        return None


def reference_leg(args, sample, k, n_required, full, ora_full, hook_pools=None, pool_all=None):
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
    headline = args.op == "AND" and args.terms == 3 and k == 10
    ref_docs = args.ref_docs if args.ref_docs >= 0 else (args.docs_per_gpu if headline else args.docs_per_gpu // 10 if args.op == "PHRASE" else args.docs_per_gpu // 5)
    if not H.have_xapian_ref() or ref_docs <= 0:
        return None
    # the headline run builds ONE index of the configuration's own size WITH positions: it serves C2, C3 (the position table is never
    # opened by AND / OR) and C5, the reference's timings of all three and the hook-parity legs (round 4: two indexes, C5's at 2 M documents)
    with_positions = args.op == "PHRASE" or (hook_pools is not None and not args.ref_no_positions)
    # the index lives in memory-backed storage when there is one: indexing through WritableDatabase is write-heavy
    tmp = tempfile.mkdtemp(prefix="xgm_ref_", dir="/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None)
    try:
        dbdir = os.path.join(tmp, "glass")
        binfo = ref_index.build(dbdir, ref_docs, nopos=not with_positions, vocab=args.vocab, budget_s=args.ref_build_budget or None)
        qfile = os.path.join(tmp, "q.txt")
        H.write_queries(qfile, [dict(q, first=0, maxitems=k) for q in sample])
        cores = max(1, os.cpu_count() or 1)
        one = time_reference(H, qfile, 1, 2, 60, dbdir)
        # all cores: the whole timed pool (>= 8 queries per thread wherever the pool allows), handed out by one atomic counter, time-boxed
        pool_file = os.path.join(tmp, "q_pool.txt")
        H.write_queries(pool_file, [dict(q, first=0, maxitems=k) for q in (pool_all or sample)])
        best, sweep, many = reference_all_cores(H, pool_file, cores, max(3.0, args.ref_seconds / 2), dbdir)
        others = {}
        if hook_pools is not None:
            # the reference itself on C3 and C5: 1 thread on the first 48 queries, all cores on the pool, same index
            for name, kk in (("C3", 100), ("C5", 10)):
                if name == "C5" and not with_positions:
                    continue
                try:
                    qf1, qfp = os.path.join(tmp, "q1_%s.txt" % name), os.path.join(tmp, "qp_%s.txt" % name)
                    H.write_queries(qf1, [dict(q, first=0, maxitems=kk) for q in hook_pools[name][:48]])
                    H.write_queries(qfp, [dict(q, first=0, maxitems=kk) for q in hook_pools[name]])
                    o1 = time_reference(H, qf1, 1, 1, 30, dbdir)
                    ob, osweep, om = reference_all_cores(H, qfp, cores, max(3.0, args.ref_seconds / 3), dbdir)
                    others[name] = {"kind": "reference", "value": o1["qps"], "unit": "queries/s", "cores": 1, "p50_ms": o1["p50_us"] / 1e3, "queries": o1["queries"],
                                    "all_cores": {"value": ob["value"], "unit": "queries/s", "cores": physical_cores(), "hardware_threads": cores, "threads": ob["threads"], "p50_ms": ob["p50_ms"], "queries": ob["queries"],
                                                  "pool": om["pool"], "seconds": ob["seconds"], "thread_counts_tried": osweep},
                                    "docs": ref_docs,
                                    "sample": "Enquire::get_mset of the vendored Xapian (oracle/_ref/xapian_ref time) on the glass index of the headline leg (%d documents%s): "
                                              "1 thread over the first 48 queries of the timed pool, all cores over %d queries of it for %.0f s" % (
                                                  ref_docs, ", positions" if with_positions else "", om["pool"], om["wall_s"])}
                except Exception as e:
                    others[name] = {"error": repr(e)}
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
        out = {"kind": "reference", "value": one["qps"], "unit": "queries/s", "cores": 1, "p50_ms": one["p50_us"] / 1e3,
               "all_cores": {"value": best["value"], "unit": "queries/s", "cores": physical_cores(), "hardware_threads": cores, "threads": best["threads"], "p50_ms": best["p50_ms"], "queries": best["queries"],
                             "pool": many["pool"], "seconds": best["seconds"], "efficiency_vs_cores_x_one_thread": best["value"] / (cores * one["qps"]),
                             "thread_counts_tried": sweep, "scheduling": many.get("scheduling")},
               "cpu_model": cpu_model(), "physical_cores": physical_cores(), "hardware_threads": cores, "index_storage": storage_of(tmp), "_others": others,
               "docs": ref_docs, "index_build": binfo, "port_same_index": {kk: vv for kk, vv in port.items() if kk != "all_cores"},
               "port_over_reference": port["value"] / one["qps"], "port_vs_reference_parity_checked": checked}
        if hook_pools is not None:
            try:
                out["_hook_parity"] = hook_parity_leg(args, tmp, dbdir, ref_docs, hook_pools, with_positions)
            except Exception as e:
                out["_hook_parity"] = {"error": repr(e)}
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(db, pool_q, args, k, timed_plans, result=None, snapshot=None):
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
    timed_rows = [0]
    snap_rows = snapshot_rows(snapshot, k, min(len(sample), BATCH)) if snapshot is not None else None

    def checker(qi, hits):          # parity of the GPU answer on every sampled query, outside the timing
        _lib.check(L.xgm_search(db._h, C.byref(timed_plans[qi]), one_hits, C.byref(one_hdr)))
        got = [(one_hits[j].docid, one_hits[j].weight) for j in range(one_hdr.n_hits)]
        assert got == [(d, w) for d, w, _ in hits], "GPU/CPU parity failure on bench query %d" % qi
        checked[0] += 1
        # ... and of the very rows the TIMED entry point delivered to the host (batch 0 of the last timed step)
        if snap_rows is not None and qi < len(snap_rows):
            assert snap_rows[qi][0] == [(d, w) for d, w, _ in hits], "timed batch / oracle parity failure on bench query %d" % qi
            timed_rows[0] += 1
    db.set_stream(0)
    full = time_port(ora, sample, args.op, k, args.cpu_seconds, n_required, checker)
    full["all_cores"] = port_all_cores(ora, sample, args.op, k, min(5.0, args.cpu_seconds), n_required)
    full["docs"] = db.info().doccount
    sample_txt = ("first %d queries of the timed pool; port = oracle/xgm_oracle.cc (the reference's glass-chunk / MultiAnd / BM25 / "
                  "ProtoMSet algorithm restated) on the same %d-doc postings copied back from HBM, %.1f s of CPU work" % (len(sample), full["docs"], full["seconds"]))
    ref = None
    try:
        hook_pools = None
        if result is not None and not args.no_hook_parity and args.op == "AND" and args.terms == 3 and k == 10:
            n_docs = db.info().doccount
            hook_pools = {"C2": [dict(q, maxitems=10) for q in pool_q[:128]],
                          "C3": H.bench_pool("OR", 5, 1, n_docs, args.vocab, n=100 + 2048, seed=QUERY_SEED, maxitems=100)[100:],
                          "C5": H.bench_pool("PHRASE", 3, 1, n_docs, args.vocab, n=100 + 2048, seed=QUERY_SEED, maxitems=10)[100:]}
        ref = reference_leg(args, sample, k, n_required, full, ora, hook_pools, pool_q)
        if ref and "_hook_parity" in ref:
            result["hook_parity"] = ref.pop("_hook_parity")
        if ref:
            # the reference's own timings of C3 / C5 become those sub-legs' cpu_baseline (the port's stays beside it)
            for name, rb in (ref.pop("_others", None) or {}).items():
                oc = (result or {}).get("other_configs", {}).get(name)
                if oc is not None and "error" not in rb:
                    rb["port_same_queries"] = oc.get("cpu_baseline")
                    oc["cpu_baseline"] = rb
                elif oc is not None:
                    oc["cpu_baseline_reference_error"] = rb["error"]
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
            if args.op == "AND" and args.terms == 3 and k == 10 and full["docs"] == 10_000_000 and os.path.exists(rpath):    # (only when --ref-docs made this run's index smaller)
                # the same leg run ONCE at the full size on this box type (too long for every default run)
                r = json.load(open(rpath))
                out["reference_full_size_measured"] = {"value": r["one_thread"]["value"], "unit": "queries/s", "cores": 1, "all_cores": r["all_cores"],
                                                       "docs": r["docs"], "source": "profiles/r02_reference_full.json (python bench.py --ref-docs 10000000)"}
        out["sample"] = ("Enquire::get_mset of the vendored Xapian (oracle/_ref/xapian_ref time), glass index of the first %d documents of the same "
                         "corpus built on this box (%.0f s on %d cores + %.0f s compact), same queries; " % (ref["docs"], ref["index_build"]["build_s"],
                                                                                                       ref["index_build"]["procs"], ref["index_build"]["compact_s"])) + sample_txt
        out["parity_checked_queries"] = checked[0]
        out["timed_batch_rows_checked_against_oracle"] = timed_rows[0]
        return out
    out = dict(full)
    out.update(kind="port", sample=sample_txt, parity_checked_queries=checked[0], timed_batch_rows_checked_against_oracle=timed_rows[0])
    return out


if __name__ == "__main__":
    main()
