"""One shard per GPU, one process per GPU: Xapiand's per-shard protocol over torch.distributed.

Reference protocol (src/database/handler.cc:1485-1549): prepare_mset on every shard → Σ statistics
(Enquire::add_prepared_mset, src/xapian/api/enquire.cc:385-394) → get_mset(0, first+maxitems) on every
shard with the merged statistics → unshard_docids + merge_mset.  Here: one all-reduce(SUM) of the
statistics per query pool, then per batch ONE all-gather of fixed-size top-k records — every rank's hits and headers packed
in one buffer ([nq][k] xgm_hit then [nq] xgm_result_hdr: a second collective per batch would cost a second launch and a second
trip round the xGMI ring for 8 KB; round 5) — RCCL over xGMI on GPUs, gloo in the CPU tests, and a merge on every rank
(xgm_merge_shards_packed_device).  No other collective is on the data path.

The search and merge steps are injectable so the collective logic can be exercised on CPU (gloo) where
the HIP path cannot run: the defaults call the C ABI (xgm_search_batch_device /
xgm_merge_shards_device); tests pass oracle-backed callables.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib
from .enquire import plan

HIT_F64 = 2      # an xgm_hit is 16 bytes = 2 float64 lanes when viewed as a tensor
HDR_F64 = 4      # an xgm_result_hdr is 32 bytes


class ShardedSearcher:
    def __init__(self, shard, rank, world, device, group=None, search_fn=None, merge_fn=None, force_collective=False):
        """`shard` exposes get_doccount/get_total_length/has_positions/get_termfreq (a Database does).
        force_collective: take the all-gather + merge path even with world == 1 (a 1-rank RCCL group on a
        single-GPU box exercises exactly the code the 8-GPU run uses).

        Stream discipline of the device path: every xgm_* call of a batch is enqueued on torch's CURRENT stream of
        `device` (bound per call with xgm_index_set_stream), the stream the collectives are ordered against, so the
        merge cannot read the gathered records before the all-gather has produced them."""
        self.shard, self.rank, self.world, self.device, self.group = shard, rank, world, device, group
        self.search_fn = search_fn or self._device_search
        self.merge_fn = merge_fn or self._device_merge
        self.force_collective = force_collective
        self._bufs = {}
        self._ks = {}
        self._host_collectives = False
        if world > 1 or force_collective:
            # gloo has no device all-gather: stage the (tiny) records through pinned host memory — the path of the
            # single-GPU test that runs two ranks on one device
            self._host_collectives = torch.device(device).type == "cuda" and dist.get_backend(group) == "gloo"

    # -- statistics ---------------------------------------------------------------------------------
    def merged_stats(self, queries):
        """One all-reduce for a whole pool of queries → list of GlobalStats (merged over shards)."""
        terms = sorted({t for q in queries for t in q.terms})
        vec = [self.shard.get_total_length(), self.shard.get_doccount(), 1 if self.shard.has_positions() else 0]
        vec += [self.shard.get_termfreq(t) for t in terms]
        t = torch.tensor(vec, dtype=torch.int64, device="cpu" if self._host_collectives else self.device)
        if self.world > 1:
            dist.all_reduce(t, group=self.group)
        vals = t.tolist()
        tf = dict(zip(terms, vals[3:]))
        out = []
        for q in queries:
            gs = _lib.GlobalStats()
            gs.total_length, gs.collection_size, gs.full_db_has_positions = vals[0], vals[1], 1 if vals[2] else 0
            for i, term in enumerate(q.terms):
                gs.termfreq[i] = tf[term]
            out.append(gs)
        return out

    def prepare(self, queries, first, maxitems):
        """Plan every query on this shard with the merged statistics (shard-local leaf order)."""
        stats = self.merged_stats(queries)
        return [plan(self.shard, q, 0, first + maxitems, global_stats=gs) for q, gs in zip(queries, stats)]

    def describe(self, queries, first, maxitems):
        """The queries as the hook receives them, plus their merged statistics: ((xgm_query_desc * n), (xgm_global_stats * n))
        for run_descs.  The per-shard request is (0, first + maxitems) exactly as Xapiand issues it (handler.cc:1532-1549)."""
        from .enquire import BM25Weight, _desc
        stats = self.merged_stats(queries)
        n = len(queries)
        descs = (_lib.QueryDesc * n)()
        gs = (_lib.GlobalStats * n)()
        self._keep = []
        for i, q in enumerate(queries):
            d = _desc(q, 0, first + maxitems, 0, BM25Weight())
            self._keep.append(d)
            descs[i] = d
            gs[i] = stats[i]
        return descs, gs

    # -- one batch ----------------------------------------------------------------------------------
    def _buffers(self, nq, k, slot=0):
        """Per (batch shape, slot): this rank's packed record (hits and hdrs are VIEWS into it), the gathered records of all ranks
        (all_hits / all_hdrs: strided views for host-side merges), the merged output.  `slot` lets a caller keep two batches in flight."""
        key = (nq, k, slot)
        if key not in self._bufs:
            mk = lambda *shape: torch.zeros(shape, dtype=torch.float64, device=self.device)  # noqa: E731
            nh, nd = nq * k * HIT_F64, nq * HDR_F64
            rec, all_rec = mk(nh + nd), mk(self.world, nh + nd)
            self._bufs[key] = dict(rec=rec, hits=rec[:nh].view(nq, k, HIT_F64), hdrs=rec[nh:].view(nq, HDR_F64), all_rec=all_rec,
                                   all_hits=all_rec[:, :nh].view(self.world, nq, k, HIT_F64), all_hdrs=all_rec[:, nh:].view(self.world, nq, HDR_F64),
                                   out_hits=mk(nq, k, HIT_F64), out_hdrs=mk(nq, HDR_F64))
        return self._bufs[key]

    def run_batch(self, batch, nq, k):
        """batch: what search_fn understands (an (xgm_query * nq) array for the device path).
        Returns (hits, hdrs) tensors holding the merged result with GLOBAL docids on every rank."""
        b = self._buffers(nq, k)
        self.search_fn(batch, nq, k, b["hits"], b["hdrs"])
        return self._gather_merge(b, nq, k)

    def run_descs(self, descs, gstats, nq, k, slot=0):
        """Like run_batch, from query DESCRIPTIONS: planning (dictionary lookups, BM25Weight::init, leaf order) happens
        inside the call (xgm_get_mset_batch_device) — what the matcher hook does per get_mset.  descs: (xgm_query_desc * nq),
        gstats: (xgm_global_stats * nq) merged statistics or None.  Everything is enqueued on torch's current stream and nothing is
        waited for: with slot alternating between 0 and 1 the host plans batch i + 1 while the GPU runs batch i."""
        b = self._buffers(nq, k, slot)
        self._bind_stream()
        _lib.check(_lib.lib().xgm_get_mset_batch_device(self.shard._h, descs, gstats, nq, k, b["hits"].data_ptr(), b["hdrs"].data_ptr()))
        return self._gather_merge(b, nq, k)

    def _gather_merge(self, b, nq, k):
        if self.world == 1 and not self.force_collective:
            return b["hits"], b["hdrs"]
        # ONE collective: output = the ranks' packed records one after the other (the layout both RCCL and gloo accept)
        if self._host_collectives:
            mine = b["rec"].cpu()
            gathered = torch.empty((self.world * mine.numel(),), dtype=torch.float64)
            dist.all_gather_into_tensor(gathered, mine, group=self.group)
            b["all_rec"].view(-1).copy_(gathered)
        else:
            dist.all_gather_into_tensor(b["all_rec"].view(-1), b["rec"], group=self.group)
        self.n_collectives = getattr(self, "n_collectives", 0) + 1
        self._all_rec = b["all_rec"]
        self.merge_fn(b["all_hits"], b["all_hdrs"], self.world, nq, k, b["out_hits"], b["out_hdrs"])
        return b["out_hits"], b["out_hdrs"]

    # -- defaults: the HIP path ---------------------------------------------------------------------
    def _bind_stream(self):
        """Run the index's work on torch's current stream (see __init__).  The null stream is a special case of the C
        ABI (handle 0 = "the library's own stream, synchronised before the call returns"): then the gathered records are
        made visible by synchronising the current stream before the merge."""
        h = torch.cuda.current_stream(self.device).cuda_stream
        self.shard.set_stream(h)
        return h

    def _device_search(self, batch, nq, k, hits, hdrs):
        self._bind_stream()
        _lib.check(_lib.lib().xgm_search_batch_device(self.shard._h, batch, nq, k, hits.data_ptr(), hdrs.data_ptr()))

    def _device_merge(self, all_hits, all_hdrs, n_shards, nq, k, out_hits, out_hdrs):
        if self._bind_stream() == 0:
            torch.cuda.current_stream(self.device).synchronize()
        ks = self._ks.get((nq, k))
        if ks is None:
            ks = self._ks[(nq, k)] = (C.c_uint32 * nq)(*([k] * nq))
        # (all_hits / all_hdrs are views into the gathered packed records: the device merge takes the records themselves)
        _lib.check(_lib.lib().xgm_merge_shards_packed_device(self.shard._h, self._all_rec.data_ptr(), n_shards, nq, k, ks,
                                                             out_hits.data_ptr(), out_hdrs.data_ptr()))


def decode_results(hits, hdrs):
    """Tensors → per query list of (docid, weight, subqs) + header dicts (host side helper)."""
    import numpy as np
    h = hits.detach().cpu().contiguous().numpy().view(np.uint8)
    nq, k = hits.shape[0], hits.shape[1]
    h = h.reshape(nq, k, 16)
    d = hdrs.detach().cpu().contiguous().numpy().view(np.uint8).reshape(nq, 32)
    out = []
    for i in range(nq):
        n = int(d[i, 0:4].view(np.uint32)[0])
        rows = [(int(h[i, j, 0:4].view(np.uint32)[0]), float(h[i, j, 8:16].view(np.float64)[0]), int(h[i, j, 4:8].view(np.uint32)[0]))
                for j in range(n)]
        out.append((rows, dict(n_hits=n, max_subqs=int(d[i, 4:8].view(np.uint32)[0]), matches=int(d[i, 8:16].view(np.uint64)[0]) & ((1 << 63) - 1), matches_lower_bound=bool(int(d[i, 8:16].view(np.uint64)[0]) >> 63),
                               max_attained=float(d[i, 16:24].view(np.float64)[0]), max_possible=float(d[i, 24:32].view(np.float64)[0]))))
    return out
