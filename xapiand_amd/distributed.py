"""One shard per GPU, one process per GPU: Xapiand's per-shard protocol over torch.distributed.

Reference protocol (src/database/handler.cc:1485-1549): prepare_mset on every shard → Σ statistics
(Enquire::add_prepared_mset, src/xapian/api/enquire.cc:385-394) → get_mset(0, first+maxitems) on every
shard with the merged statistics → unshard_docids + merge_mset.  Here: one all-reduce(SUM) of the
statistics per query pool, then per batch one all-gather of fixed-size top-k records (RCCL over xGMI on
GPUs, gloo in the CPU tests) and a merge on every rank.  No other collective is on the data path.

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
    def __init__(self, shard, rank, world, device, group=None, search_fn=None, merge_fn=None):
        """`shard` exposes get_doccount/get_total_length/has_positions/get_termfreq (a Database does)."""
        self.shard, self.rank, self.world, self.device, self.group = shard, rank, world, device, group
        self.search_fn = search_fn or self._device_search
        self.merge_fn = merge_fn or self._device_merge
        self._bufs = {}

    # -- statistics ---------------------------------------------------------------------------------
    def merged_stats(self, queries):
        """One all-reduce for a whole pool of queries → list of GlobalStats (merged over shards)."""
        terms = sorted({t for q in queries for t in q.terms})
        vec = [self.shard.get_total_length(), self.shard.get_doccount(), 1 if self.shard.has_positions() else 0]
        vec += [self.shard.get_termfreq(t) for t in terms]
        t = torch.tensor(vec, dtype=torch.int64, device=self.device)
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

    # -- one batch ----------------------------------------------------------------------------------
    def _buffers(self, nq, k):
        key = (nq, k)
        if key not in self._bufs:
            mk = lambda *shape: torch.zeros(shape, dtype=torch.float64, device=self.device)  # noqa: E731
            self._bufs[key] = dict(hits=mk(nq, k, HIT_F64), hdrs=mk(nq, HDR_F64), all_hits=mk(self.world, nq, k, HIT_F64),
                                   all_hdrs=mk(self.world, nq, HDR_F64), out_hits=mk(nq, k, HIT_F64), out_hdrs=mk(nq, HDR_F64))
        return self._bufs[key]

    def run_batch(self, batch, nq, k):
        """batch: what search_fn understands (an (xgm_query * nq) array for the device path).
        Returns (hits, hdrs) tensors holding the merged result with GLOBAL docids on every rank."""
        b = self._buffers(nq, k)
        self.search_fn(batch, nq, k, b["hits"], b["hdrs"])
        if self.world == 1:
            return b["hits"], b["hdrs"]
        # output = concatenation of the ranks' inputs along dim 0 (the layout both RCCL and gloo accept)
        dist.all_gather_into_tensor(b["all_hits"].view(self.world * nq, k, HIT_F64), b["hits"], group=self.group)
        dist.all_gather_into_tensor(b["all_hdrs"].view(self.world * nq, HDR_F64), b["hdrs"], group=self.group)
        self.merge_fn(b["all_hits"], b["all_hdrs"], self.world, nq, k, b["out_hits"], b["out_hdrs"])
        return b["out_hits"], b["out_hdrs"]

    # -- defaults: the HIP path ---------------------------------------------------------------------
    def _device_search(self, batch, nq, k, hits, hdrs):
        _lib.check(_lib.lib().xgm_search_batch_device(self.shard._h, batch, nq, k, hits.data_ptr(), hdrs.data_ptr()))

    def _device_merge(self, all_hits, all_hdrs, n_shards, nq, k, out_hits, out_hdrs):
        ks = (C.c_uint32 * nq)(*([k] * nq))
        _lib.check(_lib.lib().xgm_merge_shards_device(self.shard._h, all_hits.data_ptr(), all_hdrs.data_ptr(), n_shards, nq, k, ks,
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
        out.append((rows, dict(n_hits=n, max_subqs=int(d[i, 4:8].view(np.uint32)[0]), matches=int(d[i, 8:16].view(np.uint64)[0]),
                               max_attained=float(d[i, 16:24].view(np.float64)[0]), max_possible=float(d[i, 24:32].view(np.float64)[0]))))
    return out
