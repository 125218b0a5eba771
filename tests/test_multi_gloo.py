"""CPU-only, world_size 2, gloo: the N > 1 path of xapiand_amd.distributed — statistics all-reduce,
fixed-size top-k all-gather, unshard + merge — with the per-shard search supplied by the CPU oracle
(the HIP search cannot run without a GPU).  The merged answer must equal Xapiand's protocol run on the
oracle directly (which tests/test_oracle_vs_reference.py pins to the real reference)."""
import os
import socket
import struct

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H

N_DOCS, VOCAB, WORLD, K = 3000, 8000, 2, 10


class OracleShard:
    """Shard statistics interface (what ShardedSearcher needs) backed by the oracle corpus."""

    def __init__(self, corpus):
        self.c = corpus

    def get_doccount(self):
        return self.c.v.doccount

    def get_total_length(self):
        return self.c.v.total_length

    def has_positions(self):
        return bool(self.c.v.has_positions)

    def get_termfreq(self, term):
        return self.c.termfreq(term)


def pack_hits(rows, hdr, k):
    hits = np.zeros((k, 16), dtype=np.uint8)
    for j, (d, w, m) in enumerate(rows):
        hits[j] = np.frombuffer(struct.pack("<IId", d, m, w), dtype=np.uint8)
    h = np.frombuffer(struct.pack("<IIQdd", len(rows), hdr.max_subqs, hdr.matches, hdr.max_attained, hdr.max_possible), dtype=np.uint8)
    return hits, h


def worker(rank, port, queries, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from xapiand_amd.distributed import ShardedSearcher, decode_results
        from xapiand_amd.enquire import Query
        corpus = H.Corpus(N_DOCS, VOCAB, n_shards=WORLD, shard=rank)
        shard = OracleShard(corpus)
        qobjs = [Query(q["op"], q["terms"]) for q in queries]
        state = {}

        def search_fn(batch, nq, k, hits, hdrs):
            gstats = state["stats"]
            hb = np.zeros((nq, k, 16), dtype=np.uint8)
            db = np.zeros((nq, 32), dtype=np.uint8)
            for i, q in enumerate(batch):
                gs = gstats[i]
                g = dict(total_length=gs.total_length, collection_size=gs.collection_size, has_positions=bool(gs.full_db_has_positions),
                         termfreq=[gs.termfreq[j] for j in range(len(q["terms"]))])
                rows, hdr = H.oracle_search(corpus, q["op"], q["terms"], 0, k, q.get("window", 0), g)
                hb[i], db[i] = pack_hits(rows, hdr, k)
            hits.copy_(torch.from_numpy(hb.view(np.float64).reshape(nq, k, 2)))
            hdrs.copy_(torch.from_numpy(db.view(np.float64).reshape(nq, 4)))

        def merge_fn(all_hits, all_hdrs, n_shards, nq, k, out_hits, out_hdrs):
            res = [decode_results(all_hits[s], all_hdrs[s]) for s in range(n_shards)]
            hb = np.zeros((nq, k, 16), dtype=np.uint8)
            db = np.zeros((nq, 32), dtype=np.uint8)
            for i in range(nq):
                rows, matches, mp_, ma, ms = [], 0, 0.0, 0.0, 0
                for s in range(n_shards):
                    r, h = res[s][i]
                    rows += [((d - 1) * n_shards + s + 1, w, m) for d, w, m in r]     # multi.h:69-73
                    matches += h["matches"]
                    mp_ = max(mp_, h["max_possible"])
                    if h["max_attained"] > ma:
                        ma, ms = h["max_attained"], h["max_subqs"]
                rows.sort(key=lambda x: (-x[1], x[0]))
                rows = rows[:k]

                class Hdr:
                    max_subqs, max_attained, max_possible = ms, ma, mp_
                Hdr.matches = matches
                hb[i], db[i] = pack_hits(rows, Hdr, k)
            out_hits.copy_(torch.from_numpy(hb.view(np.float64).reshape(nq, k, 2)))
            out_hdrs.copy_(torch.from_numpy(db.view(np.float64).reshape(nq, 4)))

        ss = ShardedSearcher(shard, rank, WORLD, torch.device("cpu"), search_fn=search_fn, merge_fn=merge_fn)
        state["stats"] = ss.merged_stats(qobjs)
        hits, hdrs = ss.run_batch(queries, len(queries), K)
        got = decode_results(hits, hdrs)
        ret[rank] = [[(d, w) for d, w, _ in rows] for rows, _ in got]
        # every rank must also agree on the merged statistics
        ret["stats%d" % rank] = [(g.total_length, g.collection_size, [g.termfreq[j] for j in range(3)]) for g in state["stats"][:5]]
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_sharded_search_matches_protocol(built):
    queries = H.gen_term_queries("AND", 12, 3, 1, 64, maxitems=K, seed=91) + H.gen_term_queries("OR", 8, 4, 1, 400, maxitems=K, seed=92)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = free_port()
    procs = [mp.get_context("spawn").Process(target=worker, args=(r, port, queries, ret)) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    shards = [H.Corpus(N_DOCS, VOCAB, n_shards=WORLD, shard=s) for s in range(WORLD)]
    want = [[(d, w) for d, w, _ in H.oracle_search_sharded(shards, q["op"], q["terms"], 0, K)] for q in queries]
    assert ret[0] == want
    assert ret[1] == want
    assert ret["stats0"] == ret["stats1"]
    full = H.Corpus(N_DOCS, VOCAB)
    assert ret["stats0"][0][0] == full.v.total_length and ret["stats0"][0][1] == full.v.doccount
