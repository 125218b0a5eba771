"""CPU-only, world_size 2 and 8 (the node's shape), gloo: the N > 1 path of xapiand_amd.distributed — statistics all-reduce,
ONE packed all-gather of fixed-size top-k records per batch, unshard + merge — with the per-shard search supplied by the CPU
oracle (the HIP search cannot run without a GPU).  The merged answer must equal Xapiand's protocol run on the oracle directly
(which tests/test_oracle_vs_reference.py pins to the real reference) and, on the queries of the 8-shard golden fixture, the
MSets the compiled reference itself produced over 8 shards (tests/golden/sharded8_and3_top10.json)."""
import json
import os
import socket
import struct

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H

N_DOCS, VOCAB, K = 3000, 8000, 10


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


def worker(rank, port, batches, ret, WORLD, n_docs, vocab):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from xapiand_amd.distributed import ShardedSearcher, decode_results
        from xapiand_amd.enquire import Query
        corpus = H.Corpus(n_docs, vocab, n_shards=WORLD, shard=rank)
        shard = OracleShard(corpus)
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
        out = []
        for queries, k in batches:
            qobjs = [Query(q["op"], q["terms"]) for q in queries]
            state["stats"] = ss.merged_stats(qobjs)
            hits, hdrs = ss.run_batch(queries, len(queries), k)
            got = decode_results(hits, hdrs)
            out.append([([(d, w) for d, w, _ in rows], h["max_possible"]) for rows, h in got])
        ret[rank] = out
        ret["collectives%d" % rank] = ss.n_collectives            # ONE all-gather per batch (hits and headers packed in one record)
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


def run_ranks(world, batches, n_docs, vocab):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = free_port()
    procs = [mp.get_context("spawn").Process(target=worker, args=(r, port, batches, ret, world, n_docs, vocab)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    return ret


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_search_matches_protocol(built, world):
    queries = H.gen_term_queries("AND", 12, 3, 1, 64, maxitems=K, seed=91) + H.gen_term_queries("OR", 8, 4, 1, 400, maxitems=K, seed=92)
    ret = run_ranks(world, [(queries, K)], N_DOCS, VOCAB)
    shards = [H.Corpus(N_DOCS, VOCAB, n_shards=world, shard=s) for s in range(world)]
    want = [[(d, w) for d, w, _ in H.oracle_search_sharded(shards, q["op"], q["terms"], 0, K)] for q in queries]
    for r in range(world):
        assert [rows for rows, _ in ret[r][0]] == want, r
        assert ret["stats%d" % r] == ret["stats0"]
        assert ret["collectives%d" % r] == 1
    full = H.Corpus(N_DOCS, VOCAB)
    assert ret["stats0"][0][0] == full.v.total_length and ret["stats0"][0][1] == full.v.doccount


def test_eight_ranks_reproduce_the_reference_over_eight_shards(built):
    """The queries of the golden fixture the COMPILED REFERENCE answered through Xapiand's protocol over 8 shards, here over 8 gloo ranks:
    docid at every rank of the MSet, weight bits, max_possible (multi.h:38-73 unshard with n = 8, handler.cc:1532-1549)."""
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sharded8_and3_top10.json")))
    c = fx["corpus"]
    groups = {}
    for r in fx["results"]:
        groups.setdefault(r["query"]["first"] + r["query"]["maxitems"], []).append(r)
    batches = [([dict(op=r["query"]["op"], terms=r["query"]["terms"]) for r in rs], k) for k, rs in sorted(groups.items())]
    ret = run_ranks(8, batches, c["n_docs"], c["vocab"])
    for rank in range(8):
        assert ret["collectives%d" % rank] == len(batches)
        for (k, rs), got in zip(sorted(groups.items()), ret[rank]):
            for r, (rows, max_possible) in zip(rs, got):
                assert rows == [(d, float.fromhex(w)) for d, w, _ in r["hits"]], (rank, r["query"])
                assert max_possible == float.fromhex(r["max_possible"]), (rank, r["query"])
