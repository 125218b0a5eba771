"""Heterogeneous batches: one xgm_search_batch call mixing AND / OR / PHRASE / AND_NOT / AND_MAYBE /
FILTER / single-term queries with different k is cut into one launch per kernel class and must answer every query
exactly like the oracle — and like the same queries searched one at a time."""
import random

import pytest

import helpers as H
from xapiand_amd import Database, Query, plan, search_batch

pytestmark = pytest.mark.gpu

N_DOCS, VOCAB = 150000, 100000


def test_mixed_batch_matches_oracle(built, tmp_path):
    c = H.Corpus(N_DOCS, VOCAB)
    db = Database(c.build_segment(str(tmp_path / "m.seg")))
    qs = (H.gen_term_queries("AND", 30, 3, 1, 500, maxitems=10, seed=1) + H.gen_term_queries("OR", 20, 5, 1, 4000, maxitems=100, seed=2) +
          H.gen_phrase_queries(16, N_DOCS, VOCAB, seed=3) + H.gen_sided_queries("AND_NOT", 10, 2, 2, 1, 300, maxitems=10, seed=4) +
          H.gen_sided_queries("AND_MAYBE", 10, 2, 2, 1, 300, maxitems=20, seed=5) + H.gen_sided_queries("FILTER", 8, 2, 1, 1, 300, maxitems=10, seed=6) +
          H.gen_term_queries("AND", 6, 1, 1, 200, maxitems=10, seed=7) + H.gen_term_queries("AND", 4, 2, 1, 100, maxitems=300, seed=8) +
          H.gen_term_queries("OR", 4, 2, 1, 100, first=5, maxitems=7, seed=9))
    random.Random(5).shuffle(qs)
    plans = [plan(db, Query(q["op"], q["terms"], window=q.get("window", 0), n_required=q.get("n_required", 0)), q["first"], q["maxitems"]) for q in qs]
    got = search_batch(db, plans)
    for q, p, (hits, hdr) in zip(qs, plans, got):
        want, oh = H.oracle_search(c, q["op"], q["terms"], q["first"], q["maxitems"], q.get("window", 0), n_required=q.get("n_required", 0))
        assert [(h.docid, h.weight) for h in hits] == [(d, w) for d, w, _ in want], q
        H.check_matches(hdr.matches_exact, oh.matches, len(hits), q)
        (h1, hdr1), = search_batch(db, [p])
        assert [(h.docid, h.weight, h.subqs_matched) for h in h1] == [(h.docid, h.weight, h.subqs_matched) for h in hits], q
    db.close()


def test_server_mode_batching_queue(built, tmp_path):
    """xgm_index_set_batching: single-query calls from many threads share launches (a dispatcher thread owned by the
    index), every caller gets exactly its own answer, a declined query fails alone, and switching it off restores the
    direct path."""
    import ctypes as C
    import threading
    from xapiand_amd import _lib
    c = H.Corpus(N_DOCS, VOCAB)
    db = Database(c.build_segment(str(tmp_path / "s.seg")))
    L = _lib.lib()
    qs = (H.gen_term_queries("AND", 60, 3, 1, 500, maxitems=10, seed=21) + H.gen_term_queries("OR", 30, 4, 1, 3000, maxitems=20, seed=22) +
          H.gen_phrase_queries(20, N_DOCS, VOCAB, seed=23) + H.gen_sided_queries("AND_MAYBE", 10, 2, 2, 1, 300, maxitems=10, seed=24))
    plans = [plan(db, Query(q["op"], q["terms"], window=q.get("window", 0), n_required=q.get("n_required", 0)), q["first"], q["maxitems"]) for q in qs]
    want = [[(d, w) for d, w, _ in H.oracle_search(c, q["op"], q["terms"], q["first"], q["maxitems"], q.get("window", 0), n_required=q.get("n_required", 0))[0]] for q in qs]
    _lib.check(L.xgm_index_set_batching(db._h, 64))
    got = [None] * len(qs)
    errs = []

    def worker(t, n_threads):
        hits = (_lib.Hit * 32)()
        hdr = _lib.ResultHdr()
        for i in range(t, len(qs), n_threads):
            for _ in range(3):
                rc = L.xgm_search(db._h, C.byref(plans[i]), hits, C.byref(hdr))
                if rc != 0:
                    errs.append((i, rc))
                got[i] = [(hits[j].docid, hits[j].weight) for j in range(hdr.n_hits)]
    threads = [threading.Thread(target=worker, args=(t, 16)) for t in range(16)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs
    assert got == want
    info = (C.c_uint64 * 3)()
    _lib.check(L.xgm_debug_batching_info(db._h, info))
    assert info[1] == 3 * len(qs) and info[0] < info[1]            # fewer launches than requests: calls really shared batches
    _lib.check(L.xgm_index_set_batching(db._h, 0))
    (h1, _), = search_batch(db, [plans[0]])
    assert [(h.docid, h.weight) for h in h1] == want[0]
    db.close()
