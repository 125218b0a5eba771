"""CPU-only: the host side of the path — xgm_plan_query (reference: BM25Weight::init, MultiAndPostList leaf
order, OrContext's Huffman tree, Enquire::get_mset's clamping, get_maxpart summation) — on a HOST-ONLY index
(xgm_index_open with XGM_DEVICE_NONE: dictionary and statistics, no HBM).  Every plan's max_possible must equal,
bit for bit, what the REAL reference reported in the committed golden fixtures; the planned leaf order and
summation program must be what the oracle restates; and a search on such an index must fail loudly."""
import json
import os

import pytest

import helpers as H
from xapiand_amd import Database, Query, _lib
from xapiand_amd.enquire import BM25Weight, merged_stats, plan, search_batch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def host_env(built, tmp_path_factory):
    d = tmp_path_factory.mktemp("planner")
    c1 = H.Corpus(10000, 1000000)
    db1 = Database(c1.build_segment(str(d / "c1.seg")), device=_lib.XGM_DEVICE_NONE)
    shards = [H.Corpus(10000, 1000000, n_shards=4, shard=s) for s in range(4)]
    dbs = [Database(c.build_segment(str(d / ("s%d.seg" % s))), device=_lib.XGM_DEVICE_NONE) for s, c in enumerate(shards)]
    yield c1, db1, shards, dbs
    db1.close()
    for x in dbs:
        x.close()


def q_of(q):
    return Query(q["op"], q["terms"], window=q.get("window", 0), n_required=q.get("n_required", 0))


@pytest.mark.parametrize("name", ["c1_and3_top10", "or5_top100", "and_paging", "phrase_full", "sided_top10"])
def test_max_possible_matches_reference(host_env, name):
    _, db1, _, _ = host_env
    fx = json.load(open(os.path.join(GOLDEN_DIR, name + ".json")))
    for r in fx["results"]:
        q = r["query"]
        p = plan(db1, q_of(q), q["first"], min(q["maxitems"], 1000))      # (max_possible does not depend on maxitems)
        assert p.max_possible == float.fromhex(r["max_possible"]), q


def test_sharded_plans_use_merged_statistics(host_env):
    """Per-shard plans with the merged statistics: the reference's merged MSet reports the max over the shards."""
    _, _, _, dbs = host_env
    fx = json.load(open(os.path.join(GOLDEN_DIR, "sharded4_and3_top10.json")))
    for r in fx["results"]:
        q = r["query"]
        gs = merged_stats(dbs, q_of(q))
        mp = max(plan(db, q_of(q), 0, q["first"] + q["maxitems"], global_stats=gs).max_possible for db in dbs)
        assert mp == float.fromhex(r["max_possible"]), q


def test_plan_shapes(host_env):
    c1, db1, _, _ = host_env
    # AND: leaves in ascending shard-local termfreq (MultiAndPostList), left-deep chain
    p = plan(db1, Query("AND", ["t3", "t200", "t17"]), 0, 10)
    tfs = [c1.termfreq(t) for t in ("t3", "t200", "t17")]
    order = [p.terms[i].phrase_index for i in range(3)]
    assert [tfs[i] for i in order] == sorted(tfs)
    assert list(p.sum_prog[:p.sum_len]) == [0, 1, -1, 2, -1]
    assert p.req_mask == 0b111 and p.neg_mask == 0
    # OR: query order, a full binary tree over the 4 leaves
    p = plan(db1, Query("OR", ["t3", "t200", "t17", "t9"]), 0, 10)
    assert [p.terms[i].phrase_index for i in range(4)] == [0, 1, 2, 3] and p.sum_len == 7 and p.req_mask == 0
    # AND_NOT: required terms first (by termfreq), excluded ones after; only the left side is summed
    p = plan(db1, Query("AND_NOT", ["t200", "t3", "t17"], n_required=2), 0, 10)
    assert p.req_mask == 0b011 and p.neg_mask == 0b100 and list(p.sum_prog[:p.sum_len]) == [0, 1, -1]
    # AND_MAYBE: chain + OR tree of the optional terms
    p = plan(db1, Query("AND_MAYBE", ["t200", "t3", "t17"], n_required=1), 0, 10)
    assert p.req_mask == 0b001 and p.neg_mask == 0 and p.sum_len == 5
    # FILTER: a conjunction whose right-hand leaves weigh nothing, merged in by termfreq
    p = plan(db1, Query("FILTER", ["t200", "t3"], n_required=1), 0, 10)
    assert p.req_mask == 0b11 and sorted(p.terms[i].termweight == 0.0 for i in range(2)) == [False, True]
    # PHRASE of up to 8 terms and NEAR are planned like a conjunction with a positional filter
    p = plan(db1, Query("PHRASE", ["t1", "t2", "t3", "t4"]), 0, 10)
    assert p.phrase_active == 1 and p.window == 4 and p.req_mask == 0b1111
    p = plan(db1, Query("NEAR", ["t1", "t2", "t3"], window=7), 0, 10)
    assert p.op == _lib.XGM_OP_NEAR and p.phrase_active == 1 and p.window == 7
    # declined shapes: repeated term, phrase of more than 8 terms, first + maxitems beyond the device top-k
    for bad, first, k in ((Query("AND", ["t3", "t3"]), 0, 10), (Query("PHRASE", ["t%d" % i for i in range(1, 10)]), 0, 10),
                          (Query("AND", ["t3", "t17"]), 1000, 100)):
        with pytest.raises(_lib.XgmUnsupported):
            plan(db1, bad, first, k)


def test_host_only_index_cannot_search(host_env):
    _, db1, _, _ = host_env
    p = plan(db1, Query("AND", ["t3", "t17"]), 0, 10)
    with pytest.raises(_lib.XgmError) as e:
        search_batch(db1, [p])
    assert e.value.code == _lib.XGM_E_NO_DEVICE


def test_static_match_bounds_equal_the_references(built, tmp_path):
    """The planner's get_termfreq_min / _est / _max of the tree the reference would build: the UPPER bound
    of every MSet equals the compiled reference's for all operators and nested trees, the static lower bound never
    exceeds the reference's (which adds the documents its matcher happened to weigh), and xgm_mset_bounds keeps
    lower <= estimate <= upper."""
    import ctypes as C
    if not H.have_xapian_ref():
        pytest.skip("oracle/_ref not built")
    n_docs, vocab = 12000, 20000
    db = str(tmp_path / "db")
    H.xapian_ref("build", db, hex(H.CORPUS_SEED), n_docs, vocab, 50, 150)
    c = H.Corpus(n_docs, vocab)
    hdb = Database(c.build_segment(str(tmp_path / "c.seg")), device=_lib.XGM_DEVICE_NONE)
    qs = (H.gen_term_queries("AND", 30, 3, 1, 200, maxitems=10, seed=1) + H.gen_term_queries("OR", 30, 4, 1, 2000, maxitems=10, seed=3) +
          H.gen_sided_queries("AND_NOT", 12, 2, 2, 1, 100, seed=4) + H.gen_sided_queries("AND_MAYBE", 12, 2, 2, 1, 100, seed=5) +
          H.gen_sided_queries("FILTER", 12, 2, 2, 1, 100, seed=6) + H.gen_phrase_queries(16, n_docs, vocab, seed=7) +
          H.gen_phrase_queries(8, n_docs, vocab, seed=9, window_extra=3, op="NEAR") + H.gen_tree_queries(48, 1, 300, seed=10))
    qf, of = str(tmp_path / "q.txt"), str(tmp_path / "o.txt")
    H.write_queries(qf, qs)
    H.xapian_ref("query", qf, of, db)
    n_full = 0
    for q, r in zip(qs, H.parse_ref_output(of)):
        pq = Query.tree(q["tree"]) if q["op"] == "RPN" else Query(q["op"], q["terms"], window=q.get("window", 0), n_required=q.get("n_required", 0))
        p = plan(hdb, pq, q["first"], q["maxitems"])
        assert p.est_min <= p.est_est <= p.est_max, q
        if r["n"] == q["maxitems"]:                      # a full MSet: the reference reports its tree's static upper bound
            n_full += 1
            assert p.est_max == r["ub"], (q, p.est_max, r["ub"])
            assert p.est_min <= r["lb"], q
        hdr = _lib.ResultHdr()
        hdr.n_hits = r["n"]
        lb, est, ub = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _lib.lib().xgm_mset_bounds(C.byref(p), C.byref(hdr), C.byref(lb), C.byref(est), C.byref(ub))
        assert lb.value <= est.value <= ub.value and lb.value <= r["lb"] and ub.value == r["ub"], (q, lb.value, est.value, ub.value, r)
    assert n_full > 100


def test_disjunction_threshold_guess_is_conservative(built, tmp_path):
    """The planner's guess of a disjunction's final k-th weight (xgm_dev_query::theta_seed, xgm_api.cc or_theta_seed) only steers
    which documents the kernel weighs first — but it earns its keep only if it is (a) rarely above the true k-th weight (then the
    kernel must go round again) and (b) not far below it.  On the synthetic corpus: never above, and within a factor of two."""
    import ctypes as C
    c = H.Corpus(60000, 200000)
    db = Database(c.build_segment(str(tmp_path / "seed.seg")), device=_lib.XGM_DEVICE_NONE)
    L = _lib.lib()
    L.xgm_debug_or_bounds.argtypes = [C.c_void_p, C.POINTER(_lib.Query), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    ratios = []
    for k in (10, 100):
        for q in H.gen_term_queries("OR", 20, 5, 8, 4096, seed=77):
            p = plan(db, Query("OR", q["terms"]), 0, k)
            seed, ub, ub1 = C.c_double(), (C.c_double * 16)(), (C.c_double * 16)()
            assert L.xgm_debug_or_bounds(db._h, C.byref(p), C.byref(seed), ub, ub1) == 0
            want, _ = H.oracle_search(c, "OR", q["terms"], 0, k)
            for i in range(len(q["terms"])):
                assert 0.0 < ub1[i] <= ub[i] <= p.terms[i].termweight * 1.00000001      # wdf = 1 bound <= largest-wdf bound <= termweight
            if len(want) == k and seed.value > 0.0:
                ratios.append(seed.value / want[k - 1][1])
    assert len(ratios) >= 30 and max(ratios) <= 1.0 and min(ratios) >= 0.5, (len(ratios), min(ratios), max(ratios))
    db.close()


def test_disjunction_term_bounds_dominate_every_posting(built, tmp_path):
    """xgm_orw_kernel may skip a document only because a sum of per-term bounds stays below the running k-th weight, so the bounds
    the planner hands it (xgm_dev_query::ub — any wdf; ub1 — wdf = 1; xgm_api.cc to_dev_query) must dominate the BM25 weight
    (bm25weight.cc:170-181) of EVERY posting of the term.  Checked exhaustively over the postings of sampled terms, for the default
    parameters and for b = 0 / k1 = 0 / a large min_normlen, where the length normalisation degenerates."""
    import ctypes as C
    import numpy as np
    c = H.Corpus(60000, 200000)
    db = Database(c.build_segment(str(tmp_path / "bounds.seg")), device=_lib.XGM_DEVICE_NONE)
    L = _lib.lib()
    L.xgm_debug_or_bounds.argtypes = [C.c_void_p, C.POINTER(_lib.Query), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    doclen = np.ctypeslib.as_array(c.v.doclen, shape=(c.v.lastdocid + 1,)).astype(np.float64)
    names = c.terms()
    index_of = {t: i for i, t in enumerate(names)}
    checked = 0
    for params in ({}, {"b": 0.0}, {"k1": 0.0}, {"min_normlen": 3.0}, {"k1": 2.0, "b": 0.9}):
        for q in H.gen_term_queries("OR", 6, 5, 8, 4096, seed=5):
            p = plan(db, Query("OR", q["terms"]), 0, 10, weight=BM25Weight(**params))
            seed, ub, ub1 = C.c_double(), (C.c_double * 16)(), (C.c_double * 16)()
            assert L.xgm_debug_or_bounds(db._h, C.byref(p), C.byref(seed), ub, ub1) == 0
            for i, t in enumerate(q["terms"]):
                tb = t if isinstance(t, bytes) else t.encode()
                did, wdf = c.term_postings(index_of[tb])
                w = wdf.astype(np.float64)
                normlen = np.maximum(doclen[did] * p.len_factor, p.min_normlen)
                weight = p.terms[i].termweight * (w / (p.k1 * (normlen * p.b + (1.0 - p.b)) + w))
                assert weight.max() <= ub[i], (params, t, weight.max(), ub[i])
                one = wdf == 1
                if one.any():
                    assert weight[one].max() <= ub1[i], (params, t, weight[one].max(), ub1[i])
                checked += len(did)
    assert checked > 100000
    db.close()


def test_prefix_expansion_walks_the_dictionary_in_term_order(host_env):
    """xgm_expand_prefix / xgm_term_info (what the matcher hook expands OP_WILDCARD / OP_EDIT_DISTANCE with): the terms under a prefix
    in byte order — the order Database::allterms_begin(prefix) walks, Context<T>::expand_wildcard's input (reference
    src/xapian/api/queryinternal.cc:246-315) — with their strings and shard-local term frequencies; the count is exact whatever the
    caller's capacity is."""
    import ctypes as C
    c1, db1, _, _ = host_env
    L = _lib.lib()
    terms = c1.terms()
    df = c1.df_array()
    tf_of = {t: int(df[i]) for i, t in enumerate(terms)}
    for prefix in (b"t12", b"t9", b"t1999", b"t", b"t77777", b"", b"zz", b"t0"):
        want = sorted(t for t in terms if t.startswith(prefix))
        n_total = C.c_uint32()
        small = (C.c_uint32 * 5)()
        assert L.xgm_expand_prefix(db1._h, prefix, len(prefix), 5, small, C.byref(n_total)) == 0
        assert n_total.value == len(want), (prefix, n_total.value, len(want))
        ids = (C.c_uint32 * max(1, len(want)))()
        assert L.xgm_expand_prefix(db1._h, prefix, len(prefix), len(want), ids, C.byref(n_total)) == 0 and n_total.value == len(want)
        got = []
        for i in range(min(len(want), 4000)):
            b, bl, tf = C.c_char_p(), C.c_size_t(), C.c_uint32()
            assert L.xgm_term_info(db1._h, ids[i], C.byref(b), C.byref(bl), C.byref(tf), None) == 0
            t = C.string_at(b, bl.value)
            got.append(t)
            assert tf.value == tf_of[t], (t, tf.value, tf_of[t])
        assert got == want[:len(got)], (prefix, got[:5], want[:5])
        assert list(small[:min(5, len(want))]) == list(ids[:min(5, len(want))])
