"""Parity at BASELINE.json's sizes: the 10 M-document / 1 M-term synthetic index of configs C2 (AND-3 top-10),
C3 (OR-5 top-100), C5 (2-3-term PHRASE top-10) and the two-sided operators, searched through the C ABI in batch
mode (xgm_search_batch: the cost-model work decomposition over ~1 200 stripes, the global-threshold histogram of
the disjunction kernel, the merge kernel's large-survivor path) AND one query at a time (xgm_search: the latency
decomposition), against the CPU oracle on the very postings the device holds (copied back from HBM).  Queries are
the first 128 of bench.py's timed pool.  Bar: docid at every rank, fp64 weight bits, exact match count and
max_attained identical.  Reference behaviour: multiandpostlist.cc:150-207, orpostlist.cc:35-204,
exactphrasepostlist.cc:75-133, andmaybepostlist.cc:57-64 (oracle pinned to the compiled reference,
tests/test_oracle_vs_reference.py)."""
import ctypes as C
import os

import pytest

import helpers as H
from xapiand_amd import Database, Query, _lib, plan, search_batch

pytestmark = pytest.mark.gpu

N_DOCS = int(os.environ.get("XGM_CONFIG_DOCS", 10_000_000))
VOCAB = 1_000_000
N_CHECK = 128

CONFIGS = {
    "C2_and3_top10": dict(op="AND", terms=3, k=10),
    "C3_or5_top100": dict(op="OR", terms=5, k=100),
    "C5_phrase_top10": dict(op="PHRASE", terms=0, k=10),
    "and_not_2x2_top10": dict(op="AND_NOT", terms=4, required=2, k=10),
    "and_maybe_2x2_top10": dict(op="AND_MAYBE", terms=4, required=2, k=10),
    "filter_2x1_top10": dict(op="FILTER", terms=3, required=2, k=10),
}


@pytest.fixture(scope="module")
def big_db(built):
    db = Database.synthetic(H.CORPUS_SEED, N_DOCS, VOCAB, device=0)
    yield db
    db.close()


def gpu_rows(hits, hdr):
    return [(h.docid, h.weight, h.subqs_matched) for h in hits], dict(matches=hdr.matches_exact, max_attained=hdr.max_attained)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_config_scale_parity(big_db, name):
    cfg = CONFIGS[name]
    k = cfg["k"]
    queries = H.bench_pool(cfg["op"], cfg["terms"], cfg.get("required", 1), N_DOCS, VOCAB, maxitems=k)[100:100 + N_CHECK]
    ora = H.DeviceOracle(big_db, [t for q in queries for t in q["terms"]], positions=cfg["op"] == "PHRASE")
    want = H.oracle_search_batch(ora, queries, 0, k)
    plans = [plan(big_db, Query(q["op"], q["terms"], window=q.get("window", 0), n_required=q.get("n_required", 0)), 0, k) for q in queries]
    # batch mode
    got = search_batch(big_db, plans)
    n_nonempty = 0
    for q, (hits, hdr), (rows, oh) in zip(queries, got, want):
        g_rows, g_hdr = gpu_rows(hits, hdr)
        assert [(d, w) for d, w, _ in g_rows] == [(d, w) for d, w, _ in rows], (name, "batch", q)
        H.check_matches(g_hdr["matches"], oh["matches"], len(g_rows), (name, "batch matches", q))
        if rows:
            n_nonempty += 1
            assert g_hdr["max_attained"] == oh["max_attained"], (name, "batch max_attained", q)
            if cfg["op"] == "AND_MAYBE":
                assert [m for _, _, m in g_rows] == [m for _, _, m in rows], (name, "subqs", q)
    assert n_nonempty >= N_CHECK // 2, "config %s: most sampled queries should match something" % name
    # one query in flight
    L = _lib.lib()
    one_hits = (_lib.Hit * k)()
    one_hdr = _lib.ResultHdr()
    for q, p, (rows, oh) in zip(queries, plans, want):
        _lib.check(L.xgm_search(big_db._h, C.byref(p), one_hits, C.byref(one_hdr)))
        assert [(one_hits[j].docid, one_hits[j].weight) for j in range(one_hdr.n_hits)] == [(d, w) for d, w, _ in rows], (name, "single", q)
        H.check_matches(one_hdr.matches_exact, oh["matches"], one_hdr.n_hits, (name, "single matches", q))
    ora.close()


def test_phrase_reference_quirk_is_quantified(big_db):
    """DESIGN.md §7 / INTEGRATION.md: for PHRASE with maxitems < matches the reference's SelectPostList serves a stale
    cached weight (selectpostlist.cc:28-55), so its top-k differs from the intended one the device returns.  Count on
    the C5 query sample how many answers differ between the oracle WITH the quirk (= pinned to the compiled
    reference) and without it (= the device, asserted above); the number is quoted in INTEGRATION.md."""
    queries = H.bench_pool("PHRASE", 0, 1, N_DOCS, VOCAB, maxitems=10)[100:100 + N_CHECK]
    ora = H.DeviceOracle(big_db, [t for q in queries for t in q["terms"]], positions=True)
    intended = H.oracle_search_batch(ora, queries, 0, 10)
    quirk = H.oracle_search_batch(ora, queries, 0, 10, reference_select_bug=True)
    differ = sum(1 for (a, _), (b, _) in zip(intended, quirk) if [(d, w) for d, w, _ in a] != [(d, w) for d, w, _ in b])
    same_set = sum(1 for (a, _), (b, _) in zip(intended, quirk) if sorted(d for d, _, _ in a) == sorted(d for d, _, _ in b))
    out = os.path.join(H.ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "phrase_quirk_c5.txt"), "w") as f:
        f.write("C5 sample: %d queries; top-10 (docid, weight) differs from the reference's for %d; same docid SET for %d\n" % (len(queries), differ, same_set))
    ora.close()
    assert differ <= len(queries)


@pytest.mark.parametrize("name", ["C2_and3_top10", "C3_or5_top100"])
def test_sharded_parity_at_config_size(built, name):
    """C4's per-GPU workload on the hardware the tests get: TWO shards of 10 M documents each (the 20 M-document corpus split the
    reference's way, global doc g -> shard (g - 1) % 2: src/xapian/backends/multi.h:38-73) resident on the one MI355X, searched through
    xgm_search_sharded — merged statistics (enquire.cc:385-394), every shard planned with its own leaf order, the top-k exchange and
    the device merge (matcher.cc:653-781, mset.cc:367-395) — against Xapiand's protocol run on the CPU oracle over the very postings
    each shard holds in HBM.  128 queries of bench.py's pool for the 20 M-document corpus."""
    from xapiand_amd.enquire import search_sharded
    cfg = CONFIGS[name]
    k, world = cfg["k"], 2
    n_global = N_DOCS * world
    queries = H.bench_pool(cfg["op"], cfg["terms"], 1, n_global, VOCAB, maxitems=k)[100:100 + N_CHECK]
    dbs = [Database.synthetic(H.CORPUS_SEED, n_global, VOCAB, n_shards=world, shard=s, device=0) for s in range(world)]
    try:
        assert all(db.info().doccount == N_DOCS for db in dbs)
        terms = [t for q in queries for t in q["terms"]]
        oras = [H.DeviceOracle(db, terms) for db in dbs]
        infos = [db.info() for db in dbs]
        L = _lib.lib()
        got = search_sharded(dbs, [Query(q["op"], q["terms"]) for q in queries], 0, k)
        n_nonempty = 0
        for q, m in zip(queries, got):
            tfs = []
            for t in q["terms"]:
                tb, tot = t.encode(), 0
                for db in dbs:
                    tf = C.c_uint32()
                    _lib.check(L.xgm_lookup_term(db._h, tb, len(tb), None, C.byref(tf), None, None))
                    tot += tf.value
                tfs.append(tot)
            gs = dict(total_length=sum(i.total_length for i in infos), collection_size=sum(i.doccount for i in infos), has_positions=True, termfreq=tfs)
            allh, matches = [], 0
            for s, ora in enumerate(oras):
                hits, oh = H.oracle_search(ora, q["op"], q["terms"], 0, k, 0, gs)
                allh += [((d - 1) * world + s + 1, w) for d, w, _ in hits]
                matches += oh.matches
            allh.sort(key=lambda x: (-x[1], x[0]))
            assert [(i.docid, i.weight) for i in m] == allh[:k], (name, q)
            assert m.get_matches_exact() == matches, (name, "matches", q)
            n_nonempty += bool(allh)
        assert n_nonempty >= N_CHECK // 2
        for ora in oras:
            ora.close()
    finally:
        for db in dbs:
            db.close()


@pytest.mark.parametrize("name", ["C2_and3_top10", "C5_phrase_top10"])
def test_config_scale_parity_from_a_host_built_corpus(big_db, name):
    """The same configurations at full size with the oracle fed from the HOST: tools/xgm_corpus.h inverted on the host cores for the sampled queries' terms
    (helpers.TermsCorpus — every one of the 10 M documents generated, nothing read back from the device: VERDICT r5 weak #3; test_config_scale_parity above
    feeds the oracle the postings the device decoded).  A fault common to the device's builder and decoder would show here.  C5 in the
    reference-identical batch mode as well (XGM_REPLAY_BATCH_FROZEN against the oracle's reference mode)."""
    from xapiand_amd.enquire import search_batch_replay
    cfg = CONFIGS[name]
    k = cfg["k"]
    n = 32 if cfg["op"] != "PHRASE" else 24
    queries = H.bench_pool(cfg["op"], cfg["terms"], cfg.get("required", 1), N_DOCS, VOCAB, maxitems=k)[100:100 + n]
    corpus = H.TermsCorpus(N_DOCS, VOCAB, [t for q in queries for t in q["terms"]], positions=cfg["op"] == "PHRASE")
    info = big_db.info()
    assert (corpus.v.doccount, corpus.v.total_length) == (info.doccount, info.total_length)
    corpus.warm()
    want = H.oracle_search_batch(corpus, queries, 0, k)
    plans = [plan(big_db, Query(q["op"], q["terms"], window=q.get("window", 0)), 0, k) for q in queries]
    got = search_batch(big_db, plans)
    nonempty = 0
    for q, (hits, hdr), (rows, oh) in zip(queries, got, want):
        g_rows, g_hdr = gpu_rows(hits, hdr)
        assert [(d, w) for d, w, _ in g_rows] == [(d, w) for d, w, _ in rows], (name, "host-built corpus", q)
        H.check_matches(g_hdr["matches"], oh["matches"], len(g_rows), (name, "host-built corpus, matches", q))
        nonempty += bool(rows)
    assert nonempty >= n // 2
    if cfg["op"] == "PHRASE":
        ref = H.oracle_search_batch(corpus, queries, 0, k, reference_select_bug=True)
        for q, (page, hdr, _), (rows, _) in zip(queries, search_batch_replay(big_db, plans), ref):
            assert [(d, w) for d, w, _ in page] == [(d, w) for d, w, _ in rows], (name, "reference mode, host-built corpus", q)
    corpus.close()
