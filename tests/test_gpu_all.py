"""xgm_search_all: EVERY match of a query in ascending docid order with its weight (include/xgm.h) — what lifts the 1 024-match
ceiling of the hook's byte-compatible modes.  Checked against the oracle's FULL ranking (first + maxitems = the
shard's document count) put into docid order: docids, weight bit patterns, weighted-leaf counts, the count, the best weight.
Every operator class; matches far beyond one device page.  Also runs under the CPU emulation (tests/test_emu.py)."""
import os

import pytest

import helpers as H
from xapiand_amd import Database, Query
from xapiand_amd.enquire import plan, search_all

pytestmark = [pytest.mark.gpu]

QUICK = bool(os.environ.get("XGM_EMU_QUICK"))


def full_match_in_docid_order(c, q, n_docs):
    want, oh = H.oracle_search(c, q["op"], q["terms"], 0, n_docs, q.get("window", 0), n_required=q.get("n_required", 0))
    return sorted(want), oh


@pytest.mark.parametrize("stripe_bits", [0, 10])
def test_all_matches_in_docid_order_vs_oracle(built, tmp_path, stripe_bits):
    n_docs, vocab = (3000, 8000) if QUICK else (60000, 60000)
    c = H.Corpus(n_docs, vocab)
    db = Database(c.build_segment(str(tmp_path / "s.seg"), stripe_bits=stripe_bits))
    n = (lambda full, quick: quick if QUICK else full)
    qs = (H.gen_term_queries("OR", n(10, 2), 5, 1, 3000, seed=71) + H.gen_term_queries("OR", n(4, 1), 3, 1, 12, seed=72) +
          H.gen_term_queries("AND", n(10, 2), 3, 1, 60, seed=73) + H.gen_term_queries("AND", n(4, 1), 2, 1, 6, seed=74) +
          H.gen_sided_queries("AND_MAYBE", n(6, 1), 1, 2, 1, 200, seed=75) + H.gen_sided_queries("AND_NOT", n(6, 1), 1, 2, 1, 200, seed=76) +
          H.gen_sided_queries("FILTER", n(4, 1), 2, 1, 1, 100, seed=77) +
          H.gen_phrase_queries(n(10, 2), n_docs, vocab, seed=78) + H.gen_phrase_queries(n(4, 1), n_docs, vocab, seed=79, window_extra=3) +
          H.gen_phrase_queries(n(4, 1), n_docs, vocab, seed=80, window_extra=4, op="NEAR") +
          [dict(op="AND", terms=["t1"], first=0, maxitems=10), dict(op="AND", terms=["t2", "nonexistent"], first=0, maxitems=10)])
    n_big = n_total = 0
    for q in qs:
        want, oh = full_match_in_docid_order(c, q, n_docs)
        p = plan(db, Query(q["op"], q["terms"], window=q.get("window", 0), n_required=q.get("n_required", 0)), q["first"], q["maxitems"])
        got, hdr = search_all(db, p)
        assert got == want, (q, len(got), len(want))
        assert hdr.matches_exact == len(want) == oh.matches and hdr.n_hits == len(want), q
        if want:
            assert hdr.max_attained == oh.max_attained and hdr.max_weight_subqs_matched == oh.max_subqs, q
        n_total += len(want)
        n_big += len(want) > 1024
    assert n_big >= (2 if QUICK else 12) and n_total > (2000 if QUICK else 200000), (n_big, n_total)
    # a buffer that is too small: nothing written, the count says how much room is needed
    q = qs[0]
    p = plan(db, Query(q["op"], q["terms"]), 0, 10)
    got, hdr = search_all(db, p, cap=5)
    assert got is None and hdr.matches_exact == len(full_match_in_docid_order(c, q, n_docs)[0]) > 5
    db.close()
    c.close()


def test_all_matches_of_nested_trees(built, tmp_path):
    n_docs, vocab = (3000, 8000) if QUICK else (40000, 50000)
    c = H.Corpus(n_docs, vocab)
    db = Database(c.build_segment(str(tmp_path / "t.seg")))
    n_total = 0
    for q in H.gen_tree_queries(4 if QUICK else 30, 1, 600, seed=81):
        rows, oh, _ = H.oracle_search_tree(c, q["tree"], 0, n_docs)
        p = plan(db, Query.tree(q["tree"]), 0, 10)
        got, hdr = search_all(db, p)
        assert got == sorted(rows), q["tree"]
        assert hdr.matches_exact == oh.matches, q["tree"]
        n_total += len(rows)
    assert n_total > (500 if QUICK else 20000)
    db.close()
    c.close()


# ---- xgm_search_replay: ProtoMSet's collation on the device (round 5) -------------------------------------------------------

def known_matching_docs(weights, k, check_at_least):
    """The library's HOST restatement (xgm_known_matching_docs, pinned to the compiled reference's own figures by
    tests/test_oracle_vs_reference.py::test_known_matching_docs_is_a_function_of_the_match_in_docid_order)."""
    import ctypes as C
    from xapiand_amd import _lib
    L = _lib.lib()
    L.xgm_known_matching_docs.restype = C.c_uint64
    L.xgm_known_matching_docs.argtypes = [C.POINTER(C.c_double), C.c_uint64, C.c_uint32, C.c_uint32]
    arr = (C.c_double * max(1, len(weights)))(*weights)
    return L.xgm_known_matching_docs(arr, len(weights), k, check_at_least)


def host_frozen_replay(matches, conj, k, check_at_least):
    """What integration/xgm_matcher_hook.cc replayed on the host in round 4 (the loop pinned hook on == hook off against the compiled
    reference): ProtoMSet over the positional matches in docid order with their true weights until min_weight turns positive, then the
    weight of the next document of the underlying conjunction for every later match (selectpostlist.cc:28-55).
    matches / conj: [(docid, weight, subqs)] ascending.  Returns (page in rank order, known_matching_docs, max_weight, max_subqs)."""
    before = lambda a, b: a[1] > b[1] or (a[1] == b[1] and a[0] < b[0])
    results, known, min_w, max_w, max_m, heap_built = [], 0, 0.0, 0.0, 0, False

    def add(item):
        nonlocal known, min_w, max_w, max_m, heap_built
        known += 1
        if item[1] > max_w:
            max_w, max_m = item[1], item[2]
        if item[1] < min_w:
            return
        if len(results) < k:
            results.append(item)
            return
        if k == 0:
            return
        worst = max(range(k), key=lambda i: (-results[i][1], results[i][0]))
        if not heap_built:
            heap_built = True
            if known >= check_at_least:
                min_w = results[worst][1]
        if not before(item, results[worst]):
            return
        results[worst] = item
        worst = max(range(k), key=lambda i: (-results[i][1], results[i][0]))
        if known >= check_at_least:
            min_w = results[worst][1]

    w_star, i = None, 0
    while i < len(matches):
        h = matches[i]
        i += 1
        add(h)
        if min_w > 0.0:
            nxt = [c for c in conj if c[0] > h[0]]
            w_star = nxt[0][1] if nxt else None
            break
    if min_w > 0.0 and w_star is not None:
        for h in matches[i:]:
            if w_star < min_w:
                break
            add((h[0], w_star, h[2]))
    return sorted(results, key=lambda r: (-r[1], r[0])), known, max_w, max_m


def test_replay_counts_what_protomset_counts(built, tmp_path):
    """XGM_REPLAY_COUNT: the page ProtoMSet keeps and its known_matching_docs from the device, for every operator class, pages of
    several sizes and check_at_least inside, at and far beyond the page — against the whole match (xgm_search_all) put through the host
    restatement that is pinned to the compiled reference.  The whole match comes from the ORACLE (its full ranking put into docid order), not from the
    device: what is compared with the reference's figures is the device's replay alone."""
    from xapiand_amd.enquire import search_replay
    n_docs, vocab = (3000, 8000) if QUICK else (60000, 60000)
    c = H.Corpus(n_docs, vocab)
    db = Database(c.build_segment(str(tmp_path / "r.seg")))
    n = (lambda full, quick: quick if QUICK else full)
    qs = (H.gen_term_queries("OR", n(8, 2), 5, 1, 3000, seed=171) + H.gen_term_queries("OR", n(4, 1), 3, 1, 12, seed=172) +
          H.gen_term_queries("AND", n(6, 2), 3, 1, 60, seed=173) + H.gen_term_queries("AND", n(3, 1), 2, 1, 6, seed=174) +
          H.gen_sided_queries("AND_MAYBE", n(4, 1), 1, 2, 1, 200, seed=175) + H.gen_sided_queries("AND_NOT", n(4, 1), 1, 2, 1, 200, seed=176) +
          H.gen_phrase_queries(n(4, 1), n_docs, vocab, seed=178) +
          [dict(op="AND", terms=["t1"], first=0, maxitems=10), dict(op="AND", terms=["t2", "nonexistent"], first=0, maxitems=10)])
    shapes = [(0, 10, 0), (0, 1, 0), (5, 20, 0), (0, 100, 0), (0, 10, 40), (0, 10, 300), (0, 10, 10 ** 6), (0, 1000, 0), (0, 0, 0), (3, 0, 50)]
    events = big = 0
    for qi, q in enumerate(qs):
        query = Query(q["op"], q["terms"], window=q.get("window", 0), n_required=q.get("n_required", 0))
        every, _ = full_match_in_docid_order(c, q, n_docs)         # the ORACLE's whole match in docid order (round 6; round 5 took the device's own list)
        for first, maxitems, cal in (shapes if not QUICK else shapes[qi % 3::3]):
            k = first + maxitems
            cal_eff = max(cal, k)                                   # Enquire::get_mset: check_at_least = max(check_at_least, first + maxitems)
            p = plan(db, query, first, maxitems, check_at_least=cal_eff)
            page, hdr, known = search_replay(db, p)
            want_page = sorted(every, key=lambda r: (-r[1], r[0]))[:k]
            assert page == want_page, (q, first, maxitems, cal)
            assert known == known_matching_docs([r[1] for r in every], k, cal_eff), (q, first, maxitems, cal, known, len(every))
            assert hdr.matches_exact == len(every) and hdr.n_hits == len(want_page), q
            if every:
                best = max(every, key=lambda r: (r[1], -r[0]))
                assert hdr.max_attained == best[1] and hdr.max_weight_subqs_matched == best[2], q
            events += known < len(every)
            big += len(every) > 2048                               # (more than one block of the replay kernel)
    assert events >= (3 if QUICK else 40) and big >= (1 if QUICK else 10), (events, big)
    # which formulation walked the lists: one workgroup, or segments in parallel (lists of >= 4 x 4 096 entries — or whatever
    # XGM_REPLAY_SEG_MIN says: tests/test_gpu_variants.py and tests/test_emu.py run this test with small segments)
    import ctypes as C
    from xapiand_amd import _lib
    info = (C.c_uint64 * 2)()
    _lib.lib().xgm_debug_replay_info.argtypes = [C.POINTER(C.c_uint64)]
    _lib.lib().xgm_debug_replay_info(info)
    if os.environ.get("XGM_REPLAY_SEG_MIN"):
        assert info[1] >= (5 if QUICK else 10), list(info)
    db.close()
    c.close()


@pytest.mark.parametrize("stripe_bits", [0, 10])
def test_replay_freezes_the_weight_as_selectpostlist_does(built, tmp_path, stripe_bits):
    """XGM_REPLAY_FROZEN_WEIGHT: page, weights and count of the REFERENCE for PHRASE / windowed PHRASE / NEAR of 2-6 terms (beyond 3
    terms the workgroup kernel takes a stripe in several passes) — against (a) the oracle's reference mode, which is pinned to the
    compiled reference (check_at_least inside the page), (b) the host replay of round 4's hook over the positional match and the
    underlying conjunction, for check_at_least beyond the page too."""
    from xapiand_amd.enquire import search_replay, REPLAY_FROZEN_WEIGHT
    n_docs, vocab = (3000, 300) if QUICK else (40000, 3000)             # (a small vocabulary: phrases with hundreds of matches)
    c = H.Corpus(n_docs, vocab)
    db = Database(c.build_segment(str(tmp_path / "f.seg"), stripe_bits=stripe_bits))
    n = (lambda full, quick: quick if QUICK else full)
    qs = (H.gen_phrase_queries(n(12, 3), n_docs, vocab, seed=181) + H.gen_phrase_queries(n(6, 1), n_docs, vocab, seed=182, window_extra=3) +
          H.gen_phrase_queries(n(6, 1), n_docs, vocab, seed=183, window_extra=4, op="NEAR") +
          H.gen_phrase_queries(n(8, 2), n_docs, vocab, seed=184, lengths=(4, 5, 6)) +
          H.gen_phrase_queries(n(4, 1), n_docs, vocab, seed=185, lengths=(4, 5), window_extra=3, op="NEAR"))
    froze = differs = long_ok = 0
    for qi, q in enumerate(qs):
        query = Query(q["op"], q["terms"], window=q.get("window", 0))
        matches, _ = search_all(db, plan(db, query, 0, 10))
        conj, _ = search_all(db, plan(db, Query("AND", q["terms"]), 0, 10))
        shapes = [(0, 10, 0), (0, 3, 0), (2, 5, 0), (0, 10, 25), (0, 1, 0), (0, 10, 10 ** 6)]
        for first, maxitems, cal in (shapes if not QUICK else shapes[qi % 2::2]):
            k = first + maxitems
            cal_eff = max(cal, k)
            p = plan(db, query, first, maxitems, check_at_least=cal_eff)
            page, hdr, known = search_replay(db, p, REPLAY_FROZEN_WEIGHT)
            want_page, want_known, want_max, want_m = host_frozen_replay(matches, conj, k, cal_eff)
            assert page == want_page, (q, first, maxitems, cal, page[:3], want_page[:3])
            assert known == want_known and hdr.matches_exact == len(matches), (q, first, maxitems, cal, known, want_known)
            if matches:
                assert hdr.max_attained == want_max and hdr.max_weight_subqs_matched == want_m, q
            if cal == 0:
                ref, _ = H.oracle_search(c, q["op"], q["terms"], first, maxitems, q.get("window", 0), reference_select_bug=True)
                assert page == ref, (q, first, maxitems)          # (the oracle, like the device, hands out the first + maxitems best: the caller drops `first`)
            intended = sorted(matches, key=lambda r: (-r[1], r[0]))[:k]
            froze += len(matches) > k
            differs += page != intended
            long_ok += len(q["terms"]) > 3 and len(matches) > 0
    assert froze >= (2 if QUICK else 30) and differs >= (1 if QUICK else 10) and long_ok >= (1 if QUICK else 8), (froze, differs, long_ok)
    db.close()
    c.close()
