"""The reference-identical answer of positional queries INSIDE a batch (include/xgm.h, XGM_REPLAY_BATCH_FROZEN): xgm_andw_list_kernel's units
list their first matches in docid order with their successors in the conjunction, xgm_frozen_finish_kernel replays ProtoMSet +
SelectPostList's frozen weight over them (protomset.h:340-400, selectpostlist.cc:28-55).  Checked against (a) xgm_search_replay, the
one-query-per-call form that walks the WHOLE match (itself pinned to the compiled reference through the hook tests), (b) the oracle's
reference mode (pinned to the compiled reference, tests/test_oracle_vs_reference.py).  Also runs under the CPU emulation (tests/test_emu.py)."""
import ctypes as C
import os

import pytest

import helpers as H
from xapiand_amd import Database, Query, _lib
from xapiand_amd.enquire import plan, search_batch, search_batch_replay, search_replay, REPLAY_FROZEN_WEIGHT

pytestmark = [pytest.mark.gpu]

QUICK = bool(os.environ.get("XGM_EMU_QUICK"))
NO_LIST = bool(os.environ.get("XGM_NO_LIST_KERNEL"))      # the variant run: every row is answered when the batch is collected
LB = _lib.XGM_KNOWN_LOWER_BOUND


def replay_info():
    info = (C.c_uint64 * 3)()
    _lib.lib().xgm_debug_batch_replay_info(info)
    return list(info)


@pytest.mark.parametrize("stripe_bits", [0, 10])
def test_batch_frozen_equals_the_per_query_replay_and_the_reference(built, tmp_path, stripe_bits):
    n_docs, vocab = (3000, 300) if QUICK else (40000, 3000)             # (a small vocabulary: phrases with hundreds of matches)
    c = H.Corpus(n_docs, vocab)
    db = Database(c.build_segment(str(tmp_path / "f.seg"), stripe_bits=stripe_bits))
    n = (lambda full, quick: quick if QUICK else full)
    qs = (H.gen_phrase_queries(n(40, 6), n_docs, vocab, seed=281) + H.gen_phrase_queries(n(12, 2), n_docs, vocab, seed=282, window_extra=3) +
          H.gen_phrase_queries(n(12, 2), n_docs, vocab, seed=283, window_extra=4, op="NEAR") +
          H.gen_phrase_queries(n(10, 2), n_docs, vocab, seed=284, lengths=(4,)) +
          H.gen_phrase_queries(n(6, 1), n_docs, vocab, seed=285, lengths=(5, 6)))          # (> 4 terms: no LIST body — answered when the batch is collected)
    shapes = [(0, 10, 0), (0, 3, 0), (2, 5, 0), (0, 1, 0), (0, 64, 0), (0, 100, 0), (0, 10, 25), (0, 10, 10 ** 6)]
    plans, meta = [], []
    for qi, q in enumerate(qs):
        query = Query(q["op"], q["terms"], window=q.get("window", 0))
        for first, maxitems, cal in (shapes if not QUICK else shapes[qi % 3::3]):
            plans.append(plan(db, query, first, maxitems, check_at_least=max(cal, first + maxitems)))
            meta.append((q, first, maxitems, cal))
    # plain operators ride in the same batch: their replay bits mean nothing
    others = H.gen_term_queries("AND", n(6, 2), 3, 1, 60, seed=286) + H.gen_term_queries("OR", n(6, 2), 4, 1, 300, seed=287)
    other_plans = [plan(db, Query(q["op"], q["terms"]), 0, 10) for q in others]
    before = replay_info()
    got = search_batch_replay(db, plans + other_plans)
    after = replay_info()
    listed, declined, collected = (after[i] - before[i] for i in range(3))
    froze = differs = exact_known = 0
    for (q, first, maxitems, cal), p, (page, hdr, known) in zip(meta, plans, got):
        k = first + maxitems
        want_page, want_hdr, want_known = search_replay(db, p, REPLAY_FROZEN_WEIGHT)
        what = (q, first, maxitems, cal)
        assert page == want_page, (what, page[:3], want_page[:3])
        assert hdr.n_hits == want_hdr.n_hits and hdr.max_possible == want_hdr.max_possible, what
        if want_page:
            assert hdr.max_attained == want_hdr.max_attained and hdr.max_weight_subqs_matched == want_hdr.max_weight_subqs_matched, what
        m, m_lb = hdr.matches_exact & ~_lib.XGM_MATCHES_LOWER_BOUND, bool(hdr.matches_exact & _lib.XGM_MATCHES_LOWER_BOUND)
        assert (m <= want_hdr.matches_exact and m >= len(page)) if m_lb else m == want_hdr.matches_exact, (what, m, m_lb, want_hdr.matches_exact)
        assert ((known & ~LB) <= want_known) if known & LB else known == want_known, (what, known, want_known)
        exact_known += not (known & LB)
        if cal == 0:
            ref, _ = H.oracle_search(c, q["op"], q["terms"], first, maxitems, q.get("window", 0), reference_select_bug=True)
            assert page == ref, what          # (the oracle, like the device, hands out the first + maxitems best: the caller drops `first`)
        intended, _ = search_batch(db, [p])[0]
        froze += want_hdr.matches_exact > k
        differs += [(d, w) for d, w, _ in page] != [(h.docid, h.weight) for h in intended]
    for q, p, (page, hdr, known) in zip(others, other_plans, got[len(plans):]):
        want, _ = search_batch(db, [p])[0]
        assert [(d, w) for d, w, _ in page] == [(h.docid, h.weight) for h in want], q
        assert known == 0, q
    assert (listed == 0 if NO_LIST else listed >= (4 if QUICK else 200)) and collected >= (1 if QUICK else 30), (listed, declined, collected)
    assert froze >= (3 if QUICK else 80) and differs >= (1 if QUICK else 20), (froze, differs)
    # the same rows from the synchronous entry point, one query per call (what the matcher hook issues)
    for p, (page, hdr, _) in list(zip(plans, got))[:: (7 if not QUICK else 3)]:
        qs1 = (_lib.Query * 1)()
        C.memmove(C.byref(qs1[0]), C.byref(p), C.sizeof(_lib.Query))
        qs1[0].replay = _lib.XGM_REPLAY_BATCH_FROZEN
        k = max(1, p.first + p.maxitems)
        hits, h1 = (_lib.Hit * k)(), (_lib.ResultHdr * 1)()
        _lib.check(_lib.lib().xgm_search_batch(db._h, qs1, 1, k, hits, h1))
        assert [(hits[i].docid, hits[i].weight, hits[i].subqs_matched) for i in range(h1[0].n_hits)] == page
    # XGM_REPLAY_BATCH_COUNT on top: the same pages, every figure exact (the listing units walk their whole range; plain operators are
    # counted by xgm_search_replay when the batch is collected)
    sel = list(range(0, len(plans), 5 if not QUICK else 2))
    both = _lib.XGM_REPLAY_BATCH_FROZEN | _lib.XGM_REPLAY_BATCH_COUNT
    got2 = search_batch_replay(db, [plans[i] for i in sel] + other_plans, replay=both)
    for i, (page, hdr, known) in zip(sel, got2):
        want_page, want_hdr, want_known = search_replay(db, plans[i], REPLAY_FROZEN_WEIGHT)
        assert page == want_page and known == want_known and hdr.matches_exact == want_hdr.matches_exact, (meta[i], known, want_known, hdr.matches_exact, want_hdr.matches_exact)
    for q, p, (page, hdr, known) in zip(others, other_plans, got2[len(sel):]):
        want_page, want_hdr, want_known = search_replay(db, p)
        assert page == want_page and known == want_known and hdr.matches_exact == want_hdr.matches_exact, (q, known, want_known)
    db.close()
    c.close()


@pytest.mark.parametrize("stripe_bits", [0, 10])
def test_batch_count_of_conjunctions_equals_the_per_query_replay(built, tmp_path, stripe_bits):
    """XGM_REPLAY_BATCH_COUNT on plain conjunctions: xgm_andw_all_kernel's units list every match, xgm_count.hip scans the units' top-k lists and replays
    ProtoMSet per unit — page, header and known_matching_docs against xgm_search_replay(XGM_REPLAY_COUNT), which walks one query's whole match (itself
    checked against the host restatement pinned to the compiled reference, tests/test_gpu_all.py).  Frequent and rare terms (the dense and the flat body),
    pages of several sizes, a first > 0, FILTER, a check_at_least beyond the page (counted when the batch is collected) and a tiny arena (overflow)."""
    n_docs, vocab = (3000, 300) if QUICK else (60000, 6000)
    c = H.Corpus(n_docs, vocab)
    db = Database(c.build_segment(str(tmp_path / "c.seg"), stripe_bits=stripe_bits))
    n = (lambda full, quick: quick if QUICK else full)
    qs = (H.gen_term_queries("AND", n(30, 5), 2, 1, 30, seed=381) + H.gen_term_queries("AND", n(30, 5), 3, 1, 200, seed=382) +
          H.gen_term_queries("AND", n(20, 3), 2, 20, 3000, seed=383) + H.gen_term_queries("AND", n(10, 2), 4, 1, 40, seed=384) +
          H.gen_sided_queries("FILTER", n(10, 2), 2, 1, 1, 100, seed=385))
    shapes = [(0, 10, 0), (0, 1, 0), (3, 7, 0), (0, 64, 0), (0, 10, 500), (0, 100, 0)]
    plans, meta = [], []
    for qi, q in enumerate(qs):
        query = Query(q["op"], q["terms"], n_required=q.get("n_required", 0))
        for first, maxitems, cal in (shapes if not QUICK else shapes[qi % 2::2]):
            plans.append(plan(db, query, first, maxitems, check_at_least=max(cal, first + maxitems)))
            meta.append((q, first, maxitems, cal))
    before = replay_info()
    got = search_batch_replay(db, plans, replay=_lib.XGM_REPLAY_BATCH_COUNT)
    after = replay_info()
    on_device, declined, collected = (after[i] - before[i] for i in range(3))
    events = 0
    for what, p, (page, hdr, known) in zip(meta, plans, got):
        want_page, want_hdr, want_known = search_replay(db, p)
        assert page == want_page, (what, page[:3], want_page[:3])
        assert known == want_known and hdr.matches_exact == want_hdr.matches_exact and hdr.n_hits == want_hdr.n_hits, (what, known, want_known, hdr.matches_exact, want_hdr.matches_exact)
        if want_page:
            assert hdr.max_attained == want_hdr.max_attained and hdr.max_weight_subqs_matched == want_hdr.max_weight_subqs_matched, what
        events += known < want_hdr.matches_exact
    assert (on_device == 0 if NO_LIST else on_device >= (6 if QUICK else 300)) and collected >= (1 if QUICK else 30) and events >= (2 if QUICK else 50), (on_device, declined, collected, events)
    db.close()
    c.close()
