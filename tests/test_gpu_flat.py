"""Conjunctions led by a long-tail term (round 4): the flat posting arrays built when an index is opened (xgm_seg_dev::flat_*) and
xgm_flat_unit (xapiand_amd/csrc/xgm_flat_body.inc) — the lead term's postings streamed 64 per round, the other terms probed in their
containers or binary-searched in their own flat slices — against the oracle: docids, weight bit patterns, exact match counts;
AND of 2..4 terms and FILTER, pages up to 64, two stripe widths.  The request tallies prove the path is the one that ran (not a block
decoded).  Also runs under the CPU emulation (tests/test_emu.py)."""
import ctypes as C
import os

import pytest

import helpers as H
from xapiand_amd import Database, Query, _lib
from xapiand_amd.enquire import plan, search_batch

pytestmark = [pytest.mark.gpu]

QUICK = bool(os.environ.get("XGM_EMU_QUICK"))


@pytest.mark.parametrize("stripe_bits", [0, 10])
def test_flat_led_conjunctions_vs_oracle(built, tmp_path, stripe_bits):
    n_docs, vocab = (6000, 12000) if QUICK else (120000, 100000)
    c = H.Corpus(n_docs, vocab)
    db = Database(c.build_segment(str(tmp_path / "s.seg"), stripe_bits=stripe_bits))
    L = _lib.lib()
    n = (lambda full, quick: quick if QUICK else full)
    hi = vocab // 2
    qs = (H.gen_term_queries("AND", n(96, 16), 3, 1, hi // 4, seed=3) + H.gen_term_queries("AND", n(96, 16), 2, 1, hi, seed=4) +
          H.gen_term_queries("AND", n(48, 8), 2, hi // 20, hi, seed=5) + H.gen_term_queries("AND", n(48, 8), 4, 1, hi // 8, seed=6) +
          H.gen_sided_queries("FILTER", n(32, 6), 2, 1, 1, hi // 4, seed=7) + H.gen_sided_queries("FILTER", n(16, 4), 1, 2, 1, hi // 4, seed=8))
    for i, q in enumerate(qs):
        q["first"], q["maxitems"] = [(0, 10), (0, 10), (3, 7), (0, 64), (0, 1)][i % 5]
    plans = [plan(db, Query(q["op"], q["terms"], n_required=q.get("n_required", 0)), q["first"], q["maxitems"]) for q in qs]
    db.set_profiling(2)                                    # the tallying instantiation: what the batch requested from memory
    got = search_batch(db, plans)
    tl = (C.c_uint64 * 10)()
    assert L.xgm_last_batch_traffic(db._h, tl, 10) == 0
    db.set_profiling(0)
    n_hits = n_matching = 0
    for q, (hits, hdr) in zip(qs, got):
        want, oh = H.oracle_search(c, q["op"], q["terms"], q["first"], q["maxitems"], n_required=q.get("n_required", 0))
        assert [(h.docid, h.weight, h.subqs_matched) for h in hits] == want, q
        assert hdr.matches_exact == oh.matches and hdr.max_possible == oh.max_possible, q
        if want:
            assert hdr.max_attained == oh.max_attained, q
        n_hits += len(want)
        n_matching += oh.matches > 0
    assert n_hits > (30 if QUICK else 600) and n_matching > (8 if QUICK else 80), (n_hits, n_matching)
    # every query of this batch is a plain conjunction / FILTER of <= 4 terms with k <= 64: all-container queries take the dense body, the
    # others the flat body — no posting block is decoded, no block header read
    if not os.environ.get("XGM_NO_FLAT") and not os.environ.get("XGM_NO_DENSE"):
        assert tl[2] == 0 and tl[3] == 0, list(tl)
        assert tl[5] > 0, list(tl)                          # (the flat arrays' words are tallied as "other streamed words")
    # the same plans one at a time (latency mode cuts a query into many units)
    for q, p, (hits, hdr) in list(zip(qs, plans, got))[::7]:
        (h1, hdr1), = search_batch(db, [p])
        assert [(h.docid, h.weight) for h in h1] == [(h.docid, h.weight) for h in hits] and hdr1.matches_exact == hdr.matches_exact, q
    db.close()
    c.close()


def test_flat_led_positional_queries_vs_oracle(built, tmp_path):
    """PHRASE / windowed PHRASE / NEAR led by a long-tail term (xgm_flat_unit<PHRASE>): the lead term's flat postings streamed, the other terms
    from containers or their own flat slices, candidates weighed first, survivors' positions (flat_pos of the flat terms, bucket bases of the
    container terms) tested from LDS — hits bit-equal to the oracle, match counts exact or a flagged lower bound; the same with
    check_at_least asking for the exact count (then the queue path answers: every candidate's positions are tested)."""
    n_docs, vocab = (6000, 12000) if QUICK else (120000, 100000)
    c = H.Corpus(n_docs, vocab)
    db = Database(c.build_segment(str(tmp_path / "p.seg")))
    L = _lib.lib()
    n = (lambda full, quick: quick if QUICK else full)
    qs = (H.gen_phrase_queries(n(160, 24), n_docs, vocab, seed=31, lengths=(2, 3)) + H.gen_phrase_queries(n(48, 8), n_docs, vocab, seed=32, window_extra=3) +
          H.gen_phrase_queries(n(48, 8), n_docs, vocab, seed=33, window_extra=4, op="NEAR") + H.gen_phrase_queries(n(24, 6), n_docs, vocab, seed=34, lengths=(4,)))
    for i, q in enumerate(qs):
        q["first"], q["maxitems"] = [(0, 10), (0, 10), (2, 5), (0, 64), (0, 1)][i % 5]
    plans = [plan(db, Query(q["op"], q["terms"], window=q.get("window", 0)), q["first"], q["maxitems"]) for q in qs]
    db.set_profiling(2)
    got = search_batch(db, plans)
    tl = (C.c_uint64 * 10)()
    assert L.xgm_last_batch_traffic(db._h, tl, 10) == 0
    db.set_profiling(0)
    n_hits = 0
    for q, (hits, hdr) in zip(qs, got):
        want, oh = H.oracle_search(c, q["op"], q["terms"], q["first"], q["maxitems"], q.get("window", 0))
        assert [(h.docid, h.weight, h.subqs_matched) for h in hits] == want, q
        H.check_matches(hdr.matches_exact, oh.matches, len(hits), q)
        n_hits += len(want)
    assert n_hits > (40 if QUICK else 600), n_hits
    if not any(os.environ.get(v) for v in ("XGM_NO_FLAT", "XGM_NO_DENSE", "XGM_NO_FLAT_PHRASE", "XGM_NO_DENSE_PHRASE_BODY", "XGM_NO_POS_PRUNE", "XGM_NO_PHRASEW")):
        assert tl[2] == 0 and tl[3] == 0, list(tl)            # every query took one of the two bodies: no block decoded
    # exact counts asked for: the same hits, the exact number of positional matches
    for q in qs[::5]:
        p = plan(db, Query(q["op"], q["terms"], window=q.get("window", 0)), q["first"], q["maxitems"], check_at_least=H.EXACT_COUNT)
        (hits, hdr), = search_batch(db, [p])
        want, oh = H.oracle_search(c, q["op"], q["terms"], q["first"], q["maxitems"], q.get("window", 0))
        assert [(h.docid, h.weight) for h in hits] == [(d, w) for d, w, _ in want] and hdr.matches_exact == oh.matches, q
    db.close()
    c.close()
