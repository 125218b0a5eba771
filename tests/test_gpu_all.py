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
