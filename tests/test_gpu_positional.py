"""The positional filter (K6) in all its forms against the CPU oracle (pinned to the compiled reference by
tests/test_oracle_vs_reference.py): exact phrases of 2-8 terms, windowed phrases, NEAR (any order, span < window),
documents with more positions of a term than the LDS fast path holds, and terms whose positions need 4 bytes.
Reference: src/xapian/matcher/exactphrasepostlist.cc:75-133, phrasepostlist.cc:60-90, nearpostlist.cc:60-160."""
import random

import pytest

import helpers as H
from xapiand_amd import Database, Query, plan, search_batch

pytestmark = pytest.mark.gpu

N_DOCS, VOCAB = 60000, 3000          # a small vocabulary: long n-grams recur, windows have real work to do


def check(db, corpus, qs):
    """Both ways a positional query is answered: check_at_least inside the page (the default: candidates that cannot rank are
    dropped before their positions are tested, the count is a flagged lower bound) and asking for the exact count."""
    n_hits = 0
    for cal in (0, H.EXACT_COUNT):
        plans = [plan(db, Query(q["op"], q["terms"], window=q.get("window", 0)), q["first"], q["maxitems"], cal) for q in qs]
        got = search_batch(db, plans)
        for q, (hits, hdr) in zip(qs, got):
            want, oh = H.oracle_search(corpus, q["op"], q["terms"], q["first"], q["maxitems"], q.get("window", 0))
            assert [(h.docid, h.weight) for h in hits] == [(d, w) for d, w, _ in want], (q, cal)
            if cal:
                assert hdr.matches_exact == oh.matches, q
            else:
                H.check_matches(hdr.matches_exact, oh.matches, len(hits), q)
                n_hits += len(want)
    return n_hits


def test_phrase_near_window_and_long_phrases(built, tmp_path):
    c = H.Corpus(N_DOCS, VOCAB)
    db = Database(c.build_segment(str(tmp_path / "p.seg")))
    qs = (H.gen_phrase_queries(40, N_DOCS, VOCAB, seed=1, lengths=(2, 3)) + H.gen_phrase_queries(40, N_DOCS, VOCAB, seed=2, lengths=(4, 5, 6, 8)) +
          H.gen_phrase_queries(30, N_DOCS, VOCAB, seed=3, window_extra=3, lengths=(2, 3, 4)) +
          H.gen_phrase_queries(40, N_DOCS, VOCAB, seed=4, window_extra=2, lengths=(2, 3, 5), op="NEAR") +
          H.gen_phrase_queries(20, N_DOCS, VOCAB, seed=5, window_extra=12, lengths=(2, 3), op="NEAR"))
    # NEAR is order-free: shuffle the terms of half of them
    rng = random.Random(9)
    for q in qs:
        if q["op"] == "NEAR" and rng.random() < 0.5:
            rng.shuffle(q["terms"])
    for q in qs:
        q["maxitems"] = 150 if rng.random() < 0.3 else 10
    assert check(db, c, qs) > 500
    # and in one heterogeneous batch with plain conjunctions (kernel classes are cut apart, results are not)
    mixed = qs[:20] + [dict(q, window=0) for q in H.gen_term_queries("AND", 10, 3, 1, 300, seed=6)]
    check(db, c, mixed)
    db.close()


def test_slow_path_many_positions_and_wide_positions(built, tmp_path):
    """wdf > 16 (more than the fast path stages per document and term), wdf > 64 (more than its wide second pass does) and positions >= 65 536
    (4-byte lists)."""
    rng = random.Random(3)
    post = {"a": [], "b": [], "c": [], "w": [], "x": []}
    doclen = {}
    for d in range(1, 4001):
        na, nb = rng.choice([1, 3, 9, 17, 20, 40, 64, 70, 130]), rng.choice([1, 2, 8, 16, 18, 33, 65])      # <= 16: staged 64 documents at a time; <= 64: 16 at a time; more: one at a time
        pa = sorted(rng.sample(range(1, 400), na))
        pb = sorted(set(min(399, p + rng.choice([1, 1, 2, 5])) for p in rng.sample(pa, min(len(pa), nb))) | set(rng.sample(range(1, 400), max(0, nb - na))))
        pc = sorted(rng.sample(range(1, 400), rng.choice([1, 2, 5])))
        post["a"].append((d, len(pa), pa)); post["b"].append((d, len(pb), pb)); post["c"].append((d, len(pc), pc))
        # wide positions: a long document
        pw = sorted(rng.sample(range(60000, 70000), rng.choice([1, 3, 6])))
        px = sorted(set(p + 1 for p in pw[: rng.choice([0, 1, 2])]) | set(rng.sample(range(60000, 70000), 2)))
        post["w"].append((d, len(pw), pw)); post["x"].append((d, len(px), px))
        doclen[d] = 70000
    c = H.ManualCorpus(post, doclen)
    db = Database(c.build_segment(str(tmp_path / "s.seg")))
    # b's positions are drawn next to a's and can fall ON one of them: several terms per position — NEAR by NearPostList's full procedure
    # (xgm_index_set_near_colocated; the oracle always runs it).  With > 16 positions per list that is the serial path from the LDS copy.
    db.set_near_colocated(True)
    qs = []
    for terms in (["a", "b"], ["b", "a"], ["a", "b", "c"], ["w", "x"], ["a", "w"], ["x", "w", "a"]):
        for op, win in (("PHRASE", 0), ("PHRASE", len(terms) + 3), ("NEAR", len(terms) + 2), ("NEAR", 40)):
            qs.append(dict(op=op, terms=terms, first=0, maxitems=150, window=win))
            qs.append(dict(op=op, terms=terms, first=0, maxitems=10, window=win))
    assert check(db, c, qs) > 50
    db.close()


def test_near_over_colocated_terms(built, tmp_path):
    """Documents with several terms per position (helpers.coloc_postings): with xgm_index_set_near_colocated the device runs NearPostList's
    procedure IN FULL (duplicate-position step, nearpostlist.cc:106-140; xgm_posfilter.h near_colocated) — every kernel class that can carry
    a NEAR (the bodies for all-container / long-tail-led queries, the queue path, the workgroup kernel via the variant switches) against the
    oracle's stateless reading, the whole match and top-10 pages; without the flag the wave-parallel predicate accepts coinciding heads,
    which is what it must NOT be used for (the answers differ: checked, so that the test would notice a flag that does nothing)."""
    post, doclen = H.coloc_postings()
    c = H.ManualCorpus(post, doclen)
    db = Database(c.build_segment(str(tmp_path / "c.seg")))
    qs = H.coloc_near_queries()
    plain = [search_batch(db, [plan(db, Query(q["op"], q["terms"], window=q["window"]), 0, q["maxitems"], H.EXACT_COUNT)])[0] for q in qs[:6]]
    db.set_near_colocated(True)
    assert check(db, c, qs) > 1000
    flagged = [search_batch(db, [plan(db, Query(q["op"], q["terms"], window=q["window"]), 0, q["maxitems"], H.EXACT_COUNT)])[0] for q in qs[:6]]
    assert any(a[1].matches_exact != b[1].matches_exact for a, b in zip(plain, flagged))
    db.close()
