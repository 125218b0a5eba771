"""Nested operator trees on the device: boolean combinations, _and_not / _and_maybe / _filter around
them, OP_SYNONYM groups, OP_SCALE_WEIGHT, wqf > 1 — planned like the reference lowers them (xgm_plan.cc::plan_tree;
src/xapian/api/queryinternal.cc) and evaluated by the general match kernel — against the CPU oracle, which is pinned
bit-exactly to the compiled reference (tests/test_oracle_vs_reference.py, tests/golden/nested_trees.json)."""
import random

import pytest

import helpers as H
from xapiand_amd import Database, Query, merged_stats, plan, search_batch

pytestmark = pytest.mark.gpu

N_DOCS, VOCAB = 150000, 100000


def test_random_trees_match_oracle(built, tmp_path):
    c = H.Corpus(N_DOCS, VOCAB)
    db = Database(c.build_segment(str(tmp_path / "t.seg")))
    qs = H.gen_tree_queries(96, 1, 2000, seed=5)
    rng = random.Random(2)
    for q in qs:
        q["maxitems"] = rng.choice([10, 10, 50, 300])
        q["first"] = rng.choice([0, 0, 4])
    plans = [plan(db, Query.tree(q["tree"]), q["first"], q["maxitems"]) for q in qs]
    # one heterogeneous batch: the trees ride with flat conjunctions and disjunctions
    flat = H.gen_term_queries("AND", 8, 3, 1, 300, seed=7) + H.gen_term_queries("OR", 8, 4, 1, 3000, seed=8)
    fplans = [plan(db, Query(q["op"], q["terms"]), 0, 10) for q in flat]
    got = search_batch(db, plans + fplans)
    n_hits = 0
    for q, p, (hits, hdr) in zip(qs, plans, got):
        rows, oh, tot = H.oracle_search_tree(c, q["tree"], q["first"], q["maxitems"])
        assert [(h.docid, h.weight, h.subqs_matched) for h in hits] == rows, q["tree"]
        assert hdr.matches_exact == oh.matches and hdr.max_possible == oh.max_possible and p.total_subqs == tot, q["tree"]
        if rows:
            assert hdr.max_attained == oh.max_attained and hdr.max_weight_subqs_matched == oh.max_subqs, q["tree"]
        n_hits += len(rows)
    for q, (hits, hdr) in zip(flat, got[len(qs):]):
        want, _ = H.oracle_search(c, q["op"], q["terms"], 0, 10)
        assert [(h.docid, h.weight) for h in hits] == [(d, w) for d, w, _ in want], q
    assert n_hits > 1500
    db.close()


def test_trees_with_merged_statistics_over_shards(built, tmp_path):
    """Xapiand's per-shard protocol: every shard plans with the MERGED statistics (synonym weights included) but orders
    MultiAnd children and OR trees by its own estimates."""
    shards = [H.Corpus(40000, 50000, n_shards=2, shard=s) for s in range(2)]
    dbs = [Database(c.build_segment(str(tmp_path / ("s%d.seg" % s)))) for s, c in enumerate(shards)]
    qs = H.gen_tree_queries(36, 1, 600, seed=9)
    for q in qs:
        query = Query.tree(q["tree"])
        gs = merged_stats(dbs, query)
        terms = [t.decode() for t in query.terms]
        g = dict(total_length=gs.total_length, collection_size=gs.collection_size, termfreq={t: gs.termfreq[i] for i, t in enumerate(terms)})
        for s, (db, c) in enumerate(zip(dbs, shards)):
            (hits, hdr), = search_batch(db, [plan(db, query, 0, 20, global_stats=gs)])
            rows, oh, _ = H.oracle_search_tree(c, q["tree"], 0, 20, global_stats=g)
            assert [(h.docid, h.weight) for h in hits] == [(d, w) for d, w, _ in rows], (s, q["tree"])
            assert hdr.max_possible == oh.max_possible, (s, q["tree"])
    for d in dbs:
        d.close()
