"""CPU-only: the CPU oracle reproduces the committed golden vectors that the real reference
generated (tests/golden/make_golden.py) — docid at every rank identical, weights bit-identical."""
import glob
import json
import os

import pytest

import helpers as H

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.json")))

_cache = {}


def shards_for(fx):
    if "coloc" in fx:                      # hand-made documents with several terms per position (NEAR's duplicate-position step)
        key = ("coloc", fx["coloc"]["seed"], fx["coloc"]["n_docs"])
        if key not in _cache:
            _cache[key] = [H.ManualCorpus(*H.coloc_postings(fx["coloc"]["seed"], fx["coloc"]["n_docs"]))]
        return _cache[key]
    c = fx["corpus"]
    key = (c["seed"], c["n_docs"], c["vocab"], fx["n_shards"])
    if key not in _cache:
        _cache[key] = [H.Corpus(c["n_docs"], c["vocab"], seed=c["seed"], len_lo=c["len_lo"], len_hi=c["len_hi"],
                                n_shards=fx["n_shards"], shard=s) for s in range(fx["n_shards"])]
    return _cache[key]


def expected(r):
    return [(d, float.fromhex(w)) for d, w, _ in r["hits"]]


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_matches_golden(path):
    fx = json.load(open(path))
    shards = shards_for(fx)
    assert fx["results"]
    for r in fx["results"]:
        q = r["query"]
        if q["op"] == "RPN":
            rows, hdr, _ = H.oracle_search_tree(shards[0], H.tree_from_json(q["tree"]), q["first"], q["maxitems"])
            got = [(d, w) for d, w, _ in rows[q["first"]:]]
            assert hdr.max_possible == float.fromhex(r["max_possible"])
        elif q.get("spy") is not None:
            total, counts = H.oracle_spy(shards[0], q["op"], q["terms"], q["spy"], n_required=q.get("n_required", 0))
            assert (total, [[v.hex(), n] for v, n in counts]) == (r["spy_total"], r["spy"]), q
            continue
        elif q.get("sort"):
            mode, slot, rev = q["sort"]
            hits, hdr = H.oracle_search_sorted(shards[0], q["op"], q["terms"], q["first"], q["maxitems"], mode, slot, rev, n_required=q.get("n_required", 0))
            got = [(d, w) for d, w, _, _ in hits[q["first"]:]]
            assert [k.hex() for _, _, _, k in hits[q["first"]:]] == r["sort_keys"], q
            assert hdr.max_attained == float.fromhex(r["max_attained"]), q
        elif fx["n_shards"] == 1:
            # (co-located NEAR: the reference's answers carry its history dependence and its stale weights — the oracle's reference mode)
            hits, hdr = H.oracle_search(shards[0], q["op"], q["terms"], q["first"], q["maxitems"], q.get("window", 0), n_required=q.get("n_required", 0),
                                        reference_select_bug="coloc" in fx)
            got = [(d, w) for d, w, _ in hits[q["first"]:]]
            assert hdr.max_possible == float.fromhex(r["max_possible"])
        else:
            got = [(d, w) for d, w, _ in H.oracle_search_sharded(shards, q["op"], q["terms"], q["first"], q["maxitems"])]
        assert got == expected(r), q
