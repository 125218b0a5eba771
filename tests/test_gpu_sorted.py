"""Searches under a value sort on the device (first version) against the oracle, which is pinned to the compiled
reference for the three sorts in both directions (tests/test_oracle_vs_reference.py, tests/golden/sorted_values.json).

First run on an MI355X at the start of round 3 (profiles/r03_sorted_first_gpu.log: 8 passed); part of the -m gpu suite since.
The same file also runs under the CPU emulation of the kernels (tests/test_emu.py against tests/emu/libxgm_emu.so)."""
import os
import random

import pytest

import helpers as H
from xapiand_amd import Database, Enquire, Query, ValueCountMatchSpy, _lib
from xapiand_amd.enquire import merged_stats, plan, read_column_values, search_collapsed, search_sorted, search_sorted_batch, search_sorted_spy

pytestmark = [pytest.mark.gpu]

QUICK = bool(os.environ.get("XGM_EMU_QUICK"))          # under emulation a workgroup barrier costs 256 fiber switches: small sizes

MODES = {"V": _lib.XGM_SORT_VALUE, "VR": _lib.XGM_SORT_VALUE_RELEVANCE, "RV": _lib.XGM_SORT_RELEVANCE_VALUE}


def write_column(corpus, slot, path):
    import ctypes as C
    H.oracle_search_sorted(corpus, "OR", ["t1"], 0, 1, "V", slot, False)            # (makes the oracle index and its value slots)
    ol = H.olib()
    ol.xgo_write_value_column.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p]
    assert ol.xgo_write_value_column(corpus.oracle_index(), slot, path.encode()) == 0
    return path


@pytest.mark.parametrize("stripe_bits", [0, 10])
def test_value_sorts_vs_oracle(built, tmp_path, stripe_bits):
    c = H.Corpus(*((3000, 8000) if QUICK else (30000, 60000)))
    db = Database(c.build_segment(str(tmp_path / "s.seg"), stripe_bits=stripe_bits))
    values = {}
    for slot in range(3):
        p = write_column(c, slot, str(tmp_path / ("col%d" % slot)))
        db.attach_column(p)
        values[slot] = read_column_values(p)
    rng = random.Random(3)
    n = (lambda full, quick: quick if QUICK else full)
    base = (H.gen_term_queries("OR", n(12, 3), 3, 1, 400, maxitems=10, seed=51) + H.gen_term_queries("AND", n(12, 3), 2, 1, 60, maxitems=10, seed=52) +
            H.gen_sided_queries("AND_MAYBE", n(6, 2), 1, 2, 1, 200, maxitems=10, seed=53) + H.gen_sided_queries("AND_NOT", n(6, 2), 1, 2, 1, 200, maxitems=10, seed=54) +
            H.gen_term_queries("OR", n(6, 1), 5, 1, 3000, first=7, maxitems=93, seed=55) + H.gen_term_queries("AND", n(4, 1), 3, 1, 30, maxitems=300, seed=56))
    n_items = 0
    for q in base:
        for _ in range(n(2, 1)):
            mode, slot, rev = rng.choice(["V", "VR", "RV"]), rng.randrange(3), rng.random() < 0.5
            want, whdr = H.oracle_search_sorted(c, q["op"], q["terms"], q["first"], q["maxitems"], mode, slot, rev, n_required=q.get("n_required", 0))
            p = plan(db, Query(q["op"], q["terms"], n_required=q.get("n_required", 0)), q["first"], q["maxitems"])
            got, hdr = search_sorted(db, p, MODES[mode], slot, rev)
            assert [(d, w, m) for d, w, m, _ in got] == [(d, w, m) for d, w, m, _ in want], (q, mode, slot, rev)
            assert [values[slot][o - 1] if o else b"" for _, _, _, o in got] == [k for _, _, _, k in want], (q, mode, slot, rev)
            assert hdr.matches_exact == whdr.matches and hdr.max_attained == whdr.max_attained, (q, mode, slot, rev)
            n_items += len(got)
    assert n_items > (60 if QUICK else 500)
    db.close()
    c.close()


def test_sorted_batch_equals_single_searches_and_the_oracle(built, tmp_path):
    """xgm_search_sorted_batch (round 5): >= 64 searches of mixed shapes and page sizes under ONE sort in one launch give what
    xgm_search_sorted gives for each (hits, ordinals, match count, best weight) — and a sample of them what the pinned oracle gives."""
    c = H.Corpus(*((3000, 8000) if QUICK else (30000, 60000)))
    db = Database(c.build_segment(str(tmp_path / "b.seg")))
    values = {}
    for slot in range(3):
        p = write_column(c, slot, str(tmp_path / ("col%d" % slot)))
        db.attach_column(p)
        values[slot] = read_column_values(p)
    n = (lambda full, quick: quick if QUICK else full)
    base = (H.gen_term_queries("OR", n(28, 6), 3, 1, 400, maxitems=10, seed=151) + H.gen_term_queries("AND", n(24, 5), 2, 1, 60, maxitems=10, seed=152) +
            H.gen_sided_queries("AND_MAYBE", n(8, 2), 1, 2, 1, 200, maxitems=10, seed=153) + H.gen_sided_queries("AND_NOT", n(8, 2), 1, 2, 1, 200, maxitems=10, seed=154) +
            H.gen_term_queries("OR", n(6, 1), 5, 1, 3000, first=7, maxitems=33, seed=155) + H.gen_term_queries("AND", n(4, 1), 1, 1, 30, maxitems=20, seed=156))
    assert len(base) >= (16 if QUICK else 64)
    plans = [plan(db, Query(q["op"], q["terms"], n_required=q.get("n_required", 0)), q["first"], q["maxitems"]) for q in base]
    checked = 0
    for mode, slot, rev in (("V", 0, False), ("VR", 2, True), ("RV", 1, False)):
        res = search_sorted_batch(db, plans, MODES[mode], slot, rev)
        assert len(res) == len(base)
        for qi, (q, p, (got, hdr)) in enumerate(zip(base, plans, res)):
            one, ohdr = search_sorted(db, p, MODES[mode], slot, rev)
            assert got == one, (q, mode)
            assert (hdr.n_hits, hdr.matches_exact, hdr.max_attained, hdr.max_weight_subqs_matched) == (ohdr.n_hits, ohdr.matches_exact, ohdr.max_attained, ohdr.max_weight_subqs_matched), q
            if qi % 5 == 0:
                want, whdr = H.oracle_search_sorted(c, q["op"], q["terms"], q["first"], q["maxitems"], mode, slot, rev, n_required=q.get("n_required", 0))
                assert [(d, w, m) for d, w, m, _ in got] == [(d, w, m) for d, w, m, _ in want], (q, mode, slot, rev)
                assert [values[slot][o - 1] if o else b"" for _, _, _, o in got] == [k for _, _, _, k in want], (q, mode, slot, rev)
                assert hdr.matches_exact == whdr.matches
                checked += 1
    assert checked >= (9 if QUICK else 36)
    # ... and with a ValueCountMatchSpy riding on every search of the batch: the page of the plain batch, the counts of the single spied search
    from xapiand_amd.enquire import search_sorted_spy, search_sorted_spy_batch
    nd = len(values[1])
    plain = search_sorted_batch(db, plans, MODES["V"], 0, False)
    spied = search_sorted_spy_batch(db, plans, MODES["V"], 0, False, 1, nd)
    totals = 0
    for qi, (p, (got, hdr), (sgot, shdr, counts)) in enumerate(zip(plans, plain, spied)):
        assert sgot == got and shdr.matches_exact == hdr.matches_exact, base[qi]
        assert sum(counts) == hdr.matches_exact, base[qi]
        if qi % 3 == 0:
            _, _, one_counts = search_sorted_spy(db, p, MODES["V"], 0, False, 1, nd)
            assert counts == one_counts, base[qi]
        totals += sum(counts)
    assert totals > 0
    # ... and under a collapse key (by relevance, and under a value sort): what xgm_search_collapsed gives for each
    from xapiand_amd.enquire import search_collapsed, search_collapsed_batch
    collapsed_somewhere = 0
    for sort_by, slot, cmax in ((None, 0, 1), (MODES["V"], 0, 2)):
        res = search_collapsed_batch(db, plans, 2, cmax, sort_by, slot, False)
        assert len(res) == len(plans)
        for qi, (p, (got, hdr, lb)) in enumerate(zip(plans, res)):
            one, ohdr, olb = search_collapsed(db, p, 2, cmax, sort_by, slot, False)
            assert got == one and lb == olb, (base[qi], sort_by, cmax)
            assert (hdr.n_hits, hdr.matches_exact, hdr.max_attained) == (ohdr.n_hits, ohdr.matches_exact, ohdr.max_attained), base[qi]
            collapsed_somewhere += sum(1 for r in got if r[5])
    assert collapsed_somewhere > 0
    db.close()
    c.close()


def test_value_count_spy_vs_oracle(built, tmp_path):
    """A ValueCountMatchSpy in the same pass as a value-led search: counts per distinct value over every matching document and the
    total, against the oracle's restatement (pinned to Xapian::ValueCountMatchSpy, tests/test_oracle_vs_reference.py); the page of
    hits is the plain sorted search's."""
    c = H.Corpus(*((3000, 8000) if QUICK else (30000, 60000)))
    db = Database(c.build_segment(str(tmp_path / "s.seg")))
    values = {}
    for slot in range(3):
        p = write_column(c, slot, str(tmp_path / ("col%d" % slot)))
        db.attach_column(p)
        values[slot] = read_column_values(p)
    rng = random.Random(8)
    nq = 3 if QUICK else 10
    qs = (H.gen_term_queries("OR", nq, 3, 1, 400, maxitems=10, seed=61) + H.gen_term_queries("AND", nq, 2, 1, 60, maxitems=10, seed=62) +
          H.gen_sided_queries("AND_NOT", nq // 2 + 1, 1, 2, 1, 200, maxitems=10, seed=63))
    seen = 0
    for q in qs:
        mode, slot, rev, spy_slot = rng.choice(["V", "VR"]), rng.randrange(3), rng.random() < 0.5, rng.randrange(3)
        total, want = H.oracle_spy(c, q["op"], q["terms"], spy_slot, n_required=q.get("n_required", 0))
        p = plan(db, Query(q["op"], q["terms"], n_required=q.get("n_required", 0)), q["first"], q["maxitems"])
        plain, _ = search_sorted(db, p, MODES[mode], slot, rev)
        got, hdr, counts = search_sorted_spy(db, p, MODES[mode], slot, rev, spy_slot, len(values[spy_slot]))
        assert got == plain, q
        assert hdr.matches_exact == total == sum(counts), (q, spy_slot)
        assert [(values[spy_slot][o - 1], n) for o, n in enumerate(counts) if o and n] == want, (q, spy_slot)
        seen += total
    assert seen > (300 if QUICK else 20000)
    with pytest.raises(_lib.XgmError):                            # the counters must be the column's
        search_sorted_spy(db, p, MODES["V"], 0, False, 2, len(values[2]) + 3)
    db.close()
    c.close()


def test_positional_and_nested_queries_under_a_sort(built, tmp_path):
    """The sorted kernel is the workgroup kernel's own text compiled a second time: PHRASE (exact) and nested trees rank under a
    value sort like everything else.  PHRASE against the oracle's sorted search; a tree against its relevance ranking re-ranked
    under the comparison on the host (the oracle's tree evaluator has no sorted form: its full result is small)."""
    n_docs, vocab = (3000, 8000) if QUICK else (20000, 40000)
    c = H.Corpus(n_docs, vocab)
    db = Database(c.build_segment(str(tmp_path / "s.seg")))
    values = {}
    for slot in range(3):
        p = write_column(c, slot, str(tmp_path / ("col%d" % slot)))
        db.attach_column(p)
        values[slot] = read_column_values(p)
    rng = random.Random(21)
    n_items = 0
    for q in H.gen_phrase_queries(4 if QUICK else 16, n_docs, vocab, seed=81, lengths=(2,)):
        mode, slot, rev = rng.choice(["V", "VR", "RV"]), rng.randrange(3), rng.random() < 0.5
        want, whdr = H.oracle_search_sorted(c, "PHRASE", q["terms"], 0, 10, mode, slot, rev)
        got, hdr = search_sorted(db, plan(db, Query("PHRASE", q["terms"]), 0, 10), MODES[mode], slot, rev)
        assert [(d, w, m) for d, w, m, _ in got] == [(d, w, m) for d, w, m, _ in want], (q, mode, slot, rev)
        assert hdr.matches_exact == whdr.matches, q
        n_items += len(got)
    assert n_items > 0
    import numpy as np
    paths = {slot: str(tmp_path / ("col%d" % slot)) for slot in range(3)}
    for tq in H.gen_tree_queries(3 if QUICK else 12, 1, 200, seed=82):
        tree = tq["tree"]
        full, _, _ = H.oracle_search_tree(c, tree, 0, c.v.lastdocid)
        slot, rev = rng.randrange(3), rng.random() < 0.5
        ords = np.frombuffer(open(paths[slot], "rb").read(), dtype=np.uint32, count=c.v.lastdocid + 1, offset=24)
        ranked = sorted(full, key=lambda r: (-int(ords[r[0]]) if rev else int(ords[r[0]]), r[0]))[:10]
        got, hdr = search_sorted(db, plan(db, Query.tree(tree), 0, 10), MODES["V"], slot, rev)
        assert [(d, w) for d, w, _, _ in got] == [(d, w) for d, w, _ in ranked], (tq, slot, rev)
        assert hdr.matches_exact == len(full), tq
    db.close()
    c.close()


def test_value_sorts_over_three_shards(built, tmp_path):
    """Xapiand's per-shard protocol under a value sort: every shard searched on the device with the merged statistics (its own
    column: ordinals are per shard), the shards' pages merged on the host by the value STRINGS under the same comparison — what
    Enquire::merge_mset does with the MSets a hook hands back.  Against the oracle's sharded sorted search, which is pinned to the
    reference's multi-shard Enquire (tests/test_oracle_vs_reference.py)."""
    import functools
    n_docs, vocab, n_shards = (3000, 8000, 3) if QUICK else (18000, 40000, 3)
    shards = [H.Corpus(n_docs, vocab, n_shards=n_shards, shard=s) for s in range(n_shards)]
    dbs, values = [], []
    for s, c in enumerate(shards):
        db = Database(c.build_segment(str(tmp_path / ("s%d.seg" % s))))
        vals = {}
        for slot in range(3):
            p = write_column(c, slot, str(tmp_path / ("s%d_col%d" % (s, slot))))
            db.attach_column(p)
            vals[slot] = read_column_values(p)
        dbs.append(db); values.append(vals)
    rng = random.Random(31)
    nq = 3 if QUICK else 10
    n_items = 0
    for q in H.gen_term_queries("OR", nq, 3, 1, 300, maxitems=10, seed=91) + H.gen_term_queries("AND", nq, 2, 1, 40, maxitems=10, seed=92) + H.gen_term_queries("OR", nq // 2 + 1, 2, 1, 1500, first=5, maxitems=12, seed=93):
        mode, slot, rev = rng.choice(["V", "VR", "RV"]), rng.randrange(3), rng.random() < 0.5
        want = H.oracle_search_sharded_sorted(shards, q["op"], q["terms"], q["first"], q["maxitems"], mode, slot, rev)
        query = Query(q["op"], q["terms"])
        gs = merged_stats(dbs, query)
        rows = []
        for s, db in enumerate(dbs):
            got, _ = search_sorted(db, plan(db, query, 0, q["first"] + q["maxitems"], global_stats=gs), MODES[mode], slot, rev)
            rows += [((d - 1) * n_shards + s + 1, w, m, values[s][slot][o - 1] if o else b"") for d, w, m, o in got]

        def cmp(a, b):
            if mode == "RV" and a[1] != b[1]:
                return -1 if a[1] > b[1] else 1
            if a[3] != b[3]:
                return (-1 if a[3] > b[3] else 1) if rev else (-1 if a[3] < b[3] else 1)
            if mode == "VR" and a[1] != b[1]:
                return -1 if a[1] > b[1] else 1
            return -1 if a[0] < b[0] else (1 if a[0] > b[0] else 0)
        rows.sort(key=functools.cmp_to_key(cmp))
        assert rows[q["first"]:q["first"] + q["maxitems"]] == want, (q, mode, slot, rev)
        n_items += len(want)
    assert n_items > (40 if QUICK else 200)
    for db in dbs:
        db.close()
    for c in shards:
        c.close()


def test_collapse_vs_oracle(built, tmp_path):
    """Enquire::set_collapse_key by relevance and under the value sorts, collapse_max 1..3, against the oracle's restatement of the
    INTENDED semantics (the best documents of a key stay; the reference snapshot's collapser is a quirk): docids,
    weight bits, sort and collapse keys, collapse counts, the collapsed lower bound, the uncollapsed match count."""
    c = H.Corpus(*((3000, 8000) if QUICK else (30000, 60000)))
    db = Database(c.build_segment(str(tmp_path / "s.seg"), stripe_bits=10))
    values = {}
    for slot in range(3):
        p = write_column(c, slot, str(tmp_path / ("col%d" % slot)))
        db.attach_column(p)
        values[slot] = read_column_values(p)
    rng = random.Random(12)
    nq = 3 if QUICK else 10
    qs = (H.gen_term_queries("OR", nq, 3, 1, 400, maxitems=10, seed=71) + H.gen_term_queries("AND", nq, 2, 1, 60, maxitems=10, seed=72) +
          H.gen_sided_queries("AND_MAYBE", nq // 2 + 1, 1, 2, 1, 200, maxitems=10, seed=73) + H.gen_term_queries("OR", nq // 2 + 1, 4, 1, 2000, first=5, maxitems=40, seed=74))
    n_items = n_collapsed = 0
    for q in qs:
        for _ in range(1 if QUICK else 2):
            mode = rng.choice([None, None, "V", "VR", "RV"])
            slot, rev = rng.randrange(3), rng.random() < 0.5
            cslot, cmax = rng.choice([0, 2]), rng.randrange(1, 4)
            want, whdr = H.oracle_search_sorted(c, q["op"], q["terms"], q["first"], q["maxitems"], mode, slot, rev, n_required=q.get("n_required", 0), collapse=(cslot, cmax))
            p = plan(db, Query(q["op"], q["terms"], n_required=q.get("n_required", 0)), q["first"], q["maxitems"])
            got, hdr, clb = search_collapsed(db, p, cslot, cmax, MODES[mode] if mode else None, slot, rev)
            key = lambda o: values[cslot][o - 1] if o else b""
            assert [(d, w, m) for d, w, m, _, _, _ in got] == [(d, w, m) for d, w, m, _, _, _ in want], (q, mode, slot, rev, cslot, cmax)
            assert [(key(co), cc) for _, _, _, _, co, cc in got] == [(ck, cc) for _, _, _, _, ck, cc in want], (q, mode, cslot, cmax)
            if mode:
                assert [values[slot][o - 1] if o else b"" for _, _, _, o, _, _ in got] == [k for _, _, _, k, _, _ in want], (q, mode, slot, rev)
            assert hdr.matches_exact == whdr.matches and clb == whdr.collapsed_lower_bound, (q, cslot, cmax)
            n_items += len(got)
            n_collapsed += sum(1 for g in got if g[5])
    assert n_items > (60 if QUICK else 400) and n_collapsed > 10
    db.close()
    c.close()


def test_enquire_mirror_against_the_references_own_msets(built, tmp_path):
    """Through the Enquire mirror (set_sort_by_value* / add_matchspy, as a test of the reference would read) against the fixtures
    the compiled reference generated (tests/golden/sorted_values.json, spy_counts.json): docid, weight bits, percentage and sort
    key at every rank, max_attained; the spy's total and (value, count) list."""
    import json
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    fx = json.load(open(os.path.join(gdir, "sorted_values.json")))
    cp = fx["corpus"]
    c = H.Corpus(cp["n_docs"], cp["vocab"], seed=cp["seed"], len_lo=cp["len_lo"], len_hi=cp["len_hi"])
    db = Database(c.build_segment(str(tmp_path / "g.seg")))
    for slot in range(3):
        db.attach_column(write_column(c, slot, str(tmp_path / ("col%d" % slot))))
    setters = {"V": Enquire.set_sort_by_value, "VR": Enquire.set_sort_by_value_then_relevance, "RV": Enquire.set_sort_by_relevance_then_value}
    results = fx["results"][::3] if QUICK else fx["results"]
    n_items = 0
    for r in results:
        q = r["query"]
        mode, slot, rev = q["sort"]
        enq = Enquire(db)
        enq.set_query(Query(q["op"], q["terms"], n_required=q.get("n_required", 0)))
        setters[mode](enq, slot, rev)
        mset = enq.get_mset(q["first"], q["maxitems"])
        assert [(i.docid, i.weight.hex(), i.percent) for i in mset] == [(d, w, pct) for d, w, pct in r["hits"]], q
        assert [i.sort_key.hex() for i in mset] == r["sort_keys"], q
        if r["hits"]:
            assert mset.get_max_attained() == float.fromhex(r["max_attained"]), q
        n_items += len(r["hits"])
    assert n_items > (100 if QUICK else 400)
    fx = json.load(open(os.path.join(gdir, "spy_counts.json")))
    for r in (fx["results"][::4] if QUICK else fx["results"]):
        q = r["query"]
        enq = Enquire(db)
        enq.set_query(Query(q["op"], q["terms"], n_required=q.get("n_required", 0)))
        enq.set_sort_by_value(q["spy"], False)                 # the value leads: the spy sees every match whatever check_at_least is
        spy = ValueCountMatchSpy(q["spy"])
        enq.add_matchspy(spy)
        enq.get_mset(q["first"], q["maxitems"])
        assert spy.get_total() == r["spy_total"] and [[v.hex(), n] for v, n in spy.values()] == r["spy"], q
    db.close()
    c.close()


def test_sorted_search_declines_what_it_does_not_do(built, tmp_path):
    c = H.Corpus(5000, 20000)
    db = Database(c.build_segment(str(tmp_path / "s.seg")))
    p = plan(db, Query("OR", ["t3", "t9"]), 0, 10)
    with pytest.raises(_lib.XgmUnsupported):                       # no column attached for the slot
        search_sorted(db, p, _lib.XGM_SORT_VALUE, 0)
    db.attach_column(write_column(c, 0, str(tmp_path / "col0")))
    search_sorted(db, p, _lib.XGM_SORT_VALUE, 0)
    db.close()
    c.close()
