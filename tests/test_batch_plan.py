"""CPU-only: how a batch is cut into work units and which kernel it is dispatched to (xgm_api.cc::plan_batch),
on a host-only index.  Invariants: the units of every query tile its stripes exactly once, every unit's output
slot is unique and inside the query's slot range, the per-query unit count fits the merge kernel; dispatch: a batch
goes to the wave-autonomous kernels only when every query of it has the shape they implement."""
import ctypes as C

import pytest

import helpers as H
from xapiand_amd import Database, Query, _lib
from xapiand_amd.enquire import plan


@pytest.fixture(scope="module")
def host_db(built, tmp_path_factory):
    d = tmp_path_factory.mktemp("bp")
    c = H.Corpus(120000, 200000)                      # 15 stripes of 8192 docids
    db = Database(c.build_segment(str(d / "c.seg")), device=_lib.XGM_DEVICE_NONE)
    yield db
    db.close()


def plan_batch(db, queries, first=0, maxitems=10):
    L = _lib.lib()
    L.xgm_debug_plan_batch.restype = C.c_int64
    L.xgm_debug_plan_batch.argtypes = [C.c_void_p, C.POINTER(_lib.Query), C.c_uint32, C.c_char_p, C.POINTER(C.c_uint32), C.c_uint64]
    plans = [plan(db, q, first, maxitems) for q in queries]
    arr = (_lib.Query * len(plans))(*plans)
    name = C.create_string_buffer(32)
    cap = 1 << 16
    units = (C.c_uint32 * (4 * cap))()
    n = L.xgm_debug_plan_batch(db._h, arr, len(plans), name, units, cap)
    if n > 0 and n > cap:
        raise AssertionError("too many units")
    _lib.check(min(0, n) if n < 0 else 0)
    if n == _lib.XGM_UNSUPPORTED:
        raise _lib.XgmUnsupported()
    return name.value.decode(), [tuple(units[4 * i:4 * i + 4]) for i in range(n)]


def term_queries(op, n, nt, seed, **kw):
    return [Query(q["op"], q["terms"], n_required=q.get("n_required", 0)) for q in H.gen_term_queries(op, n, nt, 1, 4096, seed=seed, **kw)]


def sided_queries(op, n, nr, no, seed):
    return [Query(q["op"], q["terms"], n_required=q["n_required"]) for q in H.gen_sided_queries(op, n, nr, no, 1, 500, 1, 4096, seed=seed)]


def check_units(db, nq, units):
    n_stripes = (db.info().lastdocid >> 13) + 1
    per = {}
    for qi, sb, se, slot in units:
        assert qi < nq and sb < se <= n_stripes
        per.setdefault(qi, []).append((sb, se, slot))
    assert sorted(per) == list(range(nq))
    slots = sorted(s for u in per.values() for _, _, s in u)
    assert slots == list(range(len(units)))                       # dense, unique output slots
    for qi, us in per.items():
        us.sort()
        assert us[0][0] == 0 and us[-1][1] == n_stripes
        assert all(a[1] == b[0] for a, b in zip(us, us[1:]))       # tiles the stripes exactly once
        s = sorted(x[2] for x in us)
        assert s == list(range(s[0], s[0] + len(us)))              # one contiguous slot range per query
        assert len(us) * 16 <= 8192


def test_units_tile_every_query(host_db):
    for queries, k in ((term_queries("AND", 40, 3, 1), 10), (term_queries("OR", 30, 5, 2), 100), (term_queries("AND", 3, 2, 3), 10)):
        name, units = plan_batch(host_db, queries, 0, k)
        check_units(host_db, len(queries), units)


def test_kernel_dispatch(host_db):
    AND, OR = term_queries("AND", 6, 3, 11), term_queries("OR", 6, 4, 12)
    assert plan_batch(host_db, AND)[0] == "xgm_andw_kernel"
    assert plan_batch(host_db, OR, 0, 100)[0] == "xgm_orw_kernel"
    assert plan_batch(host_db, AND + OR)[0] == "xgm_match_kernel"                         # ONE launch for mixed shapes: the general kernel
    assert plan_batch(host_db, AND, 0, 500)[0] == "xgm_and_kernel"                        # k beyond the wave kernel's buffer
    assert plan_batch(host_db, term_queries("AND", 2, 1, 13))[0] == "xgm_match_kernel"    # single-term queries
    phrase = [Query("PHRASE", q["terms"]) for q in H.gen_phrase_queries(4, 120000, 200000, seed=14)]
    assert plan_batch(host_db, phrase)[0] == "xgm_andw_kernel:phrase"
    assert plan_batch(host_db, phrase + AND)[0] == "xgm_andw_kernel:phrase"               # plain conjunctions ride along
    assert plan_batch(host_db, sided_queries("AND_NOT", 4, 2, 2, 15) + AND)[0] == "xgm_andw_kernel:sided1"
    assert plan_batch(host_db, sided_queries("AND_MAYBE", 4, 2, 2, 16) + sided_queries("AND_NOT", 2, 1, 1, 17))[0] == "xgm_andw_kernel:sided2"
    assert plan_batch(host_db, sided_queries("FILTER", 4, 2, 1, 18))[0] == "xgm_andw_kernel"   # a FILTER is a conjunction
    assert plan_batch(host_db, sided_queries("AND_MAYBE", 2, 3, 6, 19))[0] == "xgm_match_kernel"  # 9 terms: beyond the register program
    assert plan_batch(host_db, sided_queries("AND_NOT", 2, 2, 1, 20) + OR)[0] == "xgm_match_kernel"


def launches(db, queries, first=0, maxitems=10):
    plans = [plan(db, q, first, maxitems) for q in queries]
    arr = (_lib.Query * len(plans))(*plans)
    out = C.create_string_buffer(512)
    n = _lib.lib().xgm_debug_batch_launches(db._h, arr, len(plans), out, 512)
    assert n >= 0
    return out.value.decode().split(";")


def test_heterogeneous_batch_is_cut_by_kernel_class(host_db):
    """A server's natural batch mixes shapes: xgm_search_batch launches one kernel per class present instead of sending
    everything to the general workgroup kernel."""
    AND, OR = term_queries("AND", 6, 3, 11), term_queries("OR", 5, 4, 12)
    phrase = [Query("PHRASE", q["terms"]) for q in H.gen_phrase_queries(4, 120000, 200000, seed=14)]
    an, am = sided_queries("AND_NOT", 3, 2, 2, 15), sided_queries("AND_MAYBE", 2, 2, 2, 16)
    single = term_queries("AND", 2, 1, 13)
    assert launches(host_db, AND) == ["xgm_andw_kernel*6"]
    mixed = [AND[0], OR[0], phrase[0], an[0], AND[1], am[0], OR[1], single[0]] + AND[2:] + OR[2:] + phrase[1:] + an[1:] + am[1:] + single[1:]
    assert launches(host_db, mixed) == ["xgm_andw_kernel*6", "xgm_andw_kernel:sided1*3", "xgm_andw_kernel:sided2*2", "xgm_andw_kernel:phrase*4",
                                        "xgm_orw_kernel*5", "xgm_match_kernel*2"]
    big = sided_queries("AND_MAYBE", 2, 3, 6, 19)                   # 9 terms: beyond the register program → the general kernel
    assert launches(host_db, AND + big) == ["xgm_andw_kernel*6", "xgm_match_kernel*2"]
    # deep pages (first + maxitems > 192) leave the wave kernels without taking the shallow ones along
    plans = [plan(host_db, q, 0, 10) for q in AND[:4]] + [plan(host_db, q, 0, 500) for q in AND[4:]]
    arr = (_lib.Query * len(plans))(*plans)
    out = C.create_string_buffer(512)
    assert _lib.lib().xgm_debug_batch_launches(host_db._h, arr, len(plans), out, 512) == 2
    assert out.value.decode().split(";") == ["xgm_andw_kernel*4", "xgm_and_kernel*2"]
