"""CPU-only: the native glass reader (xapiand_amd/csrc/xgm_glass.cc — postlist / position B-trees read
straight from the reference's on-disk format) must produce, byte for byte, what the export through the
REAL reference's public iterators produces (oracle/_ref/xapian_ref export), and the same device segment."""
import os

import pytest

import helpers as H
from xapiand_amd import _lib

pytestmark = pytest.mark.skipif(not H.have_xapian_ref(), reason="oracle/_ref/xapian_ref not built")


def both_exports(tmp_path, db):
    a, b = str(tmp_path / "iter.raw"), str(tmp_path / "native.raw")
    H.xapian_ref("export", db, a)
    _lib.check(_lib.lib().xgm_glass_export_raw(db.encode(), b.encode()))
    return open(a, "rb").read(), open(b, "rb").read()


def test_native_reader_matches_iterator_export(built, tmp_path):
    """Synthetic corpus: multi-level B-trees, posting lists of many chunks, interpolative-coded positions."""
    db = str(tmp_path / "db")
    H.xapian_ref("build", db, H.CORPUS_SEED, 12000, 50000, 50, 150)
    a, b = both_exports(tmp_path, db)
    assert a == b
    seg_a, seg_b = str(tmp_path / "a.seg"), str(tmp_path / "b.seg")
    _lib.check(_lib.lib().xgm_segment_build_from_file(str(tmp_path / "iter.raw").encode(), 0, seg_a.encode()))
    _lib.check(_lib.lib().xgm_segment_build_from_glass(db.encode(), 0, seg_b.encode()))
    assert open(seg_a, "rb").read() == open(seg_b, "rb").read()


def test_glass_info_and_export_cli(built, tmp_path):
    """xgm_glass_info reads what the reference reports for the shard, and the command-line exporter writes a
    segment carrying that revision."""
    import ctypes as C
    import json
    import subprocess
    import sys
    db = str(tmp_path / "db")
    ref = json.loads(H.xapian_ref("build", db, H.CORPUS_SEED, 700, 5000, 50, 150))
    rev, dc, ld, tl = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint64()
    _lib.check(_lib.lib().xgm_glass_info(db.encode(), C.byref(rev), C.byref(dc), C.byref(ld), C.byref(tl)))
    assert (dc.value, ld.value, tl.value) == (ref["doccount"], ref["lastdocid"], ref["total_length"])
    seg = str(tmp_path / "cli.seg")
    r = subprocess.run([sys.executable, "-m", "xapiand_amd.export", db, seg], capture_output=True, text=True, cwd=H.ROOT)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["revision"] == rev.value and out["doccount"] == ref["doccount"] and os.path.getsize(seg) == out["segment_bytes"]


def test_native_reader_format_corners(built, tmp_path):
    """Several commits, deleted / replaced documents, docids beyond 0x8000 and 0x200000 (longer chunk keys),
    terms with zero bytes, postings without positions, boolean terms."""
    db = str(tmp_path / "misc")
    H.xapian_ref("build_misc", db)
    a, b = both_exports(tmp_path, db)
    assert a == b


@pytest.mark.parametrize("variant", ["empty", "nopos"])
def test_native_reader_degenerate_databases(built, tmp_path, variant):
    """A committed database without documents, and one without any positional information."""
    db = str(tmp_path / variant)
    H.xapian_ref("build_misc", db, variant)
    a, b = both_exports(tmp_path, db)
    assert a == b


def test_native_reader_rejects_garbage(built, tmp_path):
    d = tmp_path / "notglass"
    d.mkdir()
    (d / "iamglass").write_bytes(b"definitely not a glass version file" * 3)
    assert _lib.lib().xgm_glass_export_raw(str(d).encode(), str(tmp_path / "x.raw").encode()) == _lib.XGM_E_INVALID
    assert _lib.lib().xgm_glass_export_raw(str(tmp_path / "missing").encode(), str(tmp_path / "x.raw").encode()) == _lib.XGM_E_IO


def test_statistics_bounds_are_glass_own(built, tmp_path):
    """After deletes / replaces glass keeps its (now loose) doclength lower bound and wdf upper bound; the reference's
    BM25Weight::get_maxpart — hence MSet::get_max_possible — is computed from those.  The exporter carries them from the
    version file into the segment: the planner's per-term wdf bound and the doclen bound must equal what
    the reference reports, not the tight bounds of the live postings."""
    import ctypes as C
    import json
    db = str(tmp_path / "misc")
    H.xapian_ref("build_misc", db)
    terms = ["w1", "w7", "w33", "mixed", "nopos", "Keven"]
    ref = json.loads(H.xapian_ref("info", db, *terms))
    seg = str(tmp_path / "misc.seg")
    _lib.check(_lib.lib().xgm_segment_build_from_glass(db.encode(), 0, seg.encode()))
    h = C.c_void_p()
    _lib.check(_lib.lib().xgm_index_open(seg.encode(), _lib.XGM_DEVICE_NONE, _lib.UINT64_MAX, C.byref(h)))
    info = _lib.IndexInfo()
    _lib.check(_lib.lib().xgm_index_get_info(h, C.byref(info)))
    assert info.doclen_lower_bound == ref["doclength_lower_bound"]
    assert info.revision == ref["revision"]
    for t in terms:
        ub = C.c_uint32()
        _lib.check(_lib.lib().xgm_lookup_term(h, t.encode(), len(t), None, None, None, C.byref(ub)))
        assert ub.value == ref["wdf_upper_bound"][t], t
    _lib.lib().xgm_index_close(h)


@pytest.mark.parametrize("mutate", [False, True])
def test_incremental_refresh_equals_full_export(built, tmp_path, mutate):
    """A shard one revision later — documents appended (and, second case, documents at or above a floor deleted / replaced):
    the segment refreshed from the OLD segment + only the changed part of glass is byte for byte the full export of the new
    revision; a floor that breaks the contract (a changed document below it) is refused."""
    db = str(tmp_path / "db")
    H.xapian_ref("build", db, H.CORPUS_SEED, 9000, 40000, 50, 150)
    L = _lib.lib()
    seg1 = str(tmp_path / "rev1.seg")
    _lib.check(L.xgm_segment_build_from_glass(db.encode(), 0, seg1.encode()))
    floor = 6001 if mutate else 9001
    import json
    out = json.loads(H.xapian_ref("append", db, H.CORPUS_SEED, 20001, 23000, 40000, 50, 150, *([floor] if mutate else [])))
    assert out["last_before"] == 9000 and out["lastdocid"] == 12000
    full, inc = str(tmp_path / "rev2_full.seg"), str(tmp_path / "rev2_inc.seg")
    _lib.check(L.xgm_segment_build_from_glass(db.encode(), 0, full.encode()))
    _lib.check(L.xgm_segment_refresh_from_glass(seg1.encode(), db.encode(), floor, 0, inc.encode()))
    assert open(full, "rb").read() == open(inc, "rb").read()
    # in place (the old segment file is the output): same bytes again
    import shutil
    same = str(tmp_path / "in_place.seg")
    shutil.copy(seg1, same)
    _lib.check(L.xgm_segment_refresh_from_glass(same.encode(), db.encode(), floor, 0, same.encode()))
    assert open(full, "rb").read() == open(same, "rb").read()
    # floor 1 = nothing taken from the old segment: still the same bytes
    _lib.check(L.xgm_segment_refresh_from_glass(seg1.encode(), db.encode(), 1, 0, inc.encode()))
    assert open(full, "rb").read() == open(inc, "rb").read()
    if mutate:
        # documents 6006 (deleted) / 6017 (replaced) changed: a floor above them must be refused, not silently produce another index
        assert L.xgm_segment_refresh_from_glass(seg1.encode(), db.encode(), 9001, 0, inc.encode()) == _lib.XGM_E_INVALID
    assert L.xgm_segment_refresh_from_glass(seg1.encode(), db.encode(), 9500, 0, inc.encode()) == _lib.XGM_E_INVALID   # beyond the old segment


def test_incremental_refresh_keeps_format_corners(built, tmp_path):
    """The old part holds what the segment cannot give back verbatim — terms whose positions were dropped (postings without
    positions, boolean terms), zero bytes in terms, long docid gaps: the refreshed segment is still the full export's."""
    db = str(tmp_path / "misc")
    H.xapian_ref("build_misc", db)
    L = _lib.lib()
    seg1 = str(tmp_path / "rev1.seg")
    _lib.check(L.xgm_segment_build_from_glass(db.encode(), 0, seg1.encode()))
    import json
    out = json.loads(H.xapian_ref("append", db, H.CORPUS_SEED, 1, 400, 3000, 20, 60))
    full, inc = str(tmp_path / "rev2_full.seg"), str(tmp_path / "rev2_inc.seg")
    _lib.check(L.xgm_segment_build_from_glass(db.encode(), 0, full.encode()))
    _lib.check(L.xgm_segment_refresh_from_glass(seg1.encode(), db.encode(), out["last_before"] + 1, 0, inc.encode()))
    assert open(full, "rb").read() == open(inc, "rb").read()


def test_incremental_refresh_skips_unread_blocks_around_split_items(built, tmp_path):
    """Under a floor the position table is mostly passed over: leaf blocks between two separators of one term below the floor are not
    read at all, blocks that begin and end in one term are dropped on their first and last key, other entries on their raw key.  Here
    the position lists are several KB each, which glass stores as several B-tree items that straddle the leaf blocks — every way of
    passing over must leave the walk on an item boundary: floors inside the old part (documents re-read from glass), at its end, and
    with appended documents; always the full export's bytes."""
    import json
    db = str(tmp_path / "longpos")
    H.xapian_ref("build_misc", db, "longpos")
    L = _lib.lib()
    seg1 = str(tmp_path / "rev1.seg")
    _lib.check(L.xgm_segment_build_from_glass(db.encode(), 0, seg1.encode()))
    full, inc = str(tmp_path / "full.seg"), str(tmp_path / "inc.seg")
    for floor in (2, 150, 151, 299, 300, 301):                # nothing changed yet: any floor must give back the same segment
        _lib.check(L.xgm_segment_refresh_from_glass(seg1.encode(), db.encode(), floor, 0, inc.encode()))
        assert open(seg1, "rb").read() == open(inc, "rb").read(), floor
    out = json.loads(H.xapian_ref("append", db, H.CORPUS_SEED, 1, 200, 3000, 20, 60))
    assert out["last_before"] == 300
    _lib.check(L.xgm_segment_build_from_glass(db.encode(), 0, full.encode()))
    for floor in (301, 200, 7):
        _lib.check(L.xgm_segment_refresh_from_glass(seg1.encode(), db.encode(), floor, 0, inc.encode()))
        assert open(full, "rb").read() == open(inc, "rb").read(), floor


def test_incremental_refresh_copies_whole_stripes(built, tmp_path):
    """Narrow stripes (2^10 docids): the old segment's blocks of the stripes below the floor's are copied verbatim, the old postings
    between that stripe's start and the floor are re-encoded together with glass's — deletes and replaces above the floor included."""
    import json
    db = str(tmp_path / "db")
    H.xapian_ref("build", db, H.CORPUS_SEED, 9000, 40000, 50, 150)
    L = _lib.lib()
    seg1 = str(tmp_path / "rev1.seg")
    _lib.check(L.xgm_segment_build_from_glass(db.encode(), 10, seg1.encode()))
    json.loads(H.xapian_ref("append", db, H.CORPUS_SEED, 20001, 21500, 40000, 50, 150, 5500))
    full, inc = str(tmp_path / "rev2_full.seg"), str(tmp_path / "rev2_inc.seg")
    _lib.check(L.xgm_segment_build_from_glass(db.encode(), 10, full.encode()))
    for floor in (5500, 5121, 5120, 4097, 2):
        _lib.check(L.xgm_segment_refresh_from_glass(seg1.encode(), db.encode(), floor, 10, inc.encode()))
        assert open(full, "rb").read() == open(inc, "rb").read(), floor
    assert L.xgm_segment_refresh_from_glass(seg1.encode(), db.encode(), 5500, 11, inc.encode()) == _lib.XGM_E_INVALID   # another stripe width


def test_incremental_refresh_when_the_reason_positions_were_dropped_is_gone(built, tmp_path):
    """The old segment keeps no positions for a term that had a posting without them ("mixed": add_term beside add_posting).  When
    every such document is replaced above the floor, the full export of the new revision stores the term's positions again — the
    refresh cannot get them out of the old segment and must read that term from glass in full."""
    import json
    db = str(tmp_path / "misc")
    H.xapian_ref("build_misc", db)
    L = _lib.lib()
    seg1 = str(tmp_path / "rev1.seg")
    _lib.check(L.xgm_segment_build_from_glass(db.encode(), 0, seg1.encode()))
    json.loads(H.xapian_ref("append", db, H.CORPUS_SEED, 1, 50, 3000, 20, 60, 10))      # from docid 10 on: every 7th deleted, every 11th replaced
    full, inc = str(tmp_path / "rev2_full.seg"), str(tmp_path / "rev2_inc.seg")
    _lib.check(L.xgm_segment_build_from_glass(db.encode(), 0, full.encode()))
    _lib.check(L.xgm_segment_refresh_from_glass(seg1.encode(), db.encode(), 10, 0, inc.encode()))
    assert open(full, "rb").read() == open(inc, "rb").read()


def test_native_value_column_matches_the_value_iterator(built, tmp_path):
    """Value slots (widening row (f).3): the column the device will sort / collapse by — per document the rank of its value
    among the slot's distinct values — read natively from glass's value chunks equals, byte for byte, the one made through the
    reference's ValueIterator: several commits (chunks split and re-written), deleted and replaced documents, a sparse slot."""
    import json
    db = str(tmp_path / "db")
    H.xapian_ref("build_values", db, H.CORPUS_SEED, 9000, 20000, 20, 60)
    H.xapian_ref("append", db, H.CORPUS_SEED, 20001, 20400, 20000, 20, 60, 3000)      # deletes / replaces (the replacements carry no values)
    for slot in (0, 1, 2, 7):
        a, b = str(tmp_path / ("ref%d.col" % slot)), str(tmp_path / ("nat%d.col" % slot))
        json.loads(H.xapian_ref("column", db, slot, a))
        _lib.check(_lib.lib().xgm_glass_export_column(db.encode(), slot, b.encode()))
        assert open(a, "rb").read() == open(b, "rb").read(), slot
