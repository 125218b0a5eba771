"""CPU-only: the host segment builder (exporter back end) round-trips postings exactly."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
from xapiand_amd import _lib


def decode(seg, term, cap):
    did = (C.c_uint32 * cap)()
    wdf = (C.c_uint32 * cap)()
    n = _lib.lib().xgm_segment_decode_term(seg.encode(), term, len(term), did, wdf, cap)
    assert n >= 0
    return np.array(did[:n]), np.array(wdf[:n])


@pytest.mark.parametrize("stripe_bits", [8, 10, 13])
def test_roundtrip_every_term(built, tmp_path, stripe_bits):
    c = H.Corpus(3000, 5000)
    seg = c.build_segment(str(tmp_path / "c.seg"), stripe_bits=stripe_bits)
    terms = c.terms()
    df = c.df_array()
    for i in list(range(0, len(terms), max(1, len(terms) // 200))) + [int(np.argmax(df))]:
        did, wdf = c.term_postings(i)
        gd, gw = decode(seg, terms[i], int(df[i]) + 4)
        assert np.array_equal(gd, did) and np.array_equal(gw, wdf), terms[i]
    gd, _ = decode(seg, b"no-such-term", 4)
    assert len(gd) == 0


def test_raw_file_form_matches_in_memory(built, tmp_path):
    if not H.have_xapian_ref():
        pytest.skip("oracle/_ref not built")
    db = str(tmp_path / "db")
    H.xapian_ref("build", db, H.CORPUS_SEED, 1500, 3000, 50, 150)
    raw = str(tmp_path / "db.raw")
    H.xapian_ref("export", db, raw)                      # real Xapian iterators → raw postings
    seg_a = str(tmp_path / "a.seg")
    _lib.check(_lib.lib().xgm_segment_build_from_file(raw.encode(), 0, seg_a.encode()))
    c = H.Corpus(1500, 3000)
    seg_b = c.build_segment(str(tmp_path / "b.seg"), revision=1)
    a, b = open(seg_a, "rb").read(), open(seg_b, "rb").read()
    # identical except the revision field of the header
    hdr = _lib.IndexInfo  # noqa: F841
    assert len(a) == len(b)
    diff = [i for i in range(len(a)) if a[i] != b[i]]
    assert all(56 <= i < 64 for i in diff), diff[:10]    # offsetof(xgm_seg_header, revision) == 56


def test_bad_inputs_are_rejected(built, tmp_path):
    c = H.Corpus(200, 500)
    r = c.raw_postings()
    assert _lib.lib().xgm_segment_build(C.byref(r), 20, str(tmp_path / "x.seg").encode()) == _lib.XGM_E_INVALID
    bad = tmp_path / "bad.seg"
    bad.write_bytes(b"not a segment" * 100)
    did = (C.c_uint32 * 4)()
    assert _lib.lib().xgm_segment_decode_term(str(bad).encode(), b"t1", 2, did, did, 4) == _lib.XGM_E_INVALID
