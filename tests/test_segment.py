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


def test_corrupt_segments_are_rejected_at_open(built, tmp_path):
    """A cached segment file is untrusted input: truncation, section tables that overrun their sections, non-monotone
    term tables and block headers pointing outside their term must all fail with XGM_E_INVALID before anything is
    adopted (host-only open: no GPU needed)."""
    import struct
    c = H.Corpus(400, 800)
    good = c.build_segment(str(tmp_path / "good.seg"))
    data = bytearray(open(good, "rb").read())

    def open_rc(blob, name):
        p = tmp_path / name
        p.write_bytes(bytes(blob))
        h = C.c_void_p()
        rc = _lib.lib().xgm_index_open(str(p).encode(), _lib.XGM_DEVICE_NONE, _lib.UINT64_MAX, C.byref(h))
        if rc == 0:
            _lib.lib().xgm_index_close(h)
        return rc
    assert open_rc(data, "ok.seg") == 0
    # header layout (xgm_segment.h): magic[8], 4 x u32, 4 x u32, 2 x u32, 7 x u64, then sec_off[16], sec_bytes[16]
    off_counts = 8 + 16 + 16 + 8
    n_off = off_counts + 7 * 8
    sec_off = list(struct.unpack_from("<16Q", data, n_off))
    sec_bytes = list(struct.unpack_from("<16Q", data, n_off + 128))
    assert open_rc(data[: len(data) // 2], "trunc.seg") == _lib.XGM_E_INVALID
    t = bytearray(data)
    struct.pack_into("<I", t, 8 + 12, struct.unpack_from("<I", data, 8 + 12)[0] + 1000)        # n_terms inflated beyond the tables
    assert open_rc(t, "nterms.seg") == _lib.XGM_E_INVALID
    t = bytearray(data)
    struct.pack_into("<Q", t, n_off + 128 + 8 * 12, 2 ** 63)                                    # WORDS section size wraps offset + size
    assert open_rc(t, "wrap.seg") == _lib.XGM_E_INVALID
    t = bytearray(data)
    tb = sec_off[5]                                                                              # TERM_BLK: make it non-monotone
    struct.pack_into("<Q", t, tb + 8, 2 ** 40)
    assert open_rc(t, "mono.seg") == _lib.XGM_E_INVALID
    t = bytearray(data)
    struct.pack_into("<I", t, sec_off[10], 0x7FFFFFFF)                                           # first block's payload offset outside its term
    assert open_rc(t, "blkword.seg") == _lib.XGM_E_INVALID
    t = bytearray(data)
    struct.pack_into("<I", t, sec_off[8] + 4 * (sec_bytes[8] // 4 - 1), 0)                       # a block whose first docid is 0
    assert open_rc(t, "docid.seg") == _lib.XGM_E_INVALID
