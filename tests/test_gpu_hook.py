"""Drop-in check of seam B2: the REAL vendored Xapian of the reference runs each query
through its own Enquire::get_mset twice — with its CPU matcher and with the query replaced by
integration/GpuTopKPostingSource (libxgm.so behind the reference's PostingSource plug-in API) — on a
glass database it built itself, exported through its public iterators into a device segment.  The
binary (oracle/_ref/xapian_hook) is built where /root/reference exists and travels to the GPU box."""
import json
import os
import subprocess

import pytest

import helpers as H
from xapiand_amd import _lib

HOOK = os.path.join(H.ROOT, "oracle", "_ref", "xapian_hook")


@pytest.mark.gpu
def test_reference_enquire_with_gpu_posting_source(built, tmp_path):
    if not (H.have_xapian_ref() and os.path.exists(HOOK)):
        pytest.skip("oracle/_ref not built")
    db = str(tmp_path / "db")
    H.xapian_ref("build", db, H.CORPUS_SEED, 20000, 50000, 50, 150)
    raw, seg = str(tmp_path / "db.raw"), str(tmp_path / "db.seg")
    H.xapian_ref("export", db, raw)
    _lib.check(_lib.lib().xgm_segment_build_from_file(raw.encode(), 0, seg.encode()))
    qs = (H.gen_term_queries("AND", 40, 3, 1, 300, maxitems=10, seed=91) + H.gen_term_queries("OR", 25, 5, 1, 2000, maxitems=100, seed=92) +
          H.gen_term_queries("AND", 10, 2, 1, 50, first=20, maxitems=10, seed=93) + H.gen_term_queries("OR", 5, 2, 500, 4000, maxitems=10, seed=94))
    # PHRASE: the full ranking only (maxitems >= matches) — with fewer the reference's own top-k is not a
    # prefix of its ranking (SelectPostList stale weight)
    corpus = H.Corpus(20000, 50000)
    for q in H.gen_phrase_queries(40, 20000, 50000, seed=95):
        q = dict(q, maxitems=150)
        if H.oracle_search(corpus, q["op"], q["terms"], 0, 150, window=q.get("window", 0))[1].matches <= 150:
            qs.append(q)
    assert sum(q["op"] == "PHRASE" for q in qs) >= 10
    qs.append(dict(op="AND", terms=["t5", "t5"], first=0, maxitems=10))          # a shape the device path declines: stays on the CPU matcher
    qfile = tmp_path / "q.txt"
    qfile.write_text("".join("%s %d %d %d %s\n" % (q["op"], q["first"], q["maxitems"], q.get("window", 0), " ".join(q["terms"])) for q in qs))
    r = subprocess.run([HOOK, db, seg, str(qfile)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["queries"] == len(qs) and out["mismatches"] == 0 and out["declined"] == 1, out
