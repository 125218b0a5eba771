"""A pure C client of the C ABI (examples/xgm_search.c, no Python, no torch in the process) answers like the
oracle: the boundary really is a plain C library."""
import os
import shutil
import subprocess

import pytest

import helpers as H

ROOT = H.ROOT


@pytest.mark.gpu
def test_c_client_matches_oracle(built, tmp_path):
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    libdir = os.path.join(ROOT, "xapiand_amd", "csrc")
    exe = str(tmp_path / "xgm_search")
    subprocess.run([gcc, "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "xgm_search.c"),
                    "-L", libdir, "-lxgm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    c = H.Corpus(20000, 50000)
    seg = c.build_segment(str(tmp_path / "c.seg"))
    cases = [("AND", 0, ["t3", "t17", "t120"]), ("OR", 0, ["t9", "t40", "t300", "t2000", "t7"]), ("PHRASE", 0, ["t1", "t2"]),
             ("AND_NOT", 2, ["t3", "t17", "t9"]), ("AND_MAYBE", 1, ["t40", "t3", "t300"]), ("FILTER", 1, ["t40", "t3"])]
    for op, nreq, terms in cases:
        arg = op + (":%d" % nreq if nreq else "")
        r = subprocess.run([exe, seg, arg] + terms, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.strip().splitlines()
        got = [(int(l.split()[2]), float(l.split()[4])) for l in lines[1:]]
        want, hdr = H.oracle_search(c, op, terms, 0, 10, n_required=nreq)
        assert got == [(d, w) for d, w, _ in want], (op, terms)
        shown = int(lines[0].split()[0])
        assert shown == hdr.matches if "(at least)" not in lines[0] else len(got) <= shown <= hdr.matches
