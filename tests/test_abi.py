"""CPU-only: the C-ABI library loads and exports every symbol include/xgm.h declares; the structs
the Python host mirrors have the sizes the header implies; search refuses to run without a GPU."""
import ctypes as C
import os
import re

import pytest

from xapiand_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "xgm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(xgm_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(built):
    l = _lib.lib()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(l, s), "libxgm.so does not export %s" % s
    assert sorted(_lib.api_symbols()) == syms, "python binding and header disagree"


def test_struct_layout(built):
    assert C.sizeof(_lib.Hit) == 16
    assert C.sizeof(_lib.ResultHdr) == 32
    assert C.sizeof(_lib.Term) == 16
    assert _lib.lib().xgm_version().startswith(b"xgm")


def test_search_fails_loudly_without_device(built, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = _lib.lib().xgm_index_open(b"/nonexistent.seg", 0, _lib.UINT64_MAX, C.byref(h))
    assert rc == _lib.XGM_E_IO
    import helpers as H
    c = H.Corpus(300, 2000)
    seg = c.build_segment(str(tmp_path / "a.seg"))
    rc = _lib.lib().xgm_index_open(seg.encode(), 0, _lib.UINT64_MAX, C.byref(h))
    assert rc == _lib.XGM_E_NO_DEVICE          # no CPU fallback
    assert b"HIP" in _lib.lib().xgm_last_error()


def test_header_is_plain_c_and_the_example_links(built, tmp_path):
    """include/xgm.h must be a C header (the drop-in boundary is `extern "C"`, plain pointers and sizes):
    examples/xgm_search.c is compiled as pedantic C99 and linked against libxgm.so."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    src = os.path.join(ROOT, "examples", "xgm_search.c")
    exe = str(tmp_path / "xgm_search")
    libdir = os.path.join(ROOT, "xapiand_amd", "csrc")
    subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), src,
                    "-L", libdir, "-lxgm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
