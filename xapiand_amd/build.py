"""In-tree build of libxgm.so (HIP kernels + C ABI) for gfx950 with hipcc.

The shared library is written next to the sources (xapiand_amd/csrc/libxgm.so) so that it travels
to the GPU box with the repository snapshot.  Nothing is JIT-compiled at import time.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libxgm.so")
SOURCES = ["xgm_api.cc", "xgm_plan.cc", "xgm_segment_build.cc", "xgm_kernels.hip", "xgm_or.hip", "xgm_synth.hip", "xgm_dense.hip"]
# -ffp-contract=off: BM25 must round exactly like the reference's separate mul/add/div (no FMA).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
         "-x", "hip"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libxgm.so")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP/C++ source for gfx950 and link libxgm.so.  Returns the library path."""
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(HERE, "..", "include", "xgm.h"), os.path.join(HERE, "..", "tools", "xgm_corpus.h")]
    objs = []
    hipcc = None
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            hipcc = hipcc or _hipcc()
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        hipcc = hipcc or _hipcc()
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
