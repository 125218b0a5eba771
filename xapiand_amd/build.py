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
SOURCES = ["xgm_api.cc", "xgm_plan.cc", "xgm_segment_build.cc", "xgm_glass.cc", "xgm_kernels.hip", "xgm_dense_and.hip", "xgm_or.hip", "xgm_synth.hip", "xgm_dense.hip", "xgm_all.hip", "xgm_replay.hip", "xgm_frozen.hip", "xgm_count.hip"]
# -ffp-contract=off: BM25 must round exactly like the reference's separate mul/add/div (no FMA).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
         "-x", "hip"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libxgm.so")


def _digest(paths, extra=""):
    """Content hash of the inputs of one build step.  Staleness is decided by content, not by mtime: the
    tree is copied to the GPU box, where file times no longer say anything about build order."""
    import hashlib
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _up_to_date(target, digest):
    try:
        return os.path.exists(target) and open(target + ".sha").read().strip() == digest
    except OSError:
        return False


def build(force=False, verbose=False):
    """Compile every HIP/C++ source for gfx950 and link libxgm.so.  Returns the library path."""
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h") or f.endswith(".inc")]
    headers += [os.path.join(HERE, "..", "include", "xgm.h"), os.path.join(HERE, "..", "tools", "xgm_corpus.h")]
    objs, digests = [], []
    hipcc = None
    all_dg = [_digest([os.path.join(CSRC, src)] + headers, " ".join(FLAGS)) for src in SOURCES]
    if not force and _up_to_date(LIB, _digest([], " ".join(all_dg))):
        return LIB                      # the shipped library matches the sources (objects need not be present)
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        dg = _digest([s] + headers, " ".join(FLAGS))
        digests.append(dg)
        if force or not _up_to_date(o, dg):
            hipcc = hipcc or _hipcc()
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append((subprocess.Popen(cmd), cmd, o, dg))       # translation units compile side by side
    failed = [cmd for proc, cmd, _, _ in jobs if proc.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    for _, _, o, dg in jobs:
        with open(o + ".sha", "w") as f:
            f.write(dg)
    link_dg = _digest([], " ".join(digests))
    if force or not _up_to_date(LIB, link_dg):
        hipcc = hipcc or _hipcc()
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        with open(LIB + ".sha", "w") as f:
            f.write(link_dg)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
