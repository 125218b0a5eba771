"""Build a glass index of the synthetic corpus with the REAL reference (oracle/_ref/xapian_ref) on many host cores:
contiguous slices of the corpus are indexed by parallel `xapian_ref build_range` processes (Xapian's
WritableDatabase::add_document is single-threaded: ~1.7 k documents/s per core) and merged with
`xapian_ref compact` (Database::compact), which yields the same documents under the same docids as one sequential
build (checked in tests/test_oracle_vs_reference.py).  Test/bench infrastructure only — bench.py's cpu_baseline leg
uses it to put a reference index next to the GPU on the box it runs on.

    python tools/ref_index.py <outdir> <n_docs> [--procs P] [--nopos] [--vocab V]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
XAPIAN_REF = os.path.join(ROOT, "oracle", "_ref", "xapian_ref")
CORPUS_SEED = 0x5EED0001


def build(outdir, n_docs, procs=None, nopos=False, vocab=1_000_000, len_lo=50, len_hi=150, seed=CORPUS_SEED, keep_parts=False, budget_s=None, n_shards=1, shard=0):
    """Returns dict(doccount, build_s, compact_s, procs).  outdir is replaced.  n_shards > 1: the index of round-robin shard `shard` of the n_docs-document
    corpus (global document g lives in shard (g - 1) % n_shards under local docid (g - 1) // n_shards + 1: backends/multi.h:38-73).  budget_s: give up (TimeoutError, the writers this call
    started are killed) when indexing + compaction take longer — a caller with a time limit of its own (bench.py) falls back."""
    if not os.path.exists(XAPIAN_REF):
        raise RuntimeError("oracle/_ref/xapian_ref is not built")
    # more writers than ~1/2 of the cores fight each other (measured on the 256-core MI355X host: 254 processes index
    # 10 M documents in 185 s, 50 processes 1 M in 9.5 s)
    procs = procs or max(1, min((os.cpu_count() or 2) // 2, (n_docs + 19999) // 20000))
    parts_dir = outdir.rstrip("/") + ".parts"
    shutil.rmtree(parts_dir, ignore_errors=True)
    shutil.rmtree(outdir, ignore_errors=True)
    os.makedirs(parts_dir)
    per = (n_docs + procs - 1) // procs
    t0 = time.time()
    running, parts = [], []
    for i in range(procs):
        g0, g1 = i * per + 1, min(n_docs, (i + 1) * per)
        if g0 > g1:
            break
        d = os.path.join(parts_dir, "p%04d" % i)
        parts.append(d)
        cmd = [XAPIAN_REF, "build_range", d, hex(seed), str(g0), str(g1), str(vocab), str(len_lo), str(len_hi)] + (["nopos"] if nopos else ["pos"] if n_shards > 1 else [])
        if n_shards > 1:
            cmd += [str(n_shards), str(shard)]
        running.append(subprocess.Popen(cmd, stdout=subprocess.DEVNULL))
    deadline = None if budget_s is None else t0 + budget_s
    for p in running:
        try:
            rc = p.wait(timeout=None if deadline is None else max(0.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            for q in running:
                if q.poll() is None:
                    q.kill()
            shutil.rmtree(parts_dir, ignore_errors=True)
            raise TimeoutError("indexing %d documents exceeded %.0f s" % (n_docs, budget_s))
        if rc != 0:
            raise RuntimeError("xapian_ref build_range failed")
    t1 = time.time()
    if len(parts) == 1:
        os.rename(parts[0], outdir)
        info = dict(doccount=n_docs if n_shards == 1 else (n_docs - shard + n_shards - 1) // n_shards)
    else:
        try:
            out = subprocess.run([XAPIAN_REF, "compact", outdir] + parts, check=True, capture_output=True, text=True,
                                 timeout=None if deadline is None else max(1.0, deadline - time.time())).stdout
        except subprocess.TimeoutExpired:
            shutil.rmtree(parts_dir, ignore_errors=True)
            raise TimeoutError("indexing + compacting %d documents exceeded %.0f s" % (n_docs, budget_s))
        info = json.loads(out)
    t2 = time.time()
    if not keep_parts:
        shutil.rmtree(parts_dir, ignore_errors=True)
    want_docs = n_docs if n_shards == 1 else (n_docs - shard + n_shards - 1) // n_shards
    assert info["doccount"] == want_docs, (info, want_docs)
    return dict(doccount=want_docs, build_s=t1 - t0, compact_s=t2 - t1, procs=len(parts), positions=not nopos)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("n_docs", type=int)
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--nopos", action="store_true")
    ap.add_argument("--vocab", type=int, default=1_000_000)
    a = ap.parse_args()
    print(json.dumps(build(a.outdir, a.n_docs, a.procs or None, a.nopos, a.vocab)))
