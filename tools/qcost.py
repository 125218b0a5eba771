"""Diagnostics: which QUERIES a batch's time goes to.  Per query of the last batch bench.py launched: wave cycles summed over its work
units, its share of the batch, its terms' ranks, whether every term has probe containers (the dense body) — heaviest first.
Run on the GPU box: python tools/qcost.py --op PHRASE --topk 10   (or --op OR --terms 5 --topk 100, ...)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["XGM_DEBUG_UNITS"] = "1"
os.environ.setdefault("XGM_BENCH_NO_TALLY", "1")
import numpy as np  # noqa: E402

import bench  # noqa: E402
import helpers as H  # noqa: E402
from xapiand_amd import _lib, enquire  # noqa: E402

user = sys.argv[1:]
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs", "--no-latency", "--threads", "0"] + user
L = _lib.lib()
L.xgm_debug_last_units2.restype = C.c_int64
L.xgm_debug_last_units2.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_uint64]
CLOCKS = os.environ.get("XGM_QCOST_CLOCKS") is not None       # the library is a -DXGM_DENSE_CLOCKS build: cycles per phase of the dense body in the headers
oc = enquire.Database.close


def arg(name, default):
    return user[user.index(name) + 1] if name in user else default


def close(self):
    if os.environ.get("XGM_QCOST_LIST"):
        # the reference-identical batch mode (xgm_andw_list_kernel): the same 256 queries once more with XGM_REPLAY_BATCH_FROZEN, then its units
        pool_ = H.bench_pool("PHRASE", 3, 1, 10_000_000, 1_000_000, n=100 + 16 * bench.BATCH, seed=bench.QUERY_SEED, maxitems=int(arg("--topk", 10)))
        qs_ = pool_[100 + 3 * bench.BATCH: 100 + 4 * bench.BATCH]
        plans_ = [enquire.plan(self, enquire.Query(q["op"], q["terms"]), 0, int(arg("--topk", 10))) for q in qs_]
        self.set_stream(0)
        enquire.search_batch_replay(self, plans_)
    cap = 40000
    buf = (C.c_ulonglong * (8 * cap))()
    pos = (C.c_ulonglong * cap)()
    n = L.xgm_debug_last_units2(self._h, buf, pos, cap)
    if n > 0:
        a = np.array(buf[:8 * n], dtype=np.uint64).reshape(n, 8)
        dur = (a[:, 5] - a[:, 4]).astype(np.float64)
        qi = a[:, 0].astype(np.int64)
        nq = int(qi.max()) + 1
        per_q = np.bincount(qi, weights=dur, minlength=nq)
        units_q = np.bincount(qi, minlength=nq)
        longest = np.zeros(nq)
        np.maximum.at(longest, qi, dur)
        matches_q = np.bincount(qi, weights=(a[:, 6] & np.uint64((1 << 62) - 1)).astype(np.float64), minlength=nq)
        tested_q = np.bincount(qi, weights=(a[:, 7] & np.uint64(0xFFFFF)).astype(np.float64), minlength=nq)
        calls_q = np.bincount(qi, weights=(a[:, 7] >> np.uint64(20)).astype(np.float64), minlength=nq)     # -DXGM_DENSE_CLOCKS=2 builds: survivors() calls       # positional kernels: candidates whose positions were tested
        cpos = np.array(pos[:n], dtype=np.uint64)
        ph = {"produce": (a[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.float64), "issue": (a[:, 3] >> np.uint64(32)).astype(np.float64),
              "weigh": (cpos & np.uint64(0xFFFFFFFF)).astype(np.float64), "survivors": (cpos >> np.uint64(32)).astype(np.float64)}
        op = arg("--op", "AND")
        bps = 16
        pool = H.bench_pool(op, int(arg("--terms", 3)), int(arg("--required", 1)), 10_000_000, 1_000_000, n=100 + bps * bench.BATCH, seed=bench.QUERY_SEED,
                            maxitems=int(arg("--topk", 10)))
        last_batch = 3                                        # measure(): the byte-count passes run batches 0..3 last
        qs = pool[100 + last_batch * bench.BATCH: 100 + (last_batch + 1) * bench.BATCH]
        total = per_q.sum()
        order = np.argsort(-per_q)
        dense_min = 39_000                                    # df from which a term has containers at 10 M documents
        print("QCOST batch: %d units, %d queries, total wave cycles %.3g, span %.3g cycles" % (n, nq, total, float(a[:, 5].max() - a[:, 4].min())))
        cum = 0.0
        for rank, q in enumerate(order[:24]):
            cum += per_q[q]
            terms = qs[q]["terms"] if q < len(qs) else []
            ranks = [int(t[1:]) for t in terms]
            dfs = [int(self.get_termfreq(t)) for t in terms]
            print("QCOST #%2d q%3d share %.3f cum %.3f units %4d longest unit %8.0f mean %8.0f matches %9d tested %8d ranks %s dfs %s all-dense %s" % (
                rank, q, per_q[q] / total, cum / total, units_q[q], longest[q], per_q[q] / max(1, units_q[q]), matches_q[q], tested_q[q], ranks, dfs,
                all(d >= dense_min for d in dfs)))
            if CLOCKS:
                sel = qi == q
                print("QCOST      phases of its units' cycles: " + "  ".join("%s %.3f" % (k, v[sel].sum() / max(1.0, per_q[q])) for k, v in ph.items()) + "  survivors() calls %d" % calls_q[q])
        dense_q = np.array([all(int(self.get_termfreq(t)) >= dense_min for t in qs[q]["terms"]) if q < len(qs) else False for q in range(nq)])
        print("QCOST all-dense queries: %d of %d, their share of the cycles %.3f" % (int(dense_q.sum()), nq, per_q[dense_q].sum() / total))
        print("QCOST cycles per unit: mean %.0f p50 %.0f p99 %.0f max %.0f" % (dur.mean(), np.median(dur), np.percentile(dur, 99), dur.max()))
    oc(self)


enquire.Database.close = close
bench.main()
