"""Diagnostics: per QUERY of the last xgm_orw_kernel batch of bench.py — matches, documents weighed, the terms' ranks.
Run on the GPU box: python tools/orq.py --op OR --terms 5 --topk 100"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["XGM_DEBUG_UNITS"] = "1"
import numpy as np  # noqa: E402

import bench  # noqa: E402
import helpers as H  # noqa: E402
from xapiand_amd import _lib, enquire  # noqa: E402

user = sys.argv[1:]
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-latency", "--threads", "0"] + user
L = _lib.lib()
L.xgm_debug_last_units2.restype = C.c_int64
L.xgm_debug_last_units2.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_uint64]
oc = enquire.Database.close


def arg(name, default):
    return user[user.index(name) + 1] if name in user else default


def close(self):
    cap = 20000
    buf = (C.c_ulonglong * (8 * cap))()
    pos = (C.c_ulonglong * cap)()
    n = L.xgm_debug_last_units2(self._h, buf, pos, cap)
    if n > 0:
        a = np.array(buf[:8 * n], dtype=np.uint64).reshape(n, 8)
        fixed = ((a[:, 3] >> np.uint64(31)) & np.uint64(1)).astype(np.int64)
        fixw = (a[:, 3] & np.uint64(0x7FFFFFFF)).astype(np.int64)
        firstw = (a[:, 3] >> np.uint64(32)).astype(np.int64)
        print("ORQ units", n, "went round again", int(fixed.sum()), "weighed in second passes", int(fixw.sum()), "weighed in the units' first stripe", int(firstw.sum()))
        pool = H.bench_pool(arg("--op", "AND"), int(arg("--terms", 3)), int(arg("--required", 1)), 10_000_000, 1_000_000, seed=bench.QUERY_SEED)
        nb = (len(pool) - 100) // bench.BATCH
        lo = 100 + (nb - 1) * bench.BATCH                     # the tally loop's last batch is the last launch
        per = {}
        for qi, sb, se, _, ts, te, m, wgh in a:
            p = per.setdefault(int(qi), [0, 0, 0, 0])
            p[0] += int(m); p[1] += int(wgh); p[2] += 1; p[3] += int(te - ts)
        rows = []
        for qi, (m, wgh, nu, cyc) in per.items():
            ranks = sorted(int(t[1:]) for t in pool[lo + qi]["terms"])
            rows.append((wgh, m, nu, cyc, ranks))
        rows.sort(reverse=True)
        tot_w = sum(r[0] for r in rows); tot_m = sum(r[1] for r in rows)
        print("ORQ queries", len(rows), "matches", tot_m, "weighed", tot_w, "ratio", tot_w / max(1, tot_m))
        sp = [r for r in rows if r[4][-1] > 1800]
        print("ORQ with a block-decoded term (rank > 1800):", len(sp), "weighed", sum(r[0] for r in sp), "; without:", len(rows) - len(sp), "weighed", tot_w - sum(r[0] for r in sp))
        for r in rows[:12] + rows[len(rows) // 2 - 4:len(rows) // 2 + 4] + rows[-8:]:
            print("ORQ weighed %8d matches %8d (%.4f) units %3d kcycles %7d ranks %s" % (r[0], r[1], r[0] / max(1, r[1]), r[2], r[3] // 1000, r[4]))
        # per unit: weighed in the first vs later stripes cannot be separated; print the distribution of weighed per unit
        w = a[:, 7].astype(np.float64); st = (a[:, 2] - a[:, 1]).astype(np.float64)
        print("ORQ per unit: weighed mean %.0f p50 %.0f p95 %.0f max %.0f; stripes mean %.1f; weighed per stripe mean %.1f" % (w.mean(), np.median(w), np.percentile(w, 95), w.max(), st.mean(), (w / np.maximum(1, st)).mean()))
    oc(self)


enquire.Database.close = close
bench.main()
