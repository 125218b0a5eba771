"""Diagnostics: per work-unit timeline of the last batch launch of bench.py (occupancy over time,
cycles per stripe, slowest units).  Run on the GPU box: python tools/units.py [bench args]."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["XGM_DEBUG_UNITS"] = "1"
os.environ.setdefault("XGM_BENCH_NO_TALLY", "1")
import numpy as np  # noqa: E402

import bench  # noqa: E402
from xapiand_amd import _lib, enquire  # noqa: E402

sys.argv = ["bench.py", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"] + sys.argv[1:]
L = _lib.lib()
L.xgm_debug_last_units.restype = C.c_int64
L.xgm_debug_last_units.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_uint64]
oc = enquire.Database.close


def close(self):
    cap = 20000
    buf = (C.c_ulonglong * (8 * cap))()
    n = L.xgm_debug_last_units(self._h, buf, cap)
    if n > 0:
        a = np.array(buf[:8 * n], dtype=np.uint64).reshape(n, 8)
        t0 = a[:, 4].min()
        dur = (a[:, 5] - a[:, 4]).astype(np.float64)
        st = (a[:, 4] - t0).astype(np.float64)
        en = (a[:, 5] - t0).astype(np.float64)
        span = en.max()
        print("UNITS n", n, "span_cycles", span, "dur mean/p50/p95/max", dur.mean(), np.median(dur), np.percentile(dur, 95), dur.max(),
              "avg concurrency", dur.sum() / span)
        edges = np.linspace(0, span, 11)
        print("UNITS concurrency per decile", [int(((st < edges[i + 1]) & (en > edges[i])).sum()) for i in range(10)])
        stripes = (a[:, 2] - a[:, 1]).astype(np.float64)
        print("UNITS cycles/stripe mean", (dur / np.maximum(1, stripes)).mean(), "stripes/unit mean,max", stripes.mean(), stripes.max())
        print("UNITS matches", int(a[:, 6].sum()), "documents weighed (xgm_orw_kernel only)", int(a[:, 7].sum()),
              "ratio", float(a[:, 7].sum()) / max(1.0, float(a[:, 6].sum())))
        # least-squares fit of the unit time: cycles = a * stripes + b * matches + c (the planner's cost model, xgm_api.cc)
        A = np.stack([stripes, a[:, 6].astype(np.float64), np.ones(n)], axis=1)
        coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
        print("UNITS fit cycles = %.1f * stripes + %.2f * matches + %.0f; residual rms %.0f" % (coef[0], coef[1], coef[2], float(np.sqrt(((A @ coef - dur) ** 2).mean()))))
        ph_a = (a[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.float64) * 64.0
        ph_b = (a[:, 3] >> np.uint64(32)).astype(np.float64) * 64.0
        print("UNITS conjunction queue path: share of unit cycles in probe rounds %.3f, in weigh rounds %.3f (tallying build)" % (ph_a.sum() / dur.sum(), ph_b.sum() / dur.sum()))
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        np.save(os.path.join(out, "units_%s.npy" % os.environ.get("XGM_UNITS_TAG", "last")), a)
        order = np.argsort(-dur)[:5]
        print("UNITS slowest (qi, s_begin, s_end, cycles, start)", [(int(a[i, 0]), int(a[i, 1]), int(a[i, 2]), int(dur[i]), int(st[i])) for i in order])
    oc(self)


enquire.Database.close = close
bench.main()
