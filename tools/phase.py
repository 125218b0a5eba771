import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["XGM_PHASE_TIMING"] = "1"
import bench
from xapiand_amd import _lib
sys.argv = ["bench.py", "--steps", "8", "--warmup", "2", "--no-cpu-baseline"] + sys.argv[1:]
# run the bench main, then fetch phase cycles
import io, contextlib
buf = io.StringIO()
orig_close = None
L = _lib.lib()
L.xgm_debug_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong)]
_z = (C.c_ulonglong * 8)()
L.xgm_debug_merge_cycles.argtypes = [C.POINTER(C.c_ulonglong)]
L.xgm_debug_merge_cycles(_z)        # start a fresh window (the merge timers' minimum needs its reset value)
# patch Database.close to fetch before closing
from xapiand_amd import enquire
oc = enquire.Database.close
def close(self):
    out = (C.c_ulonglong * 16)()
    L.xgm_debug_orw_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong)]
    if L.xgm_debug_orw_phase_cycles(out) == 0 and sum(out):
        v = list(out)
        tot = sum(v) or 1
        names = ["theta+bitmaps+sum", "decode+candset", "enumerate", "dense probes", "block scatter", "score-loop", "hist flush", "setup", "sort", "dlen+leaves", "tree", "hist+insert", "loads issued+hist bound+quantise", "bound sum (after the wait)", "wait for the bitmaps", "flat: zero+loads+atomics"]
        print("ORW PHASES:", {n: round(100.0 * x / tot, 1) for n, x in zip(names, v)}, "total Gcycles", round(tot / 1e9, 2))
    out = (C.c_ulonglong * 8)()
    L.xgm_debug_merge_cycles.argtypes = [C.POINTER(C.c_ulonglong)]
    if L.xgm_debug_merge_cycles(out) == 0 and sum(out):
        v = list(out)
        print("MERGE PHASES (cycles of thread 0, summed over workgroups and launches):", dict(zip(["headers", "gather", "threshold", "select", "rank+write"], v[:5])))
    out = (C.c_ulonglong * 8)()
    if L.xgm_debug_phase_cycles(out) == 0:
        v = list(out)
        tot = sum(v[:6]) or 1
        names = ["init", "P1work", "P1wait", "P3work", "P3wait", "P4+clear", "-"]
        print("PHASES(all launches since start):", {n: round(100.0 * x / tot, 1) for n, x in zip(names, v)}, "stripes", v[7],
              "cycles/stripe", round(sum(v[1:6]) / max(1, v[7])))
    oc(self)
enquire.Database.close = close
bench.main()
