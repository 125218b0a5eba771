"""Where xgm_search_replay's time goes at config size: python tools/replay_prof.py [OR|PHRASE] [n] — n queries of bench.py's C3 / C5 pool on the
10 M-document synthetic index, one at a time (run under rocprofv3 --kernel-trace --stats for the kernel split; prints the host-side wall time)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from xapiand_amd import Database, Query, _lib
from xapiand_amd.enquire import plan

op = sys.argv[1] if len(sys.argv) > 1 else "OR"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
docs = int(os.environ.get("XGM_PROF_DOCS", "10000000"))
db = Database.synthetic(0x5EED0001, docs, 1000000, device=0, with_positions=op == "PHRASE")
k = 100 if op == "OR" else 10
pool = H.bench_pool(op, 5 if op == "OR" else 3, 1, docs, 1000000, n=100 + n, seed=0x5EED0002, maxitems=k)[100:]
L = _lib.lib()
hits = (_lib.Hit * k)(); hdr = _lib.ResultHdr(); known = C.c_uint64()
plans = [plan(db, Query(q["op"], q["terms"]), 0, k) for q in pool]
mode = 1 if op == "PHRASE" else 0
for rep in range(2):
    t0 = time.perf_counter(); tot = 0
    for p in plans:
        _lib.check(L.xgm_search_replay(db._h, C.byref(p), mode, hits, C.byref(hdr), C.byref(known)))
        tot += hdr.matches_exact
    dt = time.perf_counter() - t0
    print("%s: %d replays, %.2f ms each, %.0f matches each" % (op, len(plans), dt / len(plans) * 1e3, tot / len(plans)))
db.close()
