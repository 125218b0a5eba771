"""Matches per query of a C2 batch (how long are the lists a counting replay would walk?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
from xapiand_amd import Database, Query, enquire
db = Database.synthetic(H.CORPUS_SEED, 10_000_000, 1_000_000)
pool = H.bench_pool("AND", 3, 1, 10_000_000, 1_000_000, n=100 + 1024, seed=0x51EED)[100:]
plans = [enquire.plan(db, Query(q["op"], q["terms"]), 0, 10) for q in pool]
m = []
for b in range(4):
    res = enquire.search_batch(db, plans[b * 256:(b + 1) * 256])
    mb = np.array([h.matches_exact & ((1 << 63) - 1) for _, h in res])
    m.append(mb)
    print("batch", b, "sum", int(mb.sum()), "max", int(mb.max()), "p99", int(np.percentile(mb, 99)), "p90", int(np.percentile(mb, 90)), "median", int(np.median(mb)), ">64k:", int((mb > 65536).sum()), ">1M:", int((mb > 1_000_000).sum()))
