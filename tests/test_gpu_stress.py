"""Stress of the last-unit merge (xapiand_amd/csrc/xgm_unit_finish.h).

The conjunction / positional / two-sided launches finish their queries themselves: every unit writes its list through, waits for
the acknowledgement (explicit `s_waitcnt vmcnt(0)`, tools/isa_contract.py checks the ISA), bumps the query's arrival counter, and the
last unit to arrive merges all lists.  Correctness across waves, workgroups and XCDs is a property of the HARDWARE's memory
ordering, which neither the parity suite's few launches nor the CPU emulation exercise.  Here: > 2 000 launches of mixed batches
whose queries have from one to several hundred units (batches of 1-4 queries are cut finest), plain / two-sided / positional, with a
copy kernel running beside them on another stream to perturb the timing, the units' scratch lists and headers WIPED before every
launch (XGM_DEBUG_POISON_SCRATCH: a list read before it landed shows as missing hits, it cannot pass for the previous batch's) —
once with the fused finish, once with the merge launch (XGM_NO_FUSED_MERGE=1), in separate processes (the switches are read once).
Every batch's hits and headers must be bit-equal."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

N_DOCS, VOCAB = 5_000_000, 500_000
ROUNDS = int(os.environ.get("XGM_STRESS_ROUNDS", "8"))


def worker():
    import ctypes as C
    import random
    import struct
    import threading

    import torch

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    from xapiand_amd import Database, Query, _lib
    from xapiand_amd.enquire import plan

    db = Database.synthetic(H.CORPUS_SEED, N_DOCS, VOCAB)
    L = _lib.lib()
    pool = (H.bench_pool("AND", 3, n_docs_global=N_DOCS, vocab=VOCAB, n=300, seed=5) + H.bench_pool("AND", 2, n_docs_global=N_DOCS, vocab=VOCAB, n=120, seed=6) +
            H.bench_pool("AND_NOT", 4, 2, n_docs_global=N_DOCS, vocab=VOCAB, n=120, seed=7) + H.bench_pool("AND_MAYBE", 4, 2, n_docs_global=N_DOCS, vocab=VOCAB, n=120, seed=8) +
            H.bench_pool("FILTER", 3, 2, n_docs_global=N_DOCS, vocab=VOCAB, n=60, seed=9) + H.gen_phrase_queries(120, N_DOCS, VOCAB, seed=10) +
            [dict(op="PHRASE", terms=["t%d" % a, "t%d" % b], first=0, maxitems=10, window=0) for a, b in ((1, 2), (2, 1), (3, 5), (1, 4), (6, 2))] +      # the heaviest phrases
            [dict(op="AND", terms=["t1", "t2", "t3"], first=0, maxitems=50, window=0), dict(op="AND", terms=["t2", "t9"], first=3, maxitems=100, window=0)])
    plans = [plan(db, Query(q["op"], q["terms"], window=q.get("window", 0), n_required=q.get("n_required", 0)), q["first"], q["maxitems"]) for q in pool]
    rng = random.Random(77)
    batches = []
    for _ in range(ROUNDS):
        for nq in (1, 1, 2, 3, 4, 5, 8, 16, 64, 256):
            for _rep in range(35 if nq <= 8 else 6):
                batches.append([rng.randrange(len(plans)) for _ in range(nq)])
    # a copy kernel beside the searches: 256 MB back and forth on its own stream until the searches are done
    stop = threading.Event()

    def perturb():
        s = torch.cuda.Stream()
        a = torch.empty(64 << 20, dtype=torch.int32, device="cuda")
        b = torch.empty_like(a)
        with torch.cuda.stream(s):
            while not stop.is_set():
                for _ in range(8):
                    b.copy_(a, non_blocking=True)
                    a.copy_(b, non_blocking=True)
                s.synchronize()
    th = threading.Thread(target=perturb)
    th.start()
    digest = hashlib.sha256()
    per_batch = []
    n_launches = 0
    max_units = 0
    try:
        for ids in batches:
            nq = len(ids)
            ks = max(plans[i].first + plans[i].maxitems for i in ids)
            qs = (_lib.Query * nq)(*[plans[i] for i in ids])
            hits = (_lib.Hit * (nq * ks))()
            hdrs = (_lib.ResultHdr * nq)()
            _lib.check(L.xgm_search_batch(db._h, qs, nq, ks, hits, hdrs))
            h = hashlib.sha256()
            for i in range(nq):
                hd = hdrs[i]
                # a positional query that prunes by weight reports a LOWER BOUND of its match count (include/xgm.h): how many candidates
                # had their positions tested depends on when the query-wide threshold rose — timing, not semantics; the hits do not
                # ... and whether anything was dropped at all (the flag itself) is timing too: positional queries are compared by their hits
                m = hd.matches_exact if pool[ids[i]]["op"] not in ("PHRASE", "NEAR") else 0
                h.update(struct.pack("<IIQdd", hd.n_hits, hd.max_weight_subqs_matched, m, hd.max_attained, hd.max_possible))
                n = hd.n_hits
                h.update(bytes(memoryview(hits)[i * ks:i * ks + n]))
            per_batch.append(h.hexdigest()[:16] + ":" + ",".join(sorted({pool[i]["op"] for i in ids})))
            digest.update(h.digest())
            n_launches += 1
    finally:
        stop.set()
        th.join()
    # how finely single queries are cut (diagnostics of the planner; the same in both processes)
    kern = C.create_string_buffer(64)
    units = (C.c_uint32 * (4 * 65536))()
    for i in (len(plans) - 7, len(plans) - 2, 0):
        n = L.xgm_debug_plan_batch(db._h, C.byref(plans[i]), 1, kern, units, 65536)
        max_units = max(max_units, int(n))
    db.close()
    print(json.dumps(dict(digest=digest.hexdigest(), launches=n_launches, per_batch=per_batch, max_units_single_query=max_units)))


def run_worker(extra_env):
    env = dict(os.environ, XGM_DEBUG_POISON_SCRATCH="1")
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads(line[-1])


@pytest.mark.gpu
def test_last_unit_merge_equals_merge_launch_under_stress(built):
    fused = run_worker({})
    launch = run_worker({"XGM_NO_FUSED_MERGE": "1"})
    assert fused["launches"] == launch["launches"] >= 2000, (fused["launches"], launch["launches"])
    assert fused["max_units_single_query"] >= 200, fused["max_units_single_query"]
    bad = [i for i, (a, b) in enumerate(zip(fused["per_batch"], launch["per_batch"])) if a != b]
    assert not bad, "%d of %d batches differ between the fused finish and the merge launch, first: %s" % (
        len(bad), fused["launches"], [(i, fused["per_batch"][i], launch["per_batch"][i]) for i in bad[:6]])
    assert fused["digest"] == launch["digest"]
    # ... and the fused finish agrees with itself run to run (the order in which units arrive differs every time)
    again = run_worker({})
    assert again["digest"] == fused["digest"]


if __name__ == "__main__" and "--worker" in sys.argv:
    worker()
