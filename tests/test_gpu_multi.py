"""The N > 1 path on the hardware that is available to the tests (ONE MI355X):

* two ranks SHARING the GPU, one shard each, through xapiand_amd.distributed.ShardedSearcher with its default
  (device) search and merge: real xgm_get_mset_batch_device on each shard, the statistics all-reduce, the all-gather
  of the 16-byte top-k records and xgm_merge_shards_device.  RCCL refuses two ranks on one device, so the
  collectives of this test run over gloo with the records staged through host memory (ShardedSearcher does that
  by itself when the backend is gloo and the tensors live on the GPU); everything else is the production path.
* one rank in a 1-rank RCCL group with force_collective: the all_gather_into_tensor + merge code of the 8-GPU run,
  executed by RCCL itself.
* xgm_search_sharded (the C-ABI protocol for shards living in one process): persistent buffers across calls, the
  RCCL exchange forced onto a 1-rank communicator, and shards smaller than first + maxitems.

Expected answers: Xapiand's protocol run on the CPU oracle (tests/helpers.py::oracle_search_sharded, pinned to the
compiled reference by tests/test_oracle_vs_reference.py).  Reference: src/database/handler.cc:1485-1549,
src/xapian/matcher/matcher.cc:653-781, src/xapian/api/enquire.cc:472-531."""
import ctypes as C
import json
import os
import socket
import subprocess
import sys

import pytest

import helpers as H

pytestmark = pytest.mark.gpu

N_DOCS, VOCAB, K = 60000, 50000, 10
WORKER = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
import helpers as H
from xapiand_amd import Database, Query
from xapiand_amd.distributed import ShardedSearcher, decode_results
backend, world, n_docs, vocab, k, out = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
else:
    dist.init_process_group("gloo", rank=rank, world_size=world)
db = Database.synthetic(H.CORPUS_SEED, n_docs, vocab, n_shards=world, shard=rank, device=0)
queries = json.load(open(out + ".queries"))
qobjs = [Query(q["op"], q["terms"], window=q.get("window", 0), n_required=q.get("n_required", 0)) for q in queries]
ss = ShardedSearcher(db, rank, world, dev, force_collective=True)
res = {}
with torch.cuda.stream(torch.cuda.Stream(device=dev)):
    descs, gs = ss.describe(qobjs, 0, k)
    for rep in range(3):                                 # the buffers and the stream binding are re-used
        hits, hdrs = ss.run_descs(descs, gs, len(qobjs), k)
        torch.cuda.current_stream().synchronize()
        got = decode_results(hits, hdrs)
        res["rep%d" % rep] = [[(d, w.hex()) for d, w, _ in rows] for rows, _ in got]
    plans = ss.prepare(qobjs, 0, k)                      # the planned-query entry point gives the same answer
    from xapiand_amd import _lib
    hits, hdrs = ss.run_batch((_lib.Query * len(plans))(*plans), len(plans), k)
    torch.cuda.current_stream().synchronize()
    res["planned"] = [[(d, w.hex()) for d, w, _ in rows] for rows, _ in decode_results(hits, hdrs)]
    res["matches"] = [h["matches"] for _, h in decode_results(hits, hdrs)]
json.dump(res, open("%s.rank%d" % (out, rank), "w"))
db.close()
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(tmp_path, backend, world, queries):
    out = str(tmp_path / ("multi_%s_%d" % (backend, world)))
    json.dump(queries, open(out + ".queries", "w"))
    script = str(tmp_path / "worker.py")
    open(script, "w").write(WORKER)
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
        procs.append(subprocess.Popen([sys.executable, script, H.ROOT, backend, str(world), str(N_DOCS), str(VOCAB), str(K), out], env=env))
    for p in procs:
        assert p.wait(600) == 0
    return [json.load(open("%s.rank%d" % (out, r))) for r in range(world)]


def mixed_queries():
    return (H.gen_term_queries("AND", 24, 3, 1, 200, maxitems=K, seed=31) + H.gen_term_queries("OR", 12, 4, 1, 2000, maxitems=K, seed=32) +
            H.gen_sided_queries("AND_MAYBE", 8, 2, 2, 1, 200, maxitems=K, seed=33))


def expected(world, queries):
    shards = [H.Corpus(N_DOCS, VOCAB, n_shards=world, shard=s) for s in range(world)]
    out = []
    for q in queries:
        if q["op"] in H.SIDED:
            # oracle_search_sharded covers the one-sided shapes; do the protocol by hand for the sided ones
            gs = dict(total_length=sum(c.v.total_length for c in shards), collection_size=sum(c.v.doccount for c in shards),
                      has_positions=True, termfreq=[sum(c.termfreq(t) for c in shards) for t in q["terms"]])
            allh = []
            for s, c in enumerate(shards):
                hits, _ = H.oracle_search(c, q["op"], q["terms"], 0, K, 0, gs, n_required=q["n_required"])
                allh += [((d - 1) * world + s + 1, w, m) for d, w, m in hits]
            allh.sort(key=lambda x: (-x[1], x[0]))
            out.append([(d, w.hex()) for d, w, _ in allh[:K]])
        else:
            out.append([(d, w.hex()) for d, w, _ in H.oracle_search_sharded(shards, q["op"], q["terms"], 0, K)])
    return out


def test_two_ranks_share_one_gpu_through_the_production_path(built, tmp_path):
    # each same-op run of queries forms its own batch (a batch is homogeneous in kernel class)
    for qs in (mixed_queries()[:24], mixed_queries()[24:36], mixed_queries()[36:]):
        want = expected(2, qs)
        res = run_ranks(tmp_path, "gloo", 2, qs)
        for r in (0, 1):
            for key in ("rep0", "rep1", "rep2", "planned"):
                assert [[tuple(x) for x in rows] for rows in res[r][key]] == want, (r, key, qs[0]["op"])
        assert res[0]["matches"] == res[1]["matches"]


def test_one_rank_rccl_group_runs_the_collective_path(built, tmp_path):
    qs = mixed_queries()[:24]
    want = expected(1, qs)
    res = run_ranks(tmp_path, "nccl", 1, qs)
    for key in ("rep0", "rep2", "planned"):
        assert [[tuple(x) for x in rows] for rows in res[0][key]] == want, key


def test_search_sharded_persistent_and_rccl_exchange(built, tmp_path):
    """xgm_search_sharded twice on the same shard list (buffers / streams re-used), then in a subprocess with the
    exchange forced through RCCL (XGM_SHARDED_RCCL=force: a 1-rank communicator on the single device)."""
    from xapiand_amd import Database, Query, _lib
    from xapiand_amd.enquire import search_sharded
    shards = [H.Corpus(20000, 20000, n_shards=4, shard=s) for s in range(4)]
    dbs = [Database(c.build_segment(str(tmp_path / ("s%d.seg" % s)))) for s, c in enumerate(shards)]
    qs = H.gen_term_queries("AND", 16, 2, 1, 64, maxitems=K, seed=5)
    want = [[(d, w) for d, w, _ in H.oracle_search_sharded(shards, "AND", q["terms"], 0, K)] for q in qs]
    for _ in range(3):
        msets = search_sharded(dbs, [Query("AND", q["terms"]) for q in qs], 0, K)
        assert [[(i.docid, i.weight) for i in m] for m in msets] == want
    info = (C.c_uint64 * 4)()
    _lib.check(_lib.lib().xgm_debug_sharded_info(dbs[0]._h, info))
    assert info[0] == 3 and info[2] == 1 and info[3] == 4 and info[1] == 0          # 4 shards on 1 device: peer-copy exchange
    for d in dbs:
        d.close()
    # one shard, exchange forced through ncclAllGather on a communicator the library owns
    code = r'''
import sys, ctypes as C
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import helpers as H
from xapiand_amd import Database, Query, _lib
from xapiand_amd.enquire import search_sharded
c = H.Corpus(20000, 20000)
db = Database(c.build_segment(sys.argv[2]))
qs = H.gen_term_queries("AND", 16, 2, 1, 64, maxitems=10, seed=5)
want = [[(d, w) for d, w, _ in H.oracle_search(c, "AND", q["terms"], 0, 10)[0]] for q in qs]
for _ in range(2):
    msets = search_sharded([db], [Query("AND", q["terms"]) for q in qs], 0, 10)
    assert [[(i.docid, i.weight) for i in m] for m in msets] == want
info = (C.c_uint64 * 4)()
_lib.check(_lib.lib().xgm_debug_sharded_info(db._h, info))
assert info[0] == 2 and info[1] == 2, list(info)       # both calls exchanged through RCCL
db.close()
print("rccl exchange ok")
'''
    env = dict(os.environ, XGM_SHARDED_RCCL="force")
    r = subprocess.run([sys.executable, "-c", code, H.ROOT, str(tmp_path / "one.seg")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl exchange ok" in r.stdout, r.stdout + r.stderr


def test_search_sharded_shards_smaller_than_k(built, tmp_path):
    """5 shards of 6 documents, maxitems = 10: the reference returns min(maxitems, total docs - first) hits
    (Enquire::merge_mset clamps against the summed doccount, enquire.cc:486-488), not max over shards."""
    from xapiand_amd import Database, Query
    from xapiand_amd.enquire import get_mset_sharded, search_sharded
    shards = []
    for s in range(5):
        post = {"a": [(d, 1 + (d + s) % 3, [1]) for d in range(1, 7)], "b": [(d, 1, [2]) for d in range(1, 7, 2)]}
        shards.append(H.ManualCorpus(post, {d: 10 + d + s for d in range(1, 7)}))
    dbs = [Database(c.build_segment(str(tmp_path / ("t%d.seg" % s)))) for s, c in enumerate(shards)]
    for first, maxitems in ((0, 10), (3, 10), (0, 40), (25, 10)):
        want = H.oracle_search_sharded(shards, "AND", ["a"], first, maxitems)
        assert len(want) == min(maxitems, max(0, 30 - first))
        m = search_sharded(dbs, [Query("a")], first, maxitems)[0]
        assert [(i.docid, i.weight) for i in m] == [(d, w) for d, w, _ in want], (first, maxitems)
        m2 = get_mset_sharded(dbs, Query("a"), first, maxitems)
        assert [(i.docid, i.weight) for i in m2] == [(d, w) for d, w, _ in want], (first, maxitems)
    for d in dbs:
        d.close()
