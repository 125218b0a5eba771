import sys, os, tempfile
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import helpers as H
from test_gpu_parity import run_gpu, mset_pairs, oracle_pairs
from xapiand_amd.enquire import Database
N=int(sys.argv[1]); V=20000
c = H.Corpus(N, V)
d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
db = Database(c.build_segment(os.path.join(d, "c.seg")))
qs = H.gen_term_queries("AND", 40, 3, 1, 200, maxitems=10, seed=11)
bad = 0
for i, q in enumerate(qs):
    got = mset_pairs(run_gpu(db, q)); want = oracle_pairs(c, q)[0]
    if got != want:
        bad += 1
        print(i, q["terms"], "\n got ", got[:10], "\n want", want[:10])
print("bad", bad, "of", len(qs))
