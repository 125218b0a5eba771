import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from xapiand_amd import Database, Enquire, Query
n_docs, vocab, sb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
c = H.Corpus(n_docs, vocab)
tmp = tempfile.mkdtemp()
db = Database(c.build_segment(os.path.join(tmp, "a.seg"), stripe_bits=sb))
enq = Enquire(db)
qs = (H.gen_term_queries("AND", 10, 2, 1, 64, first=7, maxitems=10, seed=22) + H.gen_term_queries("AND", 10, 3, 1, 500, seed=71) +
      H.gen_term_queries("OR", 6, 4, 1, 500, maxitems=50, seed=72) + H.gen_phrase_queries(6, n_docs, vocab, seed=73))
bad = 0
for q in qs:
    enq.set_query(Query(q["op"], q["terms"], window=q.get("window", 0)))
    got = [(i.docid, i.weight) for i in enq.get_mset(q["first"], q["maxitems"])]
    hits, hdr = H.oracle_search(c, q["op"], q["terms"], q["first"], q["maxitems"], q.get("window", 0))
    want = [(d, w) for d, w, _ in hits[q["first"]:]]
    if got != want:
        bad += 1
        nd = sum(1 for a, b in zip(got, want) if a != b)
        print("MISMATCH", q["op"], q["terms"], "first", q["first"], "len", len(got), len(want), "diff", nd, "got0", got[:2], "want0", want[:2],
              "df", [c.termfreq(t) for t in q["terms"]])
print("env", os.environ.get("XGM_DEBUG_SKIP"), "bad", bad, "of", len(qs))
