import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from xapiand_amd import Database, Enquire, Query
n_docs, vocab, sb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
c = H.Corpus(n_docs, vocab)
db = Database(c.build_segment(os.path.join(tempfile.mkdtemp(), "a.seg"), stripe_bits=sb))
enq = Enquire(db)
q = dict(op=sys.argv[4], terms=sys.argv[5:], first=0, maxitems=10)
enq.set_query(Query(q["op"], q["terms"]))
got = [(i.docid, i.weight) for i in enq.get_mset(0, 10)]
hits, hdr = H.oracle_search(c, q["op"], q["terms"], 0, 10)
print("got", got[:3], "want", [(d, w) for d, w, _ in hits][:3])
