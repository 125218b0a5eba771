"""GPU: the device-side synthetic segment builder produces exactly the segment the host builder
produces from the oracle's inversion of the same corpus (byte-identical file), and searches on it
agree with the oracle."""
import pytest

import helpers as H
from xapiand_amd import Database, Enquire, Query

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_docs,vocab,n_shards,shard,stripe_bits,positions", [
    (5000, 3000, 1, 0, 0, True), (20000, 100000, 3, 1, 10, True), (7000, 50000, 1, 0, 8, False)])
def test_gpu_builder_equals_host_builder(built, tmp_path, n_docs, vocab, n_shards, shard, stripe_bits, positions):
    c = H.Corpus(n_docs, vocab, n_shards=n_shards, shard=shard, positions=positions)
    host_seg = c.build_segment(str(tmp_path / "host.seg"), stripe_bits=stripe_bits, revision=1)
    db = Database.synthetic(H.CORPUS_SEED, n_docs, vocab, n_shards=n_shards, shard=shard, stripe_bits=stripe_bits,
                            with_positions=positions)
    dev_seg = str(tmp_path / "dev.seg")
    db.save(dev_seg)
    a, b = open(host_seg, "rb").read(), open(dev_seg, "rb").read()
    assert len(a) == len(b)
    assert a == b
    enq = Enquire(db)
    for q in H.gen_term_queries("AND", 10, 3, 1, 200, seed=81) + H.gen_term_queries("OR", 5, 4, 1, 500, maxitems=30, seed=82):
        enq.set_query(Query(q["op"], q["terms"]))
        want, _ = H.oracle_search(c, q["op"], q["terms"], 0, q["maxitems"])
        assert [(i.docid, i.weight) for i in enq.get_mset(0, q["maxitems"])] == [(d, w) for d, w, _ in want], q
    db.close()
