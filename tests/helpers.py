"""Test-side helpers: ctypes binding of the CPU oracle (oracle/libxgm_oracle.so), the runner of the
real-reference binary (oracle/_ref/xapian_ref), seeded query generators and the glue that feeds the
oracle's raw postings to the product's segment builder.  Only tests/, smoke() and bench.py's
cpu_baseline leg import this module."""
import ctypes as C
import os
import random
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(ROOT, "oracle", "libxgm_oracle.so")
XAPIAN_REF = os.path.join(ROOT, "oracle", "_ref", "xapian_ref")

CORPUS_SEED = 0x5EED0001
QUERY_SEED = 0x5EED0002
OPS = {"AND": 1, "OR": 2, "PHRASE": 3, "AND_NOT": 4, "AND_MAYBE": 5, "FILTER": 6, "NEAR": 7}   # NEAR: oracle only so far
SIDED = ("AND_NOT", "AND_MAYBE", "FILTER")   # left = AND of the first n_required terms, right = the others


MATCHES_LOWER_BOUND = 1 << 63      # include/xgm.h: XGM_MATCHES_LOWER_BOUND
EXACT_COUNT = 0x7FFFFFFF           # a check_at_least that asks for the exact match count of a positional query


def check_matches(got_raw, want, n_hits, what=None):
    """xgm_result_hdr.matches_exact against the oracle's count: exact — or, flagged (a positional query answered with
    check_at_least inside the page: candidates that cannot rank are dropped before their positions are tested), a lower
    bound that still covers the hits returned."""
    if got_raw & MATCHES_LOWER_BOUND:
        cnt = got_raw & ~MATCHES_LOWER_BOUND
        assert n_hits <= cnt <= want, (what, cnt, want, n_hits)
    else:
        assert got_raw == want, (what, got_raw, want)


class CorpusView(C.Structure):
    _fields_ = [("n_terms", C.c_uint32), ("lastdocid", C.c_uint32), ("doccount", C.c_uint32),
                ("has_positions", C.c_uint32), ("total_length", C.c_uint64), ("n_postings", C.c_uint64),
                ("n_positions", C.c_uint64),
                ("doclen", C.POINTER(C.c_uint32)), ("terms", C.POINTER(C.c_char_p)),
                ("term_len", C.POINTER(C.c_uint32)), ("df", C.POINTER(C.c_uint32)),
                ("did", C.POINTER(C.c_uint32)), ("wdf", C.POINTER(C.c_uint32)),
                ("pos_off", C.POINTER(C.c_uint64)), ("pos", C.POINTER(C.c_uint32))]


class OHit(C.Structure):
    _fields_ = [("docid", C.c_uint32), ("subqs", C.c_uint32), ("weight", C.c_double)]


class OHdr(C.Structure):
    _fields_ = [("n_hits", C.c_uint32), ("max_subqs", C.c_uint32), ("matches", C.c_uint64),
                ("max_attained", C.c_double), ("max_possible", C.c_double)]


_olib = None


def olib():
    global _olib
    if _olib is None:
        l = C.CDLL(ORACLE_LIB)
        l.xgo_corpus_build.restype = C.c_void_p
        l.xgo_corpus_build.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        l.xgo_corpus_build_terms.restype = C.c_void_p
        l.xgo_corpus_build_terms.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32),
                                             C.c_uint32, C.c_uint32]
        l.xgo_corpus_free.argtypes = [C.c_void_p]
        l.xgo_corpus_get.argtypes = [C.c_void_p, C.POINTER(CorpusView)]
        l.xgo_index_from_raw.restype = C.c_void_p
        l.xgo_index_from_raw.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32),
                                         C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                         C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint32)]
        l.xgo_index_free.argtypes = [C.c_void_p]
        l.xgo_index_termfreq.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
        l.xgo_index_warm.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
        l.xgo_search.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_uint32,
                                 C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32,
                                 C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(OHit), C.POINTER(OHdr)]
        _olib = l
    return _olib


class Corpus:
    """Deterministic synthetic corpus shard (tools/xgm_corpus.h) inverted to raw postings by the oracle."""

    def __init__(self, n_docs_global, vocab, seed=CORPUS_SEED, len_lo=50, len_hi=150, n_shards=1, shard=0, positions=True):
        self.params = dict(seed=seed, n_docs_global=n_docs_global, vocab=vocab, len_lo=len_lo, len_hi=len_hi,
                           n_shards=n_shards, shard=shard)
        self._h = olib().xgo_corpus_build(seed, n_docs_global, vocab, len_lo, len_hi, n_shards, shard, 1 if positions else 0)
        self.v = CorpusView()
        olib().xgo_corpus_get(self._h, C.byref(self.v))
        self._oidx = None

    def close(self):
        if self._oidx:
            olib().xgo_index_free(self._oidx)
            self._oidx = None
        if self._h:
            olib().xgo_corpus_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def terms(self):
        return [C.string_at(self.v.terms[i], self.v.term_len[i]) for i in range(self.v.n_terms)]

    def df_array(self):
        return np.ctypeslib.as_array(self.v.df, shape=(self.v.n_terms,))

    def term_postings(self, idx):
        df = self.df_array()
        start = int(df[:idx].sum())
        n = int(df[idx])
        did = np.ctypeslib.as_array(self.v.did, shape=(self.v.n_postings,))[start:start + n]
        wdf = np.ctypeslib.as_array(self.v.wdf, shape=(self.v.n_postings,))[start:start + n]
        return did, wdf

    def raw_postings(self, revision=1):
        """Fill the product's xgm_raw_postings from the oracle's arrays (borrowed pointers)."""
        from xapiand_amd import _lib
        r = _lib.RawPostings()
        v = self.v
        r.n_terms, r.lastdocid, r.doccount, r.has_positions = v.n_terms, v.lastdocid, v.doccount, v.has_positions
        r.total_length, r.n_postings, r.n_positions, r.revision = v.total_length, v.n_postings, v.n_positions, revision
        r.doclen, r.terms, r.term_len, r.df, r.did, r.wdf = v.doclen, v.terms, v.term_len, v.df, v.did, v.wdf
        r.pos_off, r.pos = v.pos_off, v.pos
        return r

    def build_segment(self, path, stripe_bits=0, revision=1):
        from xapiand_amd import _lib
        r = self.raw_postings(revision)
        _lib.check(_lib.lib().xgm_segment_build(C.byref(r), stripe_bits, path.encode()))
        return path

    def oracle_index(self):
        if self._oidx is None:
            v = self.v
            self._oidx = olib().xgo_index_from_raw(v.n_terms, v.lastdocid, v.doccount, v.total_length, v.doclen, v.terms,
                                                   v.term_len, v.df, v.did, v.wdf, v.pos_off, v.pos)
        return self._oidx

    def termfreq(self, term):
        t = term if isinstance(term, bytes) else term.encode()
        return olib().xgo_index_termfreq(self.oracle_index(), t, len(t))


def oracle_search(corpus, op, terms, first, maxitems, window=0, global_stats=None, reference_select_bug=False, n_required=0):
    """Run the CPU oracle.  Returns (list of (docid, weight, subqs), hdr)."""
    n = len(terms)
    opcode = OPS[op] | ((n_required or 1) << 8 if op in SIDED else 0)
    tb = [t if isinstance(t, bytes) else t.encode() for t in terms]
    arr = (C.c_char_p * n)(*tb)
    lens = (C.c_uint32 * n)(*[len(t) for t in tb])
    cap = max(1, first + maxitems)
    hits = (OHit * cap)()
    hdr = OHdr()
    bug = 1 if reference_select_bug else 0
    if global_stats is None:
        rc = olib().xgo_search(corpus.oracle_index(), opcode, n, arr, lens, window, first, maxitems, 0, 0, 0, 0, None, bug, hits, C.byref(hdr))
    else:
        tf = (C.c_uint32 * n)(*global_stats["termfreq"])
        rc = olib().xgo_search(corpus.oracle_index(), opcode, n, arr, lens, window, first, maxitems, 1,
                               global_stats["total_length"], global_stats["collection_size"],
                               1 if global_stats["has_positions"] else 0, tf, bug, hits, C.byref(hdr))
    assert rc == 0
    return [(hits[i].docid, hits[i].weight, hits[i].subqs) for i in range(hdr.n_hits)], hdr


SORT_MODES = {"V": 1, "VR": 2, "RV": 3}      # Enquire::set_sort_by_value / _value_then_relevance / _relevance_then_value


def oracle_search_sorted(corpus, op, terms, first, maxitems, mode, slot, reverse, n_required=0, collapse=None, global_stats=None):
    """The oracle with a value sort (mode "V" / "VR" / "RV"; None = relevance) and / or a collapse (slot, collapse_max) in force
    (widening row (f).3; corpus value slots: tools/xgm_corpus.h).  Returns (list of (docid, weight, subqs, sort_key bytes),
    hdr) — with collapse: (docid, weight, subqs, sort_key, collapse_key, collapse_count) and hdr.collapsed_lower_bound."""
    ol = olib()
    ol.xgo_index_set_synthetic_values.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32]
    ol.xgo_search_sorted_g.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(OHit), C.POINTER(OHdr), C.c_char_p, C.c_uint32,
                                       C.c_uint32, C.c_uint32, C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                       C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    if not getattr(corpus, "_values_set", False):
        ol.xgo_index_set_synthetic_values(corpus.oracle_index(), corpus.params["seed"], corpus.params["n_shards"], corpus.params["shard"])
        corpus._values_set = True
    n = len(terms)
    opcode = OPS[op] | ((n_required or 1) << 8 if op in SIDED else 0)
    tb = [t if isinstance(t, bytes) else t.encode() for t in terms]
    arr = (C.c_char_p * n)(*tb)
    lens = (C.c_uint32 * n)(*[len(t) for t in tb])
    cap = max(1, first + maxitems)
    hits = (OHit * cap)()
    hdr = OHdr()
    keys, ckeys = C.create_string_buffer(cap * 8), C.create_string_buffer(cap * 8)
    ccounts = (C.c_uint32 * cap)()
    clb = C.c_uint64()
    cslot, cmax = collapse if collapse else (0, 0)
    gs = global_stats
    tf = (C.c_uint32 * n)(*gs["termfreq"]) if gs else None
    rc = ol.xgo_search_sorted_g(corpus.oracle_index(), opcode, n, arr, lens, 0, first, maxitems, SORT_MODES[mode] if mode else 0, slot, 1 if reverse else 0,
                                hits, C.byref(hdr), keys, 8, cslot, cmax, ckeys, ccounts, C.byref(clb),
                                1 if gs else 0, gs["total_length"] if gs else 0, gs["collection_size"] if gs else 0,
                                (1 if gs["has_positions"] else 0) if gs else 0, tf)
    assert rc == 0
    raw, craw = keys.raw, ckeys.raw
    hdr.collapsed_lower_bound = clb.value
    if collapse:
        return [(hits[i].docid, hits[i].weight, hits[i].subqs, raw[8 * i:8 * i + 8].rstrip(b"\0"), craw[8 * i:8 * i + 8].rstrip(b"\0"), ccounts[i]) for i in range(hdr.n_hits)], hdr
    return [(hits[i].docid, hits[i].weight, hits[i].subqs, raw[8 * i:8 * i + 8].rstrip(b"\0")) for i in range(hdr.n_hits)], hdr


def oracle_search_sharded(corpora, op, terms, first, maxitems, window=0):
    """Xapiand's per-shard protocol on the oracle: merged stats, per-shard top first+maxitems, unshard, merge."""
    gs = dict(total_length=sum(c.v.total_length for c in corpora), collection_size=sum(c.v.doccount for c in corpora),
              has_positions=any(c.v.has_positions for c in corpora),
              termfreq=[sum(c.termfreq(t) for c in corpora) for t in terms])
    n = len(corpora)
    allhits = []
    for s, c in enumerate(corpora):
        hits, _ = oracle_search(c, op, terms, 0, first + maxitems, window, gs)
        allhits += [((d - 1) * n + s + 1, w, m) for d, w, m in hits]
    allhits.sort(key=lambda x: (-x[1], x[0]))
    return allhits[first:first + maxitems]


def oracle_spy(corpus, op, terms, slot, window=0, n_required=0):
    """A ValueCountMatchSpy on `slot` over every document the query matches (corpus value slots: tools/xgm_corpus.h).
    Returns (documents seen, [(value, count)] in value order)."""
    ol = olib()
    ol.xgo_index_set_synthetic_values.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32]
    ol.xgo_search_spy.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.c_uint32, C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    if not getattr(corpus, "_values_set", False):
        ol.xgo_index_set_synthetic_values(corpus.oracle_index(), corpus.params["seed"], corpus.params["n_shards"], corpus.params["shard"])
        corpus._values_set = True
    n = len(terms)
    opcode = OPS[op] | ((n_required or 1) << 8 if op in SIDED else 0)
    tb = [t if isinstance(t, bytes) else t.encode() for t in terms]
    arr = (C.c_char_p * n)(*tb)
    lens = (C.c_uint32 * n)(*[len(t) for t in tb])
    cap = 1 << 20
    values, counts = C.create_string_buffer(cap * 8), (C.c_uint32 * cap)()
    nv, total = C.c_uint32(), C.c_uint64()
    assert ol.xgo_search_spy(corpus.oracle_index(), opcode, n, arr, lens, window, slot, cap, 8, values, counts, C.byref(nv), C.byref(total)) == 0
    raw = values.raw
    return total.value, [(raw[8 * i:8 * i + 8].rstrip(b"\0"), counts[i]) for i in range(nv.value)]


def oracle_search_sharded_sorted(corpora, op, terms, first, maxitems, mode, slot, reverse):
    """oracle_search_sharded under a value sort: every shard's first + maxitems best under the comparison, merged under the same
    comparison over global docids (matcher/msetcmp.cc:64-101).  Returns [(global docid, weight, subqs, sort key)]."""
    import functools
    gs = dict(total_length=sum(c.v.total_length for c in corpora), collection_size=sum(c.v.doccount for c in corpora),
              has_positions=any(c.v.has_positions for c in corpora),
              termfreq=[sum(c.termfreq(t) for c in corpora) for t in terms])
    n = len(corpora)
    allhits = []
    for s, c in enumerate(corpora):
        hits, _ = oracle_search_sorted(c, op, terms, 0, first + maxitems, mode, slot, reverse, global_stats=gs)
        allhits += [((d - 1) * n + s + 1, w, m, k) for d, w, m, k in hits]

    def cmp(a, b):
        if mode == "RV" and a[1] != b[1]:
            return -1 if a[1] > b[1] else 1
        if a[3] != b[3]:
            return (-1 if a[3] > b[3] else 1) if reverse else (-1 if a[3] < b[3] else 1)
        if mode == "VR" and a[1] != b[1]:
            return -1 if a[1] > b[1] else 1
        return -1 if a[0] < b[0] else (1 if a[0] > b[0] else 0)
    allhits.sort(key=functools.cmp_to_key(cmp))
    return allhits[first:first + maxitems]


# ---- the real reference -------------------------------------------------------------------------

def have_xapian_ref():
    return os.path.exists(XAPIAN_REF)


def xapian_ref(*args):
    return subprocess.run([XAPIAN_REF] + [str(a) for a in args], check=True, capture_output=True, text=True).stdout


def write_queries(path, queries):
    with open(path, "w") as f:
        for q in queries:
            op = q["op"] + (":%d" % (q.get("n_required") or 1) if q["op"] in SIDED else "")
            pre = ""
            if q.get("sort"):
                pre += "SORT=%s:%d:%d " % (q["sort"][0], q["sort"][1], 1 if q["sort"][2] else 0)
            if q.get("collapse"):
                pre += "COLLAPSE=%d:%d " % tuple(q["collapse"])
            if q.get("check_at_least"):
                pre += "CAL=%d " % q["check_at_least"]
            if q.get("spy") is not None:
                if q.get("spy_aggregation"):
                    pre += "SPYA=%d:%d " % (q["spy"], q.get("agg_kind", 0))          # Xapiand's own AggregationMatchSpy; kind: oracle/ref_build/xapiand_classes.cc aggs_conf
                else:
                    pre += ("SPYC=%d " if q.get("spy_custom") else "SPY=%d ") % q["spy"]
            if q.get("cutoff"):
                pre += "CUT=%d:%r " % (q["cutoff"][0], float(q["cutoff"][1]))
            f.write("%s%s %d %d %d %s\n" % (pre, op, q["first"], q["maxitems"], q.get("window", 0), " ".join(q["terms"])))


def parse_ref_output(path):
    """Parse xapian_ref query output → list of dicts with exact (hex) weights."""
    out = []
    with open(path) as f:
        for line in f:
            p = line.split()
            if p[0] == "Q":
                out.append(dict(n=int(p[2]), lb=int(p[3]), est=int(p[4]), ub=int(p[5]), max_possible=float.fromhex(p[6]),
                                max_attained=float.fromhex(p[7]), hits=[]))
            elif p[0] == "H":
                out[-1]["hits"].append((int(p[2]), float.fromhex(p[3]), int(p[4])))
            elif p[0] == "X":       # sorted / collapsed searches: sort key, collapse key (hex, "-" = empty), collapse count
                out[-1].setdefault("extra", []).append((b"" if p[2] == "-" else bytes.fromhex(p[2]), b"" if p[3] == "-" else bytes.fromhex(p[3]), int(p[4])))
            elif p[0] == "U":
                out[-1]["uncollapsed"] = (int(p[1]), int(p[2]), int(p[3]))
            elif p[0] == "S":       # a ValueCountMatchSpy: documents it saw, then one V line per distinct value
                out[-1]["spy_total"], out[-1]["spy"] = int(p[1]), []
            elif p[0] == "V":
                out[-1]["spy"].append((bytes.fromhex(p[1]), int(p[2])))
    return out


# ---- seeded query generators ---------------------------------------------------

def log_uniform_rank(rng, lo, hi):
    import math
    return int(round(math.exp(rng.uniform(math.log(lo), math.log(hi)))))


def gen_term_queries(op, n_queries, n_terms, rank_lo, rank_hi, first=0, maxitems=10, seed=QUERY_SEED):
    rng = random.Random(seed)
    qs = []
    for _ in range(n_queries):
        ranks = set()
        while len(ranks) < n_terms:
            ranks.add(max(1, log_uniform_rank(rng, rank_lo, rank_hi)))
        ranks = list(ranks)
        rng.shuffle(ranks)
        qs.append(dict(op=op, terms=["t%d" % r for r in ranks], first=first, maxitems=maxitems, window=0))
    return qs


def gen_sided_queries(op, n_queries, n_required, n_other, rank_lo, rank_hi, other_lo=None, other_hi=None, first=0, maxitems=10, seed=QUERY_SEED):
    """AND_NOT / AND_MAYBE / FILTER: the first n_required terms form the left-hand AND, n_other terms the
    right-hand side (ranks from their own range when given)."""
    rng = random.Random(seed)
    qs = []
    for _ in range(n_queries):
        ranks = []
        while len(ranks) < n_required:
            r = max(1, log_uniform_rank(rng, rank_lo, rank_hi))
            if r not in ranks:
                ranks.append(r)
        while len(ranks) < n_required + n_other:
            r = max(1, log_uniform_rank(rng, other_lo or rank_lo, other_hi or rank_hi))
            if r not in ranks:
                ranks.append(r)
        qs.append(dict(op=op, n_required=n_required, terms=["t%d" % r for r in ranks], first=first, maxitems=maxitems, window=0))
    return qs


def hash64(seed, a, b):
    """tools/xgm_corpus.h xgm_hash3 in Python ints."""
    M = (1 << 64) - 1

    def mix(z):
        z = (z + 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)
    return mix(mix(seed ^ ((a * 0xD6E8FEB86659FD93) & M)) ^ ((b * 0xA0761D6478BD642F) & M))


_thr_cache = {}


def zipf_thresholds(vocab):
    if vocab not in _thr_cache:
        inv = 1.0 / np.arange(1, vocab + 1, dtype=np.float64)
        # sequential summation exactly like the C code (np.cumsum is sequential for float64)
        h = np.cumsum(inv)
        hv = 0.0
        for x in inv:       # same order of additions as xgm_zipf_thresholds' first loop
            hv += x
        f = h / hv
        hi = f * 4294967296.0
        hi_i = np.floor(hi)
        lo = (hi - hi_i) * 4294967296.0
        thr = (hi_i.astype(np.uint64) << np.uint64(32)) | np.floor(lo).astype(np.uint64)
        thr[-1] = np.uint64(0xFFFFFFFFFFFFFFFF)
        thr[f >= 1.0] = np.uint64(0xFFFFFFFFFFFFFFFF)
        _thr_cache[vocab] = thr
    return _thr_cache[vocab]


def doc_tokens(seed, vocab, len_lo, len_hi, g):
    """Tokens (ranks) of global doc g — Python restatement of the corpus for phrase-query sampling."""
    thr = zipf_thresholds(vocab)
    n = len_lo + hash64(seed, g, 0xFFFFFFFF) % (len_hi - len_lo + 1)
    toks = []
    for pos in range(1, n + 1):
        u = np.uint64(hash64(seed, g, pos))
        toks.append(int(np.searchsorted(thr, u, side="right")) + 1 if u >= thr[0] else 1)
    return toks


def gen_phrase_queries(n_queries, n_docs_global, vocab, len_lo=50, len_hi=150, corpus_seed=CORPUS_SEED, seed=QUERY_SEED,
                       maxitems=10, window_extra=0, lengths=(2, 3), op="PHRASE"):
    """2-3-grams that actually occur in a random document (so results are non-empty); n-grams with a
    repeated term are skipped (the device path declines them, like any other unsupported shape)."""
    rng = random.Random(seed)
    qs = []
    while len(qs) < n_queries:
        g = rng.randint(1, n_docs_global)
        toks = doc_tokens(corpus_seed, vocab, len_lo, len_hi, g)
        n = rng.choice(list(lengths))
        i = rng.randint(0, len(toks) - n)
        gram = toks[i:i + n]
        if len(set(gram)) != n:
            continue
        qs.append(dict(op=op, terms=["t%d" % r for r in gram], first=0, maxitems=maxitems,
                       window=(n + window_extra) if window_extra else 0))
    return qs


def bench_pool(op, n_terms=3, n_required=1, n_docs_global=10_000_000, vocab=1_000_000, n=1100, seed=QUERY_SEED, maxitems=10):
    """The query pool of bench.py (ranks log-uniform in [8, 4096], no repetition inside a query; PHRASE:
    2-3-grams that occur in a random document) as query dicts.  bench.py warms up on the first 100 and times the
    rest; the config-scale parity tests check the first queries of the timed part."""
    import math
    if op == "PHRASE":
        return gen_phrase_queries(n, n_docs_global, vocab, seed=seed, maxitems=maxitems)
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        ranks = set()
        while len(ranks) < n_terms:
            ranks.add(max(1, int(round(math.exp(rng.uniform(math.log(8), math.log(4096)))))))
        ranks = list(ranks)
        rng.shuffle(ranks)
        out.append(dict(op=op, terms=["t%d" % r for r in ranks], first=0, maxitems=maxitems, window=0,
                        n_required=n_required if op in SIDED else 0))
    return out


class TermsCorpus(Corpus):
    """The synthetic corpus inverted on the HOST (oracle/xgm_oracle.cc::corpus_build_terms, tools/xgm_corpus.h) for a handful of terms only: every document is
    generated — lengths and statistics are the whole shard's —, only the postings of `terms` ("t<rank>") are kept.  What feeds the oracle at a
    configuration's full size WITHOUT reading anything back from the device (DeviceOracle does); same interface as Corpus for the oracle functions."""

    def __init__(self, n_docs_global, vocab, terms, seed=CORPUS_SEED, len_lo=50, len_hi=150, n_shards=1, shard=0, positions=False, n_threads=None):
        ranks = sorted({int((t.decode() if isinstance(t, bytes) else t)[1:]) for t in terms})
        arr = (C.c_uint32 * len(ranks))(*ranks)
        self.params = dict(seed=seed, n_docs_global=n_docs_global, vocab=vocab, len_lo=len_lo, len_hi=len_hi, n_shards=n_shards, shard=shard)
        self._h = olib().xgo_corpus_build_terms(seed, n_docs_global, vocab, len_lo, len_hi, n_shards, shard, 1 if positions else 0, arr, len(ranks),
                                                n_threads or max(1, os.cpu_count() or 1))
        self.v = CorpusView()
        olib().xgo_corpus_get(self._h, C.byref(self.v))
        self._oidx = None

    def warm(self):
        for i in range(self.v.n_terms):
            olib().xgo_index_warm(self.oracle_index(), self.v.terms[i], self.v.term_len[i])


class ManualCorpus(Corpus):
    """Hand-made postings: {term: [(docid, wdf, [positions...]), ...]} plus doc lengths.  Same
    interface as Corpus so it can feed both the oracle and the segment builder."""

    def __init__(self, postings, doclen, positions=True):
        terms = sorted((t if isinstance(t, bytes) else t.encode()) for t in postings)
        by = {(t if isinstance(t, bytes) else t.encode()): v for t, v in postings.items()}
        lastdocid = max(doclen) if doclen else 0
        self._doclen = np.zeros(lastdocid + 1, dtype=np.uint32)
        for d, l in doclen.items():
            self._doclen[d] = l
        df, did, wdf, pos_off, pos = [], [], [], [0], []
        for t in terms:
            plist = sorted(by[t])
            df.append(len(plist))
            for p in plist:
                did.append(p[0]); wdf.append(p[1])
                pp = list(p[2]) if len(p) > 2 and p[2] is not None else []
                pos += pp
                pos_off.append(len(pos))
        self._df = np.array(df, dtype=np.uint32)
        self._did = np.array(did, dtype=np.uint32)
        self._wdf = np.array(wdf, dtype=np.uint32)
        self._pos_off = np.array(pos_off, dtype=np.uint64)
        self._pos = np.array(pos if pos else [0], dtype=np.uint32)
        self._term_len = np.array([len(t) for t in terms], dtype=np.uint32)
        self._terms = (C.c_char_p * len(terms))(*terms)
        self._term_bytes = terms
        v = CorpusView()
        v.n_terms, v.lastdocid = len(terms), lastdocid
        v.doccount = int((self._doclen > 0).sum())
        v.has_positions = 1 if positions else 0
        v.total_length = int(self._doclen.sum())
        v.n_postings, v.n_positions = len(did), len(pos)
        u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        v.doclen = self._doclen.ctypes.data_as(u32p)
        v.terms = C.cast(self._terms, C.POINTER(C.c_char_p))
        v.term_len = self._term_len.ctypes.data_as(u32p)
        v.df = self._df.ctypes.data_as(u32p)
        v.did = self._did.ctypes.data_as(u32p)
        v.wdf = self._wdf.ctypes.data_as(u32p)
        if positions:
            v.pos_off = self._pos_off.ctypes.data_as(u64p)
            v.pos = self._pos.ctypes.data_as(u32p)
        self.v = v
        self._h = None
        self._oidx = None
        self.params = None

    def terms(self):
        return list(self._term_bytes)


def coloc_postings(seed=0xC010C, n_docs=600):
    """Documents whose indexer puts SEVERAL TERMS AT ONE POSITION (a word's prefixed and unprefixed forms, synonyms): what engages the
    duplicate-position step of NearPostList::test_doc (nearpostlist.cc:106-140).  Returns (postings {term: [(did, wdf, [pos...])]}, doclen
    {did: length}) for ManualCorpus; write_postings_file() spells the same documents out for `xapian_ref build_postings`.  Five query terms
    ca..ce — ca / cb mostly TOGETHER on a position, cc often with one of them, cd / ce on positions of their own — among fillers."""
    import random
    rng = random.Random(seed)
    post, doclen = {}, {}
    for d in range(1, n_docs + 1):
        at = {}                                      # position -> set of terms
        n_pos = rng.randrange(12, 40)
        for p in range(1, n_pos + 1):
            r = rng.random()
            here = set()
            if r < 0.10: here |= {"ca", "cb"}
            elif r < 0.16: here.add("ca")
            elif r < 0.22: here.add("cb")
            if rng.random() < 0.12: here.add("cc")
            if rng.random() < 0.07: here.add(rng.choice(["cd", "ce"]))
            if rng.random() < 0.03: here |= {"cd", "ce"}
            if not here or rng.random() < 0.5: here.add("f%d" % rng.randrange(30))
            at[p] = here
        by = {}
        for p in sorted(at):
            for t in at[p]:
                by.setdefault(t, []).append(p)
        for t, pp in by.items():
            post.setdefault(t, []).append((d, len(pp), pp))
        doclen[d] = sum(len(pp) for pp in by.values())
    return post, doclen


def write_postings_file(path, post, doclen):
    docs = {}
    for t, pl in post.items():
        for d, _, pp in pl:
            docs.setdefault(d, []).extend("%s:%d" % (t, p) for p in pp)
    with open(path, "w") as f:
        for d in range(1, max(doclen) + 1):
            f.write(" ".join(sorted(docs.get(d, []))) + "\n")


def coloc_near_queries():
    qs = []
    for terms in (["ca", "cb"], ["cb", "ca"], ["ca", "cc"], ["ca", "cb", "cc"], ["cc", "cb", "ca"], ["cd", "ce"], ["ca", "cd", "ce"], ["ca", "cb", "cc", "cd"],
                  ["ce", "cc", "cb", "ca", "cd"], ["f1", "ca"], ["cb", "f2", "cc"]):
        for win in (len(terms), len(terms) + 1, len(terms) + 3, 12):
            if win in (len(terms), len(terms) + 3) and len(terms) <= 3:
                qs.append(dict(op="NEAR", terms=terms, first=0, maxitems=2000, window=win))      # the whole match
            qs.append(dict(op="NEAR", terms=terms, first=0, maxitems=10, window=win))
    return qs


# ---- oracle over postings copied back from a device-resident index (config-scale parity, bench cpu_baseline) ----

class DeviceOracle:
    """A CPU-oracle index over the postings of `terms` as the DEVICE holds them (decoded by K1 on the GPU and copied
    back, with the positions when `positions`), so that the oracle answers on exactly the index the HIP path
    searches — at any size, without re-inverting the corpus on the host.  Same interface as Corpus for
    oracle_search / oracle_search_batch (oracle_index())."""

    def __init__(self, db, terms, positions=False):
        from xapiand_amd import _lib
        L = _lib.lib()
        info = db.info()
        terms = sorted({t if isinstance(t, bytes) else t.encode() for t in terms})
        u32p = C.POINTER(C.c_uint32)
        self.doclen = np.zeros(info.lastdocid + 1, dtype=np.uint32)
        _lib.check(min(0, L.xgm_debug_read_doclen(db._h, self.doclen.ctypes.data_as(u32p), self.doclen.size)))
        dids, wdfs, poss, keep = [], [], [], []
        for t in terms:
            tid, tf = C.c_uint32(), C.c_uint32()
            _lib.check(L.xgm_lookup_term(db._h, t, len(t), C.byref(tid), C.byref(tf), None, None))
            if tf.value == 0:
                continue
            d = np.zeros(tf.value, dtype=np.uint32)
            w = np.zeros(tf.value, dtype=np.uint32)
            n = L.xgm_debug_decode_term_device(db._h, tid.value, d.ctypes.data_as(u32p), w.ctypes.data_as(u32p), tf.value)
            assert n == tf.value, L.xgm_last_error()
            if positions:
                p = np.zeros(int(w.sum(dtype=np.uint64)), dtype=np.uint32)
                n = L.xgm_debug_read_positions(db._h, tid.value, p.ctypes.data_as(u32p), p.size)
                assert n == p.size, L.xgm_last_error()
                poss.append(p)
            keep.append(t); dids.append(d); wdfs.append(w)
        self.terms = keep
        self.did = np.concatenate(dids) if keep else np.zeros(0, dtype=np.uint32)
        self.wdf = np.concatenate(wdfs) if keep else np.zeros(0, dtype=np.uint32)
        self.df = np.array([d.size for d in dids], dtype=np.uint32)
        self.tlen = np.array([len(t) for t in keep], dtype=np.uint32)
        self._tarr = (C.c_char_p * len(keep))(*keep)
        pos_off_p, pos_p = None, None
        if positions:
            self.pos = np.concatenate(poss) if keep else np.zeros(1, dtype=np.uint32)
            self.pos_off = np.zeros(self.wdf.size + 1, dtype=np.uint64)
            np.cumsum(self.wdf, dtype=np.uint64, out=self.pos_off[1:])
            pos_off_p, pos_p = self.pos_off.ctypes.data_as(C.POINTER(C.c_uint64)), self.pos.ctypes.data_as(u32p)
        self._oidx = olib().xgo_index_from_raw(len(keep), info.lastdocid, info.doccount, info.total_length, self.doclen.ctypes.data_as(u32p),
                                               C.cast(self._tarr, C.POINTER(C.c_char_p)), self.tlen.ctypes.data_as(u32p),
                                               self.df.ctypes.data_as(u32p), self.did.ctypes.data_as(u32p), self.wdf.ctypes.data_as(u32p),
                                               pos_off_p, pos_p)

    def oracle_index(self):
        return self._oidx

    def warm(self):
        for t in self.terms:
            olib().xgo_index_warm(self._oidx, t, len(t))          # glass chunk encoding is index-build work

    def close(self):
        if self._oidx:
            olib().xgo_index_free(self._oidx)
            self._oidx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def oracle_search_batch(corpus, queries, first, maxitems, n_threads=None, reference_select_bug=False):
    """Every query once on n_threads host threads (oracle/xgm_oracle.cc::xgo_search_batch) → [(rows, hdr dict)]."""
    ol = olib()
    nq = len(queries)
    n_threads = n_threads or max(1, min(os.cpu_count() or 1, nq))
    flat = [(t if isinstance(t, bytes) else t.encode()) for q in queries for t in q["terms"]]
    ops = (C.c_uint32 * nq)(*[OPS[q["op"]] | (((q.get("n_required") or 1) << 8) if q["op"] in SIDED else 0) for q in queries])
    nts = (C.c_uint32 * nq)(*[len(q["terms"]) for q in queries])
    wins = (C.c_uint32 * nq)(*[q.get("window", 0) for q in queries])
    terms = (C.c_char_p * len(flat))(*flat)
    lens = (C.c_uint32 * len(flat))(*[len(t) for t in flat])
    cap = max(1, first + maxitems)
    hits = (OHit * (nq * cap))()
    hdrs = (OHdr * nq)()
    ol.xgo_search_batch.restype = C.c_int
    ol.xgo_search_batch.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.POINTER(OHit), C.POINTER(OHdr)]
    rc = ol.xgo_search_batch(corpus.oracle_index(), nq, ops, nts, wins, terms, lens, first, maxitems, cap, n_threads, 1 if reference_select_bug else 0, hits, hdrs)
    assert rc == 0
    out = []
    for i in range(nq):
        h = hdrs[i]
        rows = [(hits[i * cap + j].docid, hits[i * cap + j].weight, hits[i * cap + j].subqs) for j in range(h.n_hits)]
        out.append((rows, dict(n_hits=h.n_hits, matches=h.matches, max_attained=h.max_attained, max_possible=h.max_possible, max_subqs=h.max_subqs)))
    return out


# ---- nested queries: a tree of tuples <-> the post-order token form of xapian_ref's "RPN" queries ----
#   tree := "term" | ("term", wqf) | ("AND", t, t, ...) | ("OR", ...) | ("SYN", term, ...) | ("AND_NOT", left, right...) |
#           ("AND_MAYBE", left, right...) | ("FILTER", left, right) | ("SCALE", factor, t)
T_KIND = {"TERM": 0, "AND": 1, "OR": 2, "AND_NOT": 3, "AND_MAYBE": 4, "FILTER": 5, "SYN": 6, "SCALE": 7}
_RPN_SIGN = {"AND": "&", "OR": "|", "SYN": "=", "AND_NOT": "-", "AND_MAYBE": "?"}


def tree_from_json(t):
    """JSON turns the tuples of a tree into lists: back to tuples (a ["term", wqf] pair stays a pair)."""
    if isinstance(t, list):
        return tuple(tree_from_json(x) for x in t)
    return t


def tree_rpn(tree):
    """Post-order token list of a tree (the query-file form)."""
    if isinstance(tree, str):
        return [tree]
    if len(tree) == 2 and isinstance(tree[1], int) and isinstance(tree[0], str) and tree[0] not in T_KIND:
        return ["%s#%d" % tree]
    op = tree[0]
    if op == "SCALE":
        return tree_rpn(tree[2]) + ["*%r" % float(tree[1])]
    toks = [x for k in tree[1:] for x in tree_rpn(k)]
    if op == "FILTER":
        return toks + ["!"]
    return toks + ["%s%d" % (_RPN_SIGN[op], len(tree) - 1)]


def tree_program(tree):
    """(terms, wqf, ops) with ops = [(kind, arity, term index, scale)] in post-order; terms distinct, in first-use order."""
    terms, wqf, ops = [], [], []

    def walk(t):
        if isinstance(t, str) or (len(t) == 2 and isinstance(t[1], int) and t[0] not in T_KIND):
            name, w = (t, 1) if isinstance(t, str) else t
            if name in terms:
                raise ValueError("repeated term %s" % name)
            terms.append(name); wqf.append(w)
            ops.append((T_KIND["TERM"], 0, len(terms) - 1, 1.0))
            return
        op = t[0]
        if op == "SCALE":
            walk(t[2])
            ops.append((T_KIND["SCALE"], 1, 0, float(t[1])))
            return
        for k in t[1:]:
            walk(k)
        ops.append((T_KIND[op], len(t) - 1, 0, 1.0))
    walk(tree)
    return terms, wqf, ops


def oracle_search_tree(corpus, tree, first, maxitems, global_stats=None):
    """The CPU oracle on a nested query.  Returns (rows, hdr, total_subqs)."""
    ol = olib()
    terms, wqf, ops = tree_program(tree)
    tb = [t.encode() for t in terms]
    n = len(tb)
    arr = (C.c_char_p * n)(*tb)
    lens = (C.c_uint32 * n)(*[len(t) for t in tb])
    wq = (C.c_uint32 * n)(*wqf)
    m = len(ops)
    kind = (C.c_uint8 * m)(*[o[0] for o in ops])
    arity = (C.c_uint8 * m)(*[o[1] for o in ops])
    tix = (C.c_uint16 * m)(*[o[2] for o in ops])
    scale = (C.c_double * m)(*[o[3] for o in ops])
    cap = max(1, first + maxitems)
    hits = (OHit * cap)()
    hdr = OHdr()
    tot = C.c_uint32()
    ol.xgo_search_tree.restype = C.c_int
    ol.xgo_search_tree.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32,
                                   C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), C.POINTER(C.c_double), C.c_uint32, C.c_uint32,
                                   C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(OHit), C.POINTER(OHdr), C.POINTER(C.c_uint32)]
    if global_stats is None:
        rc = ol.xgo_search_tree(corpus.oracle_index(), n, arr, lens, wq, m, kind, arity, tix, scale, first, maxitems, 0, 0, 0, None, hits, C.byref(hdr), C.byref(tot))
    else:
        tf = (C.c_uint32 * n)(*[global_stats["termfreq"][t] for t in terms])
        rc = ol.xgo_search_tree(corpus.oracle_index(), n, arr, lens, wq, m, kind, arity, tix, scale, first, maxitems, 1,
                                global_stats["total_length"], global_stats["collection_size"], tf, hits, C.byref(hdr), C.byref(tot))
    assert rc == 0, rc
    return [(hits[i].docid, hits[i].weight, hits[i].subqs) for i in range(hdr.n_hits)], hdr, tot.value


def gen_tree_queries(n_queries, rank_lo=1, rank_hi=300, seed=QUERY_SEED):
    """Random nested queries over distinct terms: the shapes Xapiand's DSL builds (boolean combinations, _filter / _and_not /
    _and_maybe around them, synonym groups from wildcards, boosts)."""
    rng = random.Random(seed)

    def terms(k, used):
        out = []
        while len(out) < k:
            t = "t%d" % max(1, log_uniform_rank(rng, rank_lo, rank_hi))
            if t not in used:
                used.add(t); out.append(t)
        return out
    qs = []
    for i in range(n_queries):
        used = set()
        shape = i % 12
        a = terms(8, used)
        if shape == 0:
            tree = ("AND", ("OR", a[0], a[1]), a[2])
        elif shape == 1:
            tree = ("OR", ("AND", a[0], a[1]), a[2], ("AND", a[3], a[4]))
        elif shape == 2:
            tree = ("AND", ("OR", a[0], a[1], a[2]), ("OR", a[3], a[4]), a[5])
        elif shape == 3:
            tree = ("AND_NOT", ("OR", a[0], a[1]), ("AND", a[2], a[3]))
        elif shape == 4:
            tree = ("AND_MAYBE", ("AND", a[0], a[1]), ("OR", a[2], ("AND", a[3], a[4])))
        elif shape == 5:
            tree = ("FILTER", ("OR", a[0], a[1], a[2]), ("OR", a[3], a[4]))
        elif shape == 6:
            tree = ("OR", ("SYN", a[0], a[1], a[2]), a[3])
        elif shape == 7:
            tree = ("AND", ("SYN", a[0], a[1]), ("SCALE", 2.5, a[2]), (a[3], 3))
        elif shape == 8:
            tree = ("SCALE", 0.5, ("OR", a[0], ("SCALE", 3.0, ("AND", a[1], a[2])), a[3]))
        elif shape == 9:
            tree = ("AND", ("AND", a[0], ("OR", a[1], a[2])), ("FILTER", a[3], ("OR", a[4], a[5])))
        elif shape == 10:
            tree = ("OR", ("AND_NOT", a[0], a[1]), ("AND_MAYBE", a[2], a[3]), (a[4], 2))
        else:
            tree = ("AND_MAYBE", ("SYN", a[0], a[1]), ("SYN", a[2], a[3], a[4]), a[5])
        qs.append(dict(op="RPN", tree=tree, terms=tree_rpn(tree), first=0, maxitems=10, window=0))
    return qs
