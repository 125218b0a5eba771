"""Host-side mirror of the reference surface for the match/rank path.

Names, argument meaning and error behaviour follow Xapian's (reference src/xapian/query.h,
enquire.h, mset.h) so the parity tests read like Xapian code:

    db = Database("shard0.seg")                       # ~ Xapian::Database(path)
    enq = Enquire(db)                                 # ~ Xapian::Enquire enq(db)
    enq.set_query(Query(Query.OP_AND, ["t1", "t5"]))  # ~ enq.set_query(Xapian::Query(OP_AND, ...))
    mset = enq.get_mset(0, 10)                        # ~ enq.get_mset(first, maxitems)
    for item in mset: item.docid, item.weight, item.rank, item.percent

Everything below the method signatures goes through the C ABI (include/xgm.h) to the HIP kernels.
Query shapes the device path declines raise `Unsupported` — in Xapiand the hook would simply let the
CPU matcher run; this package has no CPU matcher.
"""
import ctypes as C
import sys

from . import _lib
from ._lib import XgmError, XgmUnsupported as Unsupported  # noqa: F401  (re-exported)

DBL_EPSILON = sys.float_info.epsilon


class Query:
    """Subset of Xapian::Query (reference src/xapian/query.h:48-): a term; AND / OR / PHRASE of terms; or
    AND_NOT / AND_MAYBE / FILTER of (a term or an AND of terms, a term or an OR of terms — an AND for
    FILTER), the trees Xapiand's DSL builds for _and_not / _and_maybe / _filter."""
    OP_AND, OP_OR, OP_PHRASE, OP_NEAR = "AND", "OR", "PHRASE", "NEAR"
    OP_AND_NOT, OP_AND_MAYBE, OP_FILTER = "AND_NOT", "AND_MAYBE", "FILTER"
    LEAF_TERM = "TERM"
    _OPS = {OP_AND: _lib.XGM_OP_AND, OP_OR: _lib.XGM_OP_OR, OP_PHRASE: _lib.XGM_OP_PHRASE, OP_NEAR: _lib.XGM_OP_NEAR,
            OP_AND_NOT: _lib.XGM_OP_AND_NOT, OP_AND_MAYBE: _lib.XGM_OP_AND_MAYBE, OP_FILTER: _lib.XGM_OP_FILTER}
    _SIDED = (OP_AND_NOT, OP_AND_MAYBE, OP_FILTER)

    def __init__(self, op_or_term, subqueries=None, window=0, n_required=0):
        self.n_required = 0
        if subqueries is None:
            self.op = Query.LEAF_TERM
            self.terms = [_as_bytes(op_or_term)]
            self.window = 0
            return
        if op_or_term not in Query._OPS:
            raise ValueError("unsupported query operator %r" % (op_or_term,))
        if op_or_term in Query._SIDED and not n_required:
            # nested form: (left, right)
            if len(subqueries) != 2:
                raise Unsupported("%s takes a left and a right subquery" % op_or_term)
            sides = []
            for s, ok in zip(subqueries, ((Query.OP_AND,), (Query.OP_AND,) if op_or_term == Query.OP_FILTER else (Query.OP_OR,))):
                if not isinstance(s, Query):
                    s = Query(s)
                if s.op != Query.LEAF_TERM and (s.op not in ok or len(s.terms) < 1):
                    raise Unsupported("this shape of %s is not handled by the device path" % op_or_term)
                sides.append(s.terms)
            self.op, self.terms, self.window, self.n_required = op_or_term, sides[0] + sides[1], 0, len(sides[0])
            return
        terms = []
        for s in subqueries:
            if isinstance(s, Query):
                if s.op != Query.LEAF_TERM:
                    raise Unsupported("nested operators are not handled by the device path")
                terms.append(s.terms[0])
            else:
                terms.append(_as_bytes(s))
        if not terms:
            raise ValueError("empty query")
        self.op = op_or_term
        self.terms = terms
        self.window = window
        if op_or_term in Query._SIDED:
            if not 1 <= n_required < len(terms):
                raise ValueError("n_required must leave terms on both sides")
            self.n_required = n_required

    # -- nested queries ---------------------------------------------------------------------------------------------
    _T = {"AND": 1, "OR": 2, "AND_NOT": 3, "AND_MAYBE": 4, "FILTER": 5, "SYN": 6, "SYNONYM": 6, "SCALE": 7}

    @classmethod
    def tree(cls, spec):
        """A nested query from tuples: "term" | ("term", wqf) | ("AND", q, q, ...) | ("OR", ...) | ("SYNONYM", term, ...) |
        ("AND_NOT", left, right...) | ("AND_MAYBE", left, right...) | ("FILTER", left, right) | ("SCALE", factor, q) —
        the Xapian::Query trees Xapiand's DSL builds (reference src/query_dsl.cc:188-432), lowered by xgm_plan_query like
        src/xapian/api/queryinternal.cc does."""
        q = cls.__new__(cls)
        q.op, q.terms, q.window, q.n_required = "TREE", [], 0, 0
        q.wqf, q.ops = [], []

        def walk(t):
            if isinstance(t, (str, bytes)) or (len(t) == 2 and isinstance(t[1], int) and t[0] not in cls._T):
                name, w = (t, 1) if isinstance(t, (str, bytes)) else t
                q.terms.append(_as_bytes(name)); q.wqf.append(w)
                q.ops.append((0, 0, len(q.terms) - 1, 1.0))
                return
            if t[0] == "SCALE":
                walk(t[2])
                q.ops.append((7, 1, 0, float(t[1])))
                return
            for k in t[1:]:
                walk(k)
            q.ops.append((cls._T[t[0]], len(t) - 1, 0, 1.0))
        walk(spec)
        return q

    def get_type(self):
        return self.op

    def total_subqs(self):
        """(flat shapes; a nested query's count comes back in its plan: xgm_query.total_subqs)
        Weighted leaves of the query = what QueryOptimiser::inc_total_subqs counts (reference
        src/xapian/api/queryinternal.cc:1049-1056): unweighted sides (AND_NOT, FILTER) do not count."""
        if self.op in (Query.OP_AND_NOT, Query.OP_FILTER):
            return self.n_required
        return len(self.terms)

    def get_num_subqueries(self):
        return 0 if self.op == Query.LEAF_TERM else len(self.terms)

    def empty(self):
        return not self.terms

    def get_description(self):
        if self.op == Query.LEAF_TERM:
            return "Query(%s)" % self.terms[0].decode("utf-8", "replace")
        names = [t.decode("utf-8", "replace") for t in self.terms]
        if self.op in Query._SIDED:
            inner = "AND" if self.op == Query.OP_FILTER else "OR"
            return "Query(((%s) %s (%s)))" % (" AND ".join(names[:self.n_required]), self.op, (" %s " % inner).join(names[self.n_required:]))
        sep = {"AND": " AND ", "OR": " OR ", "PHRASE": " PHRASE %d " % (self.window or len(self.terms)),
               "NEAR": " NEAR %d " % (self.window or len(self.terms))}[self.op]
        return "Query((" + sep.join(names) + "))"


def _as_bytes(t):
    return t if isinstance(t, bytes) else str(t).encode("utf-8")


class BM25Weight:
    """Parameters of Xapian::BM25Weight (reference src/xapian/weight.h:635-667 defaults)."""

    def __init__(self, k1=1.0, k2=0.0, k3=1.0, b=0.5, min_normlen=0.5):
        self.k1, self.k2, self.k3, self.b, self.min_normlen = k1, k2, k3, b, min_normlen

    def name(self):
        return "Xapian::BM25Weight"


class Database:
    """A device-resident shard: one XGMSEG1 segment loaded into HBM (or built there)."""

    def __init__(self, path=None, device=0, revision=None, _handle=None):
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
        else:
            rev = _lib.UINT64_MAX if revision is None else revision
            _lib.check(_lib.lib().xgm_index_open(_as_bytes(path), device, rev, C.byref(self._h)))
        self._info = _lib.IndexInfo()
        _lib.check(_lib.lib().xgm_index_get_info(self._h, C.byref(self._info)))

    @classmethod
    def synthetic(cls, seed, n_docs_global, vocab, len_lo=50, len_hi=150, n_shards=1, shard=0, device=0,
                  stripe_bits=0, with_positions=True):
        """Build the synthetic corpus shard (tools/xgm_corpus.h) directly in HBM with the GPU builder."""
        p = _lib.SynthParams(seed, vocab, len_lo, len_hi, n_docs_global, n_shards, shard, stripe_bits,
                             1 if with_positions else 0)
        h = C.c_void_p()
        _lib.check(_lib.lib().xgm_index_build_synthetic(C.byref(p), device, C.byref(h)))
        return cls(_handle=h)

    def close(self):
        if self._h:
            _lib.lib().xgm_index_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- Xapian::Database accessors -------------------------------------------------------------
    def get_doccount(self):
        return self._info.doccount

    def get_lastdocid(self):
        return self._info.lastdocid

    def get_total_length(self):
        return self._info.total_length

    def get_average_length(self):
        return self._info.total_length / self._info.doccount if self._info.doccount else 0.0

    def has_positions(self):
        return bool(self._info.has_positions)

    def get_revision(self):
        return self._info.revision

    def info(self):
        return self._info

    def get_termfreq(self, term):
        tf = C.c_uint32()
        t = _as_bytes(term)
        _lib.check(_lib.lib().xgm_lookup_term(self._h, t, len(t), None, C.byref(tf), None, None))
        return tf.value

    def get_collection_freq(self, term):
        cf = C.c_uint32()
        t = _as_bytes(term)
        _lib.check(_lib.lib().xgm_lookup_term(self._h, t, len(t), None, None, C.byref(cf), None))
        return cf.value

    def get_wdf_upper_bound(self, term):
        ub = C.c_uint32()
        t = _as_bytes(term)
        _lib.check(_lib.lib().xgm_lookup_term(self._h, t, len(t), None, None, None, C.byref(ub)))
        return ub.value

    def save(self, path):
        _lib.check(_lib.lib().xgm_index_save(self._h, _as_bytes(path)))

    def attach_column(self, column_path):
        """xgm_index_attach_column: the ordinals of a value slot (a column file of xgm_glass_export_column) into HBM.  The distinct
        values stay on the host: MSet items get their sort / collapse keys from them."""
        import struct
        _lib.check(_lib.lib().xgm_index_attach_column(self._h, _as_bytes(column_path)))
        with open(column_path, "rb") as f:
            slot = struct.unpack_from("<I", f.read(12), 8)[0]
        if not hasattr(self, "_column_values"):
            self._column_values = {}
        self._column_values[slot] = read_column_values(column_path)

    def column_values(self, slot):
        vals = getattr(self, "_column_values", {}).get(slot)
        if vals is None:
            raise Unsupported("no column attached for value slot %d" % slot)
        return vals

    def set_stream(self, hip_stream):
        _lib.check(_lib.lib().xgm_index_set_stream(self._h, C.c_void_p(hip_stream)))

    def set_near_colocated(self, may_exist=True):
        """Distinct terms may share a position in this shard: OP_NEAR by NearPostList's full procedure (nearpostlist.cc:106-140)."""
        _lib.check(_lib.lib().xgm_index_set_near_colocated(self._h, 1 if may_exist else 0))

    def set_profiling(self, on):
        """on: bit 0 = time the match kernel with HIP events, bit 1 = launch the tallying instantiation (xgm.h)."""
        _lib.check(_lib.lib().xgm_index_set_profiling(self._h, int(on)))

    def last_kernel_ms(self):
        return _lib.lib().xgm_last_kernel_ms(self._h)

    def last_kernel_name(self):
        return (_lib.lib().xgm_last_kernel_name(self._h) or b"").decode()


class MSetItem:
    __slots__ = ("docid", "weight", "rank", "percent", "subqs_matched", "sort_key", "collapse_key", "collapse_count")

    def __init__(self, docid, weight, rank, percent, subqs_matched):
        self.docid, self.weight, self.rank, self.percent, self.subqs_matched = docid, weight, rank, percent, subqs_matched
        self.sort_key, self.collapse_key, self.collapse_count = b"", b"", 0        # MSetIterator::get_sort_key / get_collapse_key / get_collapse_count

    def __repr__(self):
        return "MSetItem(rank=%d, docid=%d, weight=%r, percent=%d)" % (self.rank, self.docid, self.weight, self.percent)


class MSet:
    """Mirror of Xapian::MSet for the fields the path produces.  matches_* are EXACT here (the
    reference estimates them, protomset.h:497-619) — documented parity exception."""

    def __init__(self, first, hits, hdr, total_subqs, plan=None):
        self._first = first
        self._bounds = None
        if plan is not None:
            lb, est, ub = C.c_uint32(), C.c_uint32(), C.c_uint32()
            _lib.lib().xgm_mset_bounds(C.byref(plan), C.byref(hdr), C.byref(lb), C.byref(est), C.byref(ub))
            self._bounds = (lb.value, est.value, ub.value)
        # MSet::Internal percent_scale_factor, protomset.h:466-471 and :682
        if hdr.n_hits and hdr.max_attained != 0.0 and total_subqs:
            scale = hdr.max_weight_subqs_matched / float(total_subqs)
            scale /= hdr.max_attained
            self._percent_scale_factor = scale * 100.0
        else:
            self._percent_scale_factor = 0.0
        self._items = []
        for rank, h in enumerate(hits):
            if rank < first:
                continue
            self._items.append(MSetItem(h.docid, h.weight, rank, self.convert_to_percent(h.weight), h.subqs_matched))
        self._matches = hdr.matches_exact & ~_lib.XGM_MATCHES_LOWER_BOUND
        self._matches_is_lower_bound = bool(hdr.matches_exact & _lib.XGM_MATCHES_LOWER_BOUND)
        self._max_possible = hdr.max_possible
        self._max_attained = hdr.max_attained

    def convert_to_percent(self, weight):
        """MSet::Internal::convert_to_percent, reference src/xapian/api/mset.cc:334-362."""
        if self._percent_scale_factor == 0.0:
            return 100
        if weight <= 0.0:
            return 0
        percent = int(weight * self._percent_scale_factor + 100.0 * DBL_EPSILON)
        if percent <= 0:
            return 1
        return min(percent, 100)

    def size(self):
        return len(self._items)

    def __len__(self):
        return len(self._items)

    def __iter__(self):
        return iter(self._items)

    def __getitem__(self, i):
        return self._items[i]

    def get_firstitem(self):
        return self._first

    def get_matches_exact(self):
        """The exact number of matching documents (the reference only bounds and estimates it) — unless
        matches_is_lower_bound(): a positional query answered with check_at_least inside the page."""
        return self._matches

    def matches_is_lower_bound(self):
        return self._matches_is_lower_bound

    def get_matches_estimated(self):
        return self._bounds[1] if self._bounds else self._matches

    def get_matches_lower_bound(self):
        return self._bounds[0] if self._bounds else self._matches

    def get_matches_upper_bound(self):
        """Static bound of the postlist tree the reference would build: identical to Xapian's (xgm_mset_bounds)."""
        return self._bounds[2] if self._bounds else self._matches

    def get_max_possible(self):
        return self._max_possible

    def get_max_attained(self):
        return self._max_attained


def _desc(query, first, maxitems, check_at_least, weight):
    n = len(query.terms)
    if n > _lib.XGM_MAX_TERMS:
        raise Unsupported("too many terms")
    d = _lib.QueryDesc()
    if query.op == Query.LEAF_TERM:
        d.op = _lib.XGM_OP_AND
    elif query.op == "TREE":
        d.op = _lib.XGM_OP_TREE
        if len(query.ops) > _lib.XGM_MAX_TREE:
            raise Unsupported("query tree too large")
        d.n_tree = len(query.ops)
        for i, (kind, arity, term, scale) in enumerate(query.ops):
            d.tree[i].kind, d.tree[i].arity, d.tree[i].term = kind, arity, term
            d.tree_scale[i] = scale
        for i, w in enumerate(query.wqf):
            d.wqf[i] = w
    else:
        d.op = Query._OPS[query.op]
    d.n_terms = n
    for i, t in enumerate(query.terms):
        d.terms[i] = t
        d.term_len[i] = len(t)
    d.window = query.window
    d.n_required = query.n_required
    d.first, d.maxitems, d.check_at_least = first, maxitems, check_at_least
    d.k1, d.k2, d.k3, d.b, d.min_normlen = weight.k1, weight.k2, weight.k3, weight.b, weight.min_normlen
    return d


def plan(db, query, first, maxitems, check_at_least=0, weight=None, global_stats=None):
    """xgm_plan_query: lower a Query against one shard (optionally with merged statistics)."""
    weight = weight or BM25Weight()
    d = _desc(query, first, maxitems, check_at_least, weight)
    q = _lib.Query()
    gs = C.byref(global_stats) if global_stats is not None else None
    _lib.check(_lib.lib().xgm_plan_query(db._h, C.byref(d), gs, C.byref(q)))
    q._keepalive = d
    return q


def read_column_values(path):
    """The distinct values of a column file (xgm_glass_export_column), ascending: ordinal o > 0 of a hit is values[o - 1]."""
    import struct
    b = open(path, "rb").read()
    if b[:8] != b"XGMCOL1\0":
        raise ValueError("%s is not a column file" % path)
    slot, lastdocid, n, _ = struct.unpack_from("<4I", b, 8)
    o = 24 + 4 * (lastdocid + 1)
    off = struct.unpack_from("<%dQ" % (n + 1), b, o)
    base = o + 8 * (n + 1)
    return [b[base + off[i]:base + off[i + 1]] for i in range(n)]


def search_sorted(db, planned, sort_by, slot, reverse=False):
    """xgm_search_sorted: one planned query under Enquire::set_sort_by_value (sort_by 1) / _value_then_relevance (2) /
    _relevance_then_value (3) on a value slot whose column is attached (Database.attach_column).
    Returns ([(docid, weight, subqs, ordinal)], hdr)."""
    k = max(1, planned.first + planned.maxitems)
    hits = (_lib.Hit * k)()
    ords = (C.c_uint32 * k)()
    hdr = _lib.ResultHdr()
    spec = _lib.SortSpec(sort_by, slot, 1 if reverse else 0, 0)
    _lib.check(_lib.lib().xgm_search_sorted(db._h, C.byref(planned), C.byref(spec), hits, ords, C.byref(hdr)))
    return [(hits[i].docid, hits[i].weight, hits[i].subqs_matched, ords[i]) for i in range(hdr.n_hits)], hdr


def search_sorted_batch(db, plans, sort_by, slot, reverse=False):
    """xgm_search_sorted_batch: every planned query under ONE sort in one launch → [([(docid, weight, subqs, ordinal)], hdr)]."""
    nq = len(plans)
    ks = max(1, max(p.first + p.maxitems for p in plans))
    qs = (_lib.Query * nq)(*plans)
    hits = (_lib.Hit * (nq * ks))()
    ords = (C.c_uint32 * (nq * ks))()
    hdrs = (_lib.ResultHdr * nq)()
    spec = _lib.SortSpec(sort_by, slot, 1 if reverse else 0, 0)
    _lib.check(_lib.lib().xgm_search_sorted_batch(db._h, qs, nq, C.byref(spec), ks, hits, ords, hdrs))
    return [([(hits[q * ks + i].docid, hits[q * ks + i].weight, hits[q * ks + i].subqs_matched, ords[q * ks + i]) for i in range(hdrs[q].n_hits)], hdrs[q]) for q in range(nq)]


def search_sorted_spy_batch(db, plans, sort_by, slot, reverse, spy_slot, n_distinct):
    """xgm_search_sorted_spy_batch: search_sorted_batch with a ValueCountMatchSpy on spy_slot for every query → [(hits, hdr, counts)]."""
    nq = len(plans)
    ks = max(1, max(p.first + p.maxitems for p in plans))
    qs = (_lib.Query * nq)(*plans)
    hits = (_lib.Hit * (nq * ks))()
    ords = (C.c_uint32 * (nq * ks))()
    hdrs = (_lib.ResultHdr * nq)()
    nc = n_distinct + 1
    counts = (C.c_uint32 * (nq * nc))()
    spec = _lib.SortSpec(sort_by, slot, 1 if reverse else 0, 0)
    _lib.check(_lib.lib().xgm_search_sorted_spy_batch(db._h, qs, nq, C.byref(spec), ks, hits, ords, hdrs, spy_slot, counts, nc))
    return [([(hits[q * ks + i].docid, hits[q * ks + i].weight, hits[q * ks + i].subqs_matched, ords[q * ks + i]) for i in range(hdrs[q].n_hits)], hdrs[q],
             list(counts[q * nc:(q + 1) * nc])) for q in range(nq)]


def search_sorted_spy(db, planned, sort_by, slot, reverse, spy_slot, n_distinct):
    """xgm_search_sorted_spy: search_sorted plus a ValueCountMatchSpy on spy_slot (a column with n_distinct values attached).
    Returns (hits, hdr, counts) with counts[o] = matching documents whose value has ordinal o (0 = no value)."""
    k = max(1, planned.first + planned.maxitems)
    hits = (_lib.Hit * k)()
    ords = (C.c_uint32 * k)()
    hdr = _lib.ResultHdr()
    counts = (C.c_uint32 * (n_distinct + 1))()
    spec = _lib.SortSpec(sort_by, slot, 1 if reverse else 0, 0)
    _lib.check(_lib.lib().xgm_search_sorted_spy(db._h, C.byref(planned), C.byref(spec), hits, ords, C.byref(hdr), spy_slot, counts, n_distinct + 1))
    return [(hits[i].docid, hits[i].weight, hits[i].subqs_matched, ords[i]) for i in range(hdr.n_hits)], hdr, list(counts)


def search_collapsed(db, planned, collapse_slot, collapse_max, sort_by=None, slot=0, reverse=False):
    """xgm_search_collapsed: set_collapse_key(collapse_slot, collapse_max), ranked by relevance (sort_by None) or under a value sort.
    Returns ([(docid, weight, subqs, sort ordinal, collapse ordinal, collapse count)], hdr, collapsed lower bound)."""
    k = max(1, planned.first + planned.maxitems)
    hits = (_lib.Hit * k)()
    ords, cords, ccounts = (C.c_uint32 * k)(), (C.c_uint32 * k)(), (C.c_uint32 * k)()
    hdr = _lib.ResultHdr()
    clb = C.c_uint64()
    spec = _lib.SortSpec(sort_by, slot, 1 if reverse else 0, 0) if sort_by else None
    _lib.check(_lib.lib().xgm_search_collapsed(db._h, C.byref(planned), C.byref(spec) if spec else None, collapse_slot, collapse_max, hits, ords, cords, ccounts,
                                               C.byref(hdr), C.byref(clb)))
    return [(hits[i].docid, hits[i].weight, hits[i].subqs_matched, ords[i], cords[i], ccounts[i]) for i in range(hdr.n_hits)], hdr, clb.value


def search_collapsed_batch(db, plans, collapse_slot, collapse_max, sort_by=None, slot=0, reverse=False):
    """xgm_search_collapsed_batch: every planned query under one collapse key (and one sort or relevance) in one launch →
    [([(docid, weight, subqs, sort ordinal, collapse ordinal, collapse count)], hdr, collapsed lower bound)]."""
    nq = len(plans)
    ks = max(1, max(p.first + p.maxitems for p in plans))
    qs = (_lib.Query * nq)(*plans)
    hits = (_lib.Hit * (nq * ks))()
    ords, cords, ccounts = (C.c_uint32 * (nq * ks))(), (C.c_uint32 * (nq * ks))(), (C.c_uint32 * (nq * ks))()
    hdrs = (_lib.ResultHdr * nq)()
    clb = (C.c_uint64 * nq)()
    spec = _lib.SortSpec(sort_by, slot, 1 if reverse else 0, 0) if sort_by else None
    _lib.check(_lib.lib().xgm_search_collapsed_batch(db._h, qs, nq, C.byref(spec) if spec else None, collapse_slot, collapse_max, ks, hits, ords, cords,
                                                     ccounts, hdrs, clb))
    return [([(hits[q * ks + i].docid, hits[q * ks + i].weight, hits[q * ks + i].subqs_matched, ords[q * ks + i], cords[q * ks + i], ccounts[q * ks + i])
              for i in range(hdrs[q].n_hits)], hdrs[q], clb[q]) for q in range(nq)]


def search_all(db, planned, cap=None):
    """xgm_search_all: EVERY matching document of one planned query in ascending docid order — the sequence the reference's matcher
    loop is shown (matcher.cc:482-536).  Returns ([(docid, weight, subqs)], hdr); cap defaults to the plan's own upper bound."""
    if cap is None:
        cap = max(1, planned.est_max or db.get_doccount())
    hits = (_lib.Hit * max(1, cap))()
    hdr = _lib.ResultHdr()
    n = C.c_uint64()
    _lib.check(_lib.lib().xgm_search_all(db._h, C.byref(planned), hits, cap, C.byref(n), C.byref(hdr)))
    if n.value > cap:
        return None, hdr                      # (the room needed is hdr.matches_exact)
    return [(hits[i].docid, hits[i].weight, hits[i].subqs_matched) for i in range(n.value)], hdr


REPLAY_COUNT, REPLAY_FROZEN_WEIGHT = 0, 1


def search_replay(db, planned, mode=REPLAY_COUNT):
    """xgm_search_replay: ProtoMSet's collation of one planned query replayed on the device over its whole match in docid order
    (protomset.h:340-400; for PHRASE / NEAR with REPLAY_FROZEN_WEIGHT also SelectPostList's frozen weight, selectpostlist.cc:28-55).
    Returns ([(docid, weight, subqs)] = the page as the reference keeps it, hdr, known_matching_docs)."""
    k = max(1, planned.first + planned.maxitems)
    hits = (_lib.Hit * k)()
    hdr = _lib.ResultHdr()
    known = C.c_uint64()
    _lib.check(_lib.lib().xgm_search_replay(db._h, C.byref(planned), mode, hits, C.byref(hdr), C.byref(known)))
    return [(hits[i].docid, hits[i].weight, hits[i].subqs_matched) for i in range(hdr.n_hits)], hdr, known.value


def search_batch_replay(db, plans, replay=_lib.XGM_REPLAY_BATCH_FROZEN):
    """A batch in flight whose queries carry XGM_REPLAY_BATCH_* bits (xgm_search_batch_begin / xgm_batch_end / xgm_batch_known): the reference's own
    answers at batch throughput → list of ([(docid, weight, subqs)], hdr, known_matching_docs incl. XGM_KNOWN_LOWER_BOUND)."""
    nq = len(plans)
    if nq == 0:
        return []
    k_stride = max(1, max(p.first + p.maxitems for p in plans))
    qs = (_lib.Query * nq)()
    for i, p in enumerate(plans):
        C.memmove(C.byref(qs[i]), C.byref(p), C.sizeof(_lib.Query))
        qs[i].replay = replay
    L = _lib.lib()
    f = C.c_void_p()
    _lib.check(L.xgm_search_batch_begin(db._h, qs, nq, k_stride, C.byref(f)))
    try:
        hp, dp, kp = C.POINTER(_lib.Hit)(), C.POINTER(_lib.ResultHdr)(), C.POINTER(C.c_uint64)()
        _lib.check(L.xgm_batch_end(f, C.byref(hp), C.byref(dp)))
        _lib.check(L.xgm_batch_known(f, C.byref(kp)))
        out = []
        for i in range(nq):
            hdr = _lib.ResultHdr()
            C.memmove(C.byref(hdr), C.byref(dp[i]), C.sizeof(_lib.ResultHdr))
            out.append(([(hp[i * k_stride + j].docid, hp[i * k_stride + j].weight, hp[i * k_stride + j].subqs_matched) for j in range(hdr.n_hits)], hdr,
                        kp[i] if kp else 0))
        return out
    finally:
        L.xgm_batch_release(f)


def search_batch(db, plans):
    """xgm_search_batch over already planned queries → list of (hits[], hdr)."""
    nq = len(plans)
    if nq == 0:
        return []
    k_stride = max(1, max(p.first + p.maxitems for p in plans))
    qs = (_lib.Query * nq)(*plans)
    hits = (_lib.Hit * (nq * k_stride))()
    hdrs = (_lib.ResultHdr * nq)()
    _lib.check(_lib.lib().xgm_search_batch(db._h, qs, nq, k_stride, hits, hdrs))
    out = []
    for i in range(nq):
        n = hdrs[i].n_hits
        out.append(([hits[i * k_stride + j] for j in range(n)], hdrs[i]))
    return out


class ValueCountMatchSpy:
    """Mirror of Xapian::ValueCountMatchSpy (reference src/xapian/api/matchspy.cc:296-340): how many of the matching documents carry
    each value of a slot."""

    def __init__(self, slot):
        self.slot = slot
        self._total = 0
        self._values = {}

    def _add(self, total, counts):
        self._total += total
        for v, n in counts.items():
            self._values[v] = self._values.get(v, 0) + n

    def get_total(self):
        return self._total

    def values(self):
        """(value, frequency) in value order: ValueCountMatchSpy::values_begin() .. values_end()."""
        return sorted(self._values.items())


class Enquire:
    """Mirror of Xapian::Enquire for BM25 searches on one shard: by relevance, or — first version, DESIGN.md 8 (f).3 — under a value
    sort, with a collapse key, with a ValueCountMatchSpy."""

    def __init__(self, db):
        self._db = db
        self._query = None
        self._weight = BM25Weight()

    def set_query(self, query):
        self._query = query

    def get_query(self):
        return self._query

    def set_weighting_scheme(self, weight):
        if not isinstance(weight, BM25Weight):
            raise Unsupported("only BM25Weight runs on the device path")
        self._weight = weight

    # Enquire::set_sort_by_* / set_collapse_key / add_matchspy (reference src/xapian/api/enquire.cc): value slots whose columns are
    # attached to the database (Database.attach_column)
    def set_sort_by_relevance(self):
        self._sort = None

    def set_sort_by_value(self, sort_key, reverse):
        self._sort = (_lib.XGM_SORT_VALUE, sort_key, bool(reverse))

    def set_sort_by_value_then_relevance(self, sort_key, reverse):
        self._sort = (_lib.XGM_SORT_VALUE_RELEVANCE, sort_key, bool(reverse))

    def set_sort_by_relevance_then_value(self, sort_key, reverse):
        self._sort = (_lib.XGM_SORT_RELEVANCE_VALUE, sort_key, bool(reverse))

    def set_collapse_key(self, collapse_key, collapse_max=1):
        self._collapse = (collapse_key, collapse_max) if collapse_max else None

    def add_matchspy(self, spy):
        if not isinstance(spy, ValueCountMatchSpy):
            raise Unsupported("only ValueCountMatchSpy runs on the device path")
        self._spies = getattr(self, "_spies", []) + [spy]

    def clear_matchspies(self):
        self._spies = []

    def _get_mset_by_value(self, p, sort, collapse, spies):
        db = self._db
        if collapse:
            if spies:
                raise Unsupported("a collapse key together with a MatchSpy")
            rows, hdr, _ = search_collapsed(db, p, collapse[0], collapse[1], *(sort if sort else (None, 0, False)))
        elif spies:
            if len(spies) > 1 or not sort or sort[0] == _lib.XGM_SORT_RELEVANCE_VALUE:
                raise Unsupported("one MatchSpy, under a sort the value leads")
            vals = db.column_values(spies[0].slot)
            rows, hdr, counts = search_sorted_spy(db, p, sort[0], sort[1], sort[2], spies[0].slot, len(vals))
            spies[0]._add(hdr.matches_exact, {vals[o - 1]: n for o, n in enumerate(counts) if o and n})
            rows = [r + (0, 0) for r in rows]
        else:
            rows, hdr = search_sorted(db, p, *sort)
            rows = [r + (0, 0) for r in rows]
        hits = []
        for d, w, m, _, _, _ in rows:
            h = _lib.Hit()
            h.docid, h.weight, h.subqs_matched = d, w, m
            hits.append(h)
        mset = MSet(p.first, hits, hdr, p.total_subqs if self._query.op == "TREE" else self._query.total_subqs())
        svals = db.column_values(sort[1]) if sort else None
        cvals = db.column_values(collapse[0]) if collapse else None
        for item, (_, _, _, o, co, cc) in zip(mset._items, rows[p.first:]):
            item.sort_key = svals[o - 1] if (svals and o) else b""
            item.collapse_key = cvals[co - 1] if (cvals and co) else b""
            item.collapse_count = cc
        return mset

    def get_mset(self, first, maxitems, check_at_least=0):
        """Enquire::get_mset (reference src/xapian/api/enquire.cc:237, 396-470)."""
        if self._query is None or self._query.empty():
            return MSet(first, [], _lib.ResultHdr(), 0)
        p = plan(self._db, self._query, first, maxitems, check_at_least, self._weight)
        sort, collapse, spies = getattr(self, "_sort", None), getattr(self, "_collapse", None), getattr(self, "_spies", [])
        if sort or collapse or spies:
            return self._get_mset_by_value(p, sort, collapse, spies)
        (hits, hdr), = search_batch(self._db, [p])
        return MSet(p.first, hits, hdr, p.total_subqs if self._query.op == "TREE" else self._query.total_subqs(), plan=p)


def merged_stats(dbs, query):
    """What Enquire::add_prepared_mset accumulates over the shards (enquire.cc:385-394,
    weightinternal.cc:55-71): Σ total_length, Σ doccount, Σ termfreq per query term."""
    gs = _lib.GlobalStats()
    gs.total_length = sum(db.get_total_length() for db in dbs)
    gs.collection_size = sum(db.get_doccount() for db in dbs)
    gs.full_db_has_positions = 1 if any(db.has_positions() for db in dbs) else 0
    for i, t in enumerate(query.terms):
        gs.termfreq[i] = sum(db.get_termfreq(t) for db in dbs)
    return gs


def search_sharded(dbs, queries, first, maxitems, check_at_least=0, weight=None):
    """xgm_search_sharded: the whole per-shard protocol (merged statistics, per-shard plans and
    searches, unshard + merge on the device) in ONE C call, for a batch of queries → [MSet]."""
    weight = weight or BM25Weight()
    nq = len(queries)
    if nq == 0:
        return []
    descs = (_lib.QueryDesc * nq)()
    keep = []
    for i, q in enumerate(queries):
        d = _desc(q, first, maxitems, check_at_least, weight)
        keep.append(d)
        descs[i] = d
    k_stride = max(1, first + maxitems)
    handles = (C.c_void_p * len(dbs))(*[db._h for db in dbs])
    hits = (_lib.Hit * (nq * k_stride))()
    hdrs = (_lib.ResultHdr * nq)()
    _lib.check(_lib.lib().xgm_search_sharded(handles, len(dbs), descs, nq, k_stride, hits, hdrs))
    return [MSet(first, [hits[i * k_stride + j] for j in range(hdrs[i].n_hits)], hdrs[i], queries[i].total_subqs()) for i in range(nq)]


def get_mset_sharded(dbs, query, first, maxitems, check_at_least=0, weight=None):
    """Xapiand's per-shard protocol (reference src/database/handler.cc:1485-1549) with every shard
    on the local device(s): merged stats → per-shard search for first+maxitems → unshard + merge
    (host side here; the multi-GPU path does the same merge on device after the RCCL all-gather)."""
    gs = merged_stats(dbs, query)
    per = []
    tree_subqs = 0
    for db in dbs:
        p = plan(db, query, 0, first + maxitems, check_at_least, weight, gs)
        tree_subqs = max(tree_subqs, p.total_subqs)          # (the matcher takes the largest answer amongst the shards, matcher.cc:385-388)
        (hits, hdr), = search_batch(db, [p])
        per.append((hits, hdr))
    n_shards = len(dbs)
    allhits = []
    hdr = _lib.ResultHdr()
    for s, (hits, h) in enumerate(per):
        for x in hits:
            g = _lib.Hit((x.docid - 1) * n_shards + s + 1, x.subqs_matched, x.weight)
            allhits.append(g)
        LB = _lib.XGM_MATCHES_LOWER_BOUND
        hdr.matches_exact = ((hdr.matches_exact & ~LB) + (h.matches_exact & ~LB)) | ((hdr.matches_exact | h.matches_exact) & LB)
        hdr.max_possible = max(hdr.max_possible, h.max_possible)
        if h.max_attained > hdr.max_attained:
            hdr.max_attained = h.max_attained
            hdr.max_weight_subqs_matched = h.max_weight_subqs_matched
    allhits.sort(key=lambda x: (-x.weight, x.docid))
    allhits = allhits[: first + maxitems]
    hdr.n_hits = len(allhits)
    return MSet(first, allhits, hdr, tree_subqs if query.op == "TREE" else query.total_subqs())
