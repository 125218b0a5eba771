/* Host-side query planner: xgm_query_desc → xgm_query.
 *
 * Restates, for the query shapes the device path accepts, what the reference does between
 * Enquire::get_mset and the first PostList::next():
 *   - Enquire::Internal::get_mset clamps first/maxitems to the doccount (api/enquire.cc:419-426);
 *   - Weight::init_ + BM25Weight::init compute the per-term weight and len_factor from the MERGED
 *     collection statistics (weight/weight.cc:60-84, weight/bm25weight.cc:46-130);
 *   - AndContext::postlist builds a MultiAndPostList whose children are copied in ascending
 *     SHARD-LOCAL termfreq order with std::partial_sort_copy (matcher/multiandpostlist.h:117-130);
 *     its get_weight() sums left to right starting from 0.0 (multiandpostlist.cc:150-160);
 *   - OrContext::postlist builds a Huffman-shaped tree of binary OrPostLists with the heap of
 *     common/heap.h (api/queryinternal.cc:440-489); a node's weight is l + r (orpostlist.cc:94-103);
 *   - QueryWindowed::postlist_windowed wraps the AND in an ExactPhrase/PhrasePostList, or degrades to
 *     plain AND when no shard has positions (api/queryinternal.cc:2300-2354, 580-601).
 */
#include <algorithm>
#include <cmath>
#include <cstring>

#include "xgm_internal.h"

namespace {

struct Leaf {
    uint32_t tf;      /* shard-local termfreq = LeafPostList::get_termfreq_est (leafpostlist.cc:50-54) */
    uint32_t idx;     /* index in query order */
};

/* The comparator the reference hands to std::partial_sort_copy (multiandpostlist.h:35-40). */
struct TfAscending {
    bool operator()(const Leaf& a, const Leaf& b) const { return a.tf < b.tf; }
};

/* Binary heap with the sift-down of the reference's common/heap.h (the libc++ algorithm):
 * comp(a, b) = a.tf > b.tf, i.e. the smallest termfreq is on top
 * (ComparePostListTermFreqAscending, queryinternal.cc:140-147).  Ties matter for the tree shape, so
 * the child selection and the loop condition follow heap.h:145-199 exactly. */
struct HeapItem {
    uint64_t tf;
    int node;         /* >= 0: node id in the tree being built */
};
inline bool heap_comp(const HeapItem& a, const HeapItem& b) { return a.tf > b.tf; }

void heap_sift_down(std::vector<HeapItem>& h, size_t len, size_t start) {
    if (len < 2 || (len - 2) / 2 < start) return;
    size_t child = 2 * start + 1;
    if (child + 1 < len && heap_comp(h[child], h[child + 1])) ++child;
    if (heap_comp(h[child], h[start])) return;
    HeapItem top = h[start];
    do {
        h[start] = h[child];
        start = child;
        if ((len - 2) / 2 < child) break;
        child = 2 * child + 1;
        if (child + 1 < len && heap_comp(h[child], h[child + 1])) ++child;
    } while (!heap_comp(h[child], top));
    h[start] = top;
}

void heap_make(std::vector<HeapItem>& h) {
    size_t n = h.size();
    if (n > 1)
        for (ptrdiff_t s = (ptrdiff_t)((n - 2) / 2); s >= 0; --s) heap_sift_down(h, n, (size_t)s);
}

void heap_pop(std::vector<HeapItem>& h) {       /* Heap::pop then pop_back */
    size_t len = h.size();
    if (len > 1) {
        std::swap(h[0], h[len - 1]);
        heap_sift_down(h, len - 1, 0);
    }
    h.pop_back();
}

struct TreeNode { int l, r; };   /* children: < n_leaves → leaf (plan index), else inner node id */

void emit_postorder(const std::vector<TreeNode>& nodes, int n_leaves, int id, xgm_query* q) {
    if (id < n_leaves) {
        q->sum_prog[q->sum_len++] = (int8_t)id;
        return;
    }
    const TreeNode& nd = nodes[(size_t)(id - n_leaves)];
    emit_postorder(nodes, n_leaves, nd.l, q);
    emit_postorder(nodes, n_leaves, nd.r, q);
    q->sum_prog[q->sum_len++] = XGM_SUM_ADD;
}

/* BM25Weight::init, bm25weight.cc:46-130 with rset_size == 0, factor == 1, wqf == 1. */
double bm25_termweight(uint32_t collection_size, uint32_t termfreq, double k1, double k3) {
    double tw = (collection_size - termfreq + 0.5) / (termfreq + 0.5);   /* unsigned subtraction as in the reference */
    if (tw < 2) tw = tw * 0.5 + 1;
    double termweight = std::log(tw) * 1.0;
    if (k3 != 0) {
        double wqf_double = 1;
        termweight *= (k3 + 1) * wqf_double / (k3 + wqf_double);
    }
    termweight *= (k1 + 1);
    return termweight;
}

/* BM25Weight::get_maxpart, bm25weight.cc:183-207. */
double bm25_maxpart(double termweight, double len_factor, double k1, double b, double min_normlen, uint32_t wdf_ub,
                    uint32_t doclen_lb) {
    double denom = k1;
    if (k1 != 0.0) {
        if (b != 0.0) {
            double normlen_lb = std::max(std::max(wdf_ub, doclen_lb) * len_factor, min_normlen);
            denom *= (normlen_lb * b + (1 - b));
        }
    }
    double wdf_max = wdf_ub;
    denom += wdf_max;
    return termweight * (wdf_max / denom);
}

}  // namespace

extern "C" int xgm_plan_query(const xgm_index* idx, const xgm_query_desc* d, const xgm_global_stats* gs, xgm_query* out) {
    if (!idx || !d || !out) return xgm_set_error(XGM_E_INVALID, "null argument");
    memset(out, 0, sizeof *out);
    const uint32_t n = d->n_terms;
    if (d->op < XGM_OP_AND || d->op > XGM_OP_NEAR) return XGM_UNSUPPORTED;
    if (n == 0 || n > XGM_MAX_TERMS) return XGM_UNSUPPORTED;
    /* AND_NOT / AND_MAYBE / FILTER: left = AND of the first nr terms, right = the others
     * (QueryAndNot / QueryAndMaybe / QueryFilter::postlist, api/queryinternal.cc:2208-2283) */
    const bool sided = d->op == XGM_OP_AND_NOT || d->op == XGM_OP_AND_MAYBE || d->op == XGM_OP_FILTER;
    const uint32_t nr = sided ? d->n_required : n;
    if (sided && (nr == 0 || nr >= n)) return xgm_set_error(XGM_E_INVALID, "n_required must leave terms on both sides");
    if (d->k2 != 0.0) return XGM_UNSUPPORTED;          /* would need ExtraWeightPostList (localsubmatch.cc:183-193) */
    if (!(d->k1 >= 0.0) || !(d->b >= 0.0 && d->b <= 1.0) || !(d->k3 >= 0.0) || !(d->min_normlen >= 0.0))
        return xgm_set_error(XGM_E_INVALID, "bad BM25 parameters");
    for (uint32_t i = 0; i < n; ++i) {
        if (!d->terms[i] || d->term_len[i] == 0) return XGM_UNSUPPORTED;     /* empty term = MatchAll */
        for (uint32_t j = 0; j < i; ++j)
            if (d->term_len[i] == d->term_len[j] && memcmp(d->terms[i], d->terms[j], d->term_len[i]) == 0)
                return XGM_UNSUPPORTED;                 /* repeated term: wqf merging / shared poslists */
    }

    const uint32_t collection_size = gs ? gs->collection_size : idx->hdr.doccount;
    const uint64_t total_length = gs ? gs->total_length : idx->hdr.total_length;
    const bool full_db_has_positions = gs ? (gs->full_db_has_positions != 0) : (idx->hdr.has_positions != 0);

    /* Enquire::Internal::get_mset, enquire.cc:419-426 (doccount of the database being searched:
     * in Xapiand's per-shard protocol that is the shard) */
    {
        uint32_t docs = idx->hdr.doccount;
        uint32_t first = std::min(d->first, docs);
        uint32_t maxitems = std::min(d->maxitems, docs - first);
        uint32_t cal = std::min(d->check_at_least, docs);
        cal = std::max(cal, first + maxitems);
        out->first = first;
        out->maxitems = maxitems;
        out->check_at_least = cal;
    }
    if ((uint64_t)out->first + out->maxitems > XGM_MAX_K) return XGM_UNSUPPORTED;

    out->op = d->op;
    out->n_terms = n;
    out->k1 = d->k1;
    out->b = d->b;
    out->min_normlen = d->min_normlen;
    /* BM25Weight::init tail, bm25weight.cc:117-127 + Weight::Internal::get_average_length
     * (weightinternal.h:235-241) */
    if (d->k2 == 0 && (d->b == 0 || d->k1 == 0)) {
        out->len_factor = 0;
    } else {
        double avg = collection_size == 0 ? 0.0 : (double)total_length / collection_size;
        out->len_factor = avg != 0 ? 1 / avg : 0;
    }

    uint32_t local_id[XGM_MAX_TERMS], local_tf[XGM_MAX_TERMS];
    double tw[XGM_MAX_TERMS], maxpart[XGM_MAX_TERMS];
    bool wide_needed = false;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t id = UINT32_MAX;
        xgm_lookup_term_id(idx, d->terms[i], d->term_len[i], &id);
        local_id[i] = id;
        local_tf[i] = id == UINT32_MAX ? 0u : idx->term_df[id];
        uint32_t global_tf = gs ? gs->termfreq[i] : local_tf[i];
        if (global_tf < local_tf[i] || global_tf > collection_size) return xgm_set_error(XGM_E_INVALID, "inconsistent global statistics");
        tw[i] = bm25_termweight(collection_size, global_tf, d->k1, d->k3);
        uint32_t wdf_ub = id == UINT32_MAX ? 0u : idx->term_wdfub[id];
        maxpart[i] = bm25_maxpart(tw[i], out->len_factor, d->k1, d->b, d->min_normlen, wdf_ub, idx->hdr.doclen_lower_bound);
        (void)wide_needed;
    }

    /* positional filter */
    bool phrase_active = false;
    bool shard_empty = false;
    const bool positional = d->op == XGM_OP_PHRASE || d->op == XGM_OP_NEAR;      /* QueryWindowed::postlist_windowed, queryinternal.cc:2300-2354 */
    if (positional && n > 1) {
        if (full_db_has_positions) {
            if (!idx->hdr.has_positions) {
                shard_empty = true;                    /* queryinternal.cc:2308-2318 */
            } else {
                phrase_active = true;
                for (uint32_t i = 0; i < n; ++i)
                    if (local_id[i] != UINT32_MAX && !(idx->term_flags[local_id[i]] & XGM_TF_POS_OK)) return XGM_UNSUPPORTED;
            }
        }
    }
    out->window = positional ? (d->window ? d->window : n) : 0;
    out->phrase_active = phrase_active ? 1u : 0u;
    if (phrase_active && out->window < n) return XGM_UNSUPPORTED;   /* Xapian rejects/normalises this upstream */

    /* leaf order */
    uint32_t order[XGM_MAX_TERMS];
    if (d->op == XGM_OP_OR) {
        for (uint32_t i = 0; i < n; ++i) order[i] = i;
    } else {
        /* the AND side (all terms, or the left-hand nr) in MultiAnd order; a right-hand side keeps query order */
        Leaf in[XGM_MAX_TERMS], sorted[XGM_MAX_TERMS];
        for (uint32_t i = 0; i < nr; ++i) in[i] = Leaf{local_tf[i], i};
        /* same library algorithm, same comparator shape as the reference → same tie behaviour */
        std::partial_sort_copy(in, in + nr, sorted, sorted + nr, TfAscending());
        for (uint32_t i = 0; i < nr; ++i) order[i] = sorted[i].idx;
        for (uint32_t i = nr; i < n; ++i) order[i] = i;
        if (d->op == XGM_OP_FILTER) {
            /* QueryFilter = MultiAnd{l, r x 0}: the document's weight is l's — the chain over the weighted
             * leaves in THEIR MultiAnd order.  The unweighted leaves contribute +0.0 wherever they stand
             * (x + 0.0 == x exactly for x >= +0.0), so they are merged in by termfreq: the plan is then
             * ascending in termfreq like a plain conjunction's, which is what the conjunction kernels'
             * rarest-term-first / dense-suffix layout expects. */
            uint32_t merged[XGM_MAX_TERMS], a = 0, b = nr, m = 0;
            uint32_t extra[XGM_MAX_TERMS];
            for (uint32_t i = nr; i < n; ++i) extra[i] = i;
            std::stable_sort(extra + nr, extra + n, [&](uint32_t x, uint32_t y) { return local_tf[x] < local_tf[y]; });
            while (a < nr || b < n) {
                if (b >= n || (a < nr && local_tf[order[a]] <= local_tf[extra[b]])) merged[m++] = order[a++];
                else merged[m++] = extra[b++];
            }
            for (uint32_t i = 0; i < n; ++i) order[i] = merged[i];
        }
    }
    for (uint32_t p = 0; p < n; ++p) {
        uint32_t i = order[p];
        out->terms[p].term_id = local_id[i];
        out->terms[p].phrase_index = i;
        out->terms[p].termweight = (d->op == XGM_OP_FILTER && i >= nr) ? 0.0 : tw[i];
        if (d->op == XGM_OP_FILTER && i >= nr) maxpart[i] = 0.0;
    }

    /* weight summation program + max_possible in the same association */
    std::vector<TreeNode> nodes;
    int root;
    /* Huffman-shaped OrPostList tree over plan positions [lo, n) (OrContext::postlist); returns its root */
    auto or_tree = [&](uint32_t lo) -> int {
        if (n - lo == 1) return (int)lo;
        std::vector<HeapItem> heap;
        for (uint32_t p = lo; p < n; ++p) heap.push_back(HeapItem{local_tf[order[p]], (int)p});
        heap_make(heap);
        while (true) {
            HeapItem r = heap.front();
            heap_pop(heap);
            HeapItem l = heap.front();
            nodes.push_back(TreeNode{l.node, r.node});
            int id = (int)n + (int)nodes.size() - 1;
            if (heap.size() == 1) return id;
            heap[0].node = id;
            heap[0].tf = l.tf + r.tf;
            heap_sift_down(heap, heap.size(), 0);      /* Heap::replace */
        }
    };
    if (d->op == XGM_OP_OR && n > 1) {
        root = or_tree(0);
    } else {
        /* MultiAndPostList::get_weight: ((0 + w0) + w1) + ...; 0 + w0 == w0 exactly.  Only the AND side
         * carries weight for AND_NOT / FILTER; AND_MAYBE adds the right-hand OR tree where it matches
         * (AndMaybePostList::get_weight, andmaybepostlist.cc:57-64). */
        root = 0;
        const uint32_t chain = d->op == XGM_OP_FILTER ? n : nr;     /* FILTER: the unweighted leaves ride along as +0.0 */
        for (uint32_t p = 1; p < chain; ++p) {
            nodes.push_back(TreeNode{root, (int)p});
            root = (int)n + (int)nodes.size() - 1;
        }
        if (d->op == XGM_OP_AND_MAYBE) {
            int r_root = or_tree(nr);
            nodes.push_back(TreeNode{root, r_root});
            root = (int)n + (int)nodes.size() - 1;
        }
    }
    out->sum_len = 0;
    emit_postorder(nodes, (int)n, root, out);

    {
        double val[2 * XGM_MAX_TERMS];
        for (uint32_t p = 0; p < n; ++p) val[p] = maxpart[order[p]];
        for (size_t j = 0; j < nodes.size(); ++j) val[n + j] = val[nodes[j].l] + val[nodes[j].r];
        double mp = val[root];
        if (d->op != XGM_OP_OR && n > 1) {
            /* MultiAndPostList::recalc_maxweight starts from 0.0 (multiandpostlist.cc:169-179): same value */
        }
        out->max_possible = mp;
    }

    /* who must / must not index a matching document */
    out->req_mask = 0; out->neg_mask = 0;
    for (uint32_t p = 0; p < n; ++p) {
        if (d->op == XGM_OP_OR) break;
        if (d->op == XGM_OP_FILTER || p < nr) out->req_mask |= 1u << p;
        else if (d->op == XGM_OP_AND_NOT) out->neg_mask |= 1u << p;
    }
    bool any_absent = false, all_absent = true;
    for (uint32_t p = 0; p < n; ++p) {
        if (out->terms[p].term_id == UINT32_MAX) { if (d->op == XGM_OP_OR || ((out->req_mask >> p) & 1u)) any_absent = true; }
        else all_absent = false;
    }
    if (d->op == XGM_OP_OR ? all_absent : any_absent) shard_empty = true;
    if (shard_empty) {
        /* mark: n_terms stays, every leaf absent → the kernel exits at once */
        for (uint32_t p = 0; p < n; ++p) out->terms[p].term_id = UINT32_MAX;
    }
    if (phrase_active && n > 8) return XGM_UNSUPPORTED;   /* XGM_PHRASE_MAX_TERMS: per-lane cursors of the positional filter */
    return XGM_OK;
}
