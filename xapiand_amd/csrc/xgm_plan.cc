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

/* BM25Weight::init, bm25weight.cc:46-130 with rset_size == 0. */
double bm25_termweight(uint32_t collection_size, uint32_t termfreq, double k1, double k3, uint32_t wqf = 1, double factor = 1.0) {
    double tw = (collection_size - termfreq + 0.5) / (termfreq + 0.5);   /* unsigned subtraction as in the reference */
    if (tw < 2) tw = tw * 0.5 + 1;
    double termweight = std::log(tw) * factor;
    if (k3 != 0) {
        double wqf_double = wqf;
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


/* ---- match-count bounds and estimate -------------------------------------------------------------------
 * PostList::get_termfreq_min / _est / _max of the tree the reference would build, from the shard's own termfreqs —
 * what Matcher::get_local_mset hands to ProtoMSet::finalise (matcher.cc:432-434, protomset.h:484-619):
 * leafpostlist.cc:51-54, multiandpostlist.cc:55-105, orpostlist.cc:80-83, 353-384, andnotpostlist.cc:30-61,
 * boolorpostlist.cc:44-52, 158-190, AndMaybe = its left side, Exact phrase / Phrase / Near = the AND's estimate / 4, / 3, / 2
 * with a lower bound of 0 (exactphrasepostlist.cc:149-160, phrasepostlist.cc:108-114, nearpostlist.cc:203-210). */
struct Est { uint32_t mn, est, mx; };

Est est_leaf(uint32_t tf) { return Est{tf, tf, tf}; }

Est est_mand(const std::vector<Est>& k, uint32_t db_size) {          /* kids in MultiAnd (plist) order */
    Est r;
    uint32_t sum = k[0].mn;
    if (sum) {
        for (size_t i = 1; i < k.size(); ++i) {
            const uint32_t sum_old = sum;
            sum += k[i].mn;
            if (sum >= sum_old && sum <= db_size) { sum = 0; break; }
            sum -= db_size;
        }
    }
    r.mn = sum;
    r.mx = k[0].mx;
    for (size_t i = 1; i < k.size(); ++i) r.mx = std::min(r.mx, k[i].mx);
    double e = k[0].est;
    for (size_t i = 1; i < k.size(); ++i) e = (e * k[i].est) / db_size;
    r.est = db_size ? (uint32_t)(e + 0.5) : 0u;
    return r;
}

Est est_or(const Est& l, const Est& r, uint32_t db_size) {
    Est o;
    o.mn = std::max(l.mn, r.mn);
    uint32_t mx = l.mx + r.mx;
    if (mx > db_size || mx < l.mx) mx = db_size;
    o.mx = mx;
    const double a = l.est, b = r.est, n = db_size;
    o.est = n == 0.0 ? 0u : (uint32_t)(a + b - (a * b / n) + 0.5);
    return o;
}

Est est_andnot(const Est& l, const Est& r, uint32_t db_size) {
    Est o;
    o.mn = l.mn <= r.mx ? 0u : l.mn - r.mx;
    o.mx = std::min(db_size - r.mn, l.mx);
    double e = l.est;
    e = (e * (db_size - (double)r.est)) / db_size;
    o.est = db_size ? (uint32_t)(e + 0.5) : 0u;
    return o;
}

Est est_boolor(const std::vector<Est>& k, uint32_t db_size) {        /* an OP_SYNONYM's BoolOrPostList, kids in query order */
    Est o;
    o.mn = k[0].mn;
    for (size_t i = 1; i < k.size(); ++i) o.mn = std::max(o.mn, k[i].mn);
    uint32_t mx = k[0].mx;
    bool capped = false;
    for (size_t i = 1; i < k.size() && !capped; ++i) {
        const uint32_t old = mx;
        mx += k[i].mx;
        if (mx >= db_size || mx < old) { mx = db_size; capped = true; }
    }
    o.mx = mx;
    if (!db_size) { o.est = 0; return o; }
    const double scale = 1.0 / db_size;
    double P = k[0].est * scale;
    for (size_t i = 1; i < k.size(); ++i) { const double Pi = k[i].est * scale; P += Pi - P * Pi; }
    o.est = (uint32_t)(P * db_size + 0.5);
    return o;
}

/* the Huffman-shaped OrPostList tree over `leaves` (OrContext::postlist): its estimate triple */
Est est_or_tree(const std::vector<Est>& leaves, uint32_t db_size) {
    if (leaves.size() == 1) return leaves[0];
    std::vector<Est> val(leaves);
    std::vector<HeapItem> heap;
    for (size_t i = 0; i < leaves.size(); ++i) heap.push_back(HeapItem{leaves[i].est, (int)i});
    heap_make(heap);
    while (true) {
        HeapItem r = heap.front();
        heap_pop(heap);
        HeapItem l = heap.front();
        val.push_back(est_or(val[(size_t)l.node], val[(size_t)r.node], db_size));
        const int id = (int)val.size() - 1;
        if (heap.size() == 1) return val[(size_t)id];
        heap[0].node = id;
        heap[0].tf = l.tf + r.tf;
        heap_sift_down(heap, heap.size(), 0);
    }
}

/* ---- nested queries (XGM_OP_TREE) ------------------------------------------------------------------------------------
 * The reference's Query → PostList lowering restated for trees (api/queryinternal.cc): QueryTerm / QueryScaleWeight
 * (:1049-1080: the factor multiplies down into BM25Weight::init), QueryAndLike::postlist_sub_and_like (:2083-2103: nested
 * ANDs and FILTERs flatten into ONE MultiAndPostList, children in ascending termfreq ESTIMATE, multiandpostlist.h:117-130),
 * do_or_like (:1790-1820: nested ORs flatten into one Huffman tree, :440-489), QueryAndNot / QueryAndMaybe / QueryFilter
 * (:2208-2283), do_synonym + LocalSubMatch::make_synonym_postlist (:1822-1898, matcher/localsubmatch.cc:199-229).
 * Estimates: leafpostlist.cc:51-54, orpostlist.cc:365-384, multiandpostlist.cc:92-105, andnotpostlist.cc:50-61,
 * boolorpostlist.cc:175-230. */
struct ANode { int kind; std::vector<int> kids; int term = -1; double scale = 1.0; };
enum { P_GROUP = 0, P_MAND, P_OR, P_ANDNOT, P_MAYBE };
struct PNode { int type; std::vector<int> kids; int group = -1; uint32_t est = 0; double maxw = 0.0; int dev = -1; /* operand id on the device */ Est e{0, 0, 0}; };

struct TreePlanner {
    const xgm_index* idx; const xgm_query_desc* d; const xgm_global_stats* gs; xgm_query* out;
    std::vector<ANode> ast; std::vector<PNode> pl;
    uint32_t N = 0, db_size = 0; double len_factor = 0;
    uint32_t local_id[XGM_MAX_TERMS], tf_local[XGM_MAX_TERMS], tf_global[XGM_MAX_TERMS];
    bool used[XGM_MAX_TERMS] = {};
    int rc = XGM_OK;

    int add(PNode&& n) { pl.push_back(std::move(n)); return (int)pl.size() - 1; }
    int new_group(double weight, bool scored, double maxw, uint32_t est) {
        if (out->n_groups >= XGM_MAX_TERMS) { rc = XGM_UNSUPPORTED; return -1; }
        const uint32_t g = out->n_groups++;
        out->group_weight[g] = weight;
        if (scored) { out->group_scored |= 1u << g; ++out->total_subqs; }
        PNode n; n.type = P_GROUP; n.group = (int)g; n.est = est; n.maxw = maxw; n.e = Est{est, est, est};
        return add(std::move(n));
    }
    int leaf(int t, double factor) {
        const bool weighted = factor != 0.0;
        const double w = bm25_termweight(N, tf_global[t], d->k1, d->k3, d->wqf[t] ? d->wqf[t] : 1u, factor);
        const uint32_t wdf_ub = local_id[t] == UINT32_MAX ? 0u : idx->term_wdfub[local_id[t]];
        const int x = new_group(weighted ? w : 0.0, weighted, weighted ? bm25_maxpart(w, len_factor, d->k1, d->b, d->min_normlen, wdf_ub, idx->hdr.doclen_lower_bound) : 0.0, tf_local[t]);
        if (x >= 0) out->group_of[t] = (uint8_t)pl[x].group;
        return x;
    }
    int synonym(const std::vector<int>& terms, double factor) {
        const bool weighted = factor != 0.0;
        uint32_t est = 0, tf_syn = 0;
        if (db_size) {
            const double scale = 1.0 / db_size;
            double P = tf_local[terms[0]] * scale;
            for (size_t i = 1; i < terms.size(); ++i) { const double Pi = tf_local[terms[i]] * scale; P += Pi - P * Pi; }
            est = (uint32_t)(P * db_size + 0.5);
        }
        if (N) {
            const double scale = 1.0 / N;
            double P = tf_global[terms[0]] * scale;
            for (size_t i = 1; i < terms.size(); ++i) { const double Pi = tf_global[terms[i]] * scale; P += Pi - P * Pi; }
            tf_syn = (uint32_t)(P * N + 0.5);
        }
        const double w = bm25_termweight(N, tf_syn, d->k1, d->k3, 1u, factor);
        /* Weight::init_ (synonym case, weight.cc:86-115): the wdf bound of a synonym is the doclength upper bound */
        const int x = new_group(weighted ? w : 0.0, weighted,
                                weighted ? bm25_maxpart(w, len_factor, d->k1, d->b, d->min_normlen, idx->hdr.doclen_upper_bound, idx->hdr.doclen_lower_bound) : 0.0, est);
        if (x >= 0) {
            for (int t : terms) out->group_of[t] = (uint8_t)pl[x].group;
            std::vector<Est> k;
            for (int t : terms) k.push_back(est_leaf(tf_local[t]));
            pl[x].e = est_boolor(k, db_size);
        }
        return x;
    }
    int mand(const std::vector<int>& ctx) {
        PNode n; n.type = P_MAND;
        std::vector<Leaf> in(ctx.size()), sorted(ctx.size());
        for (size_t i = 0; i < ctx.size(); ++i) in[i] = Leaf{pl[ctx[i]].est, (uint32_t)i};
        std::partial_sort_copy(in.begin(), in.end(), sorted.begin(), sorted.end(), TfAscending());
        for (const Leaf& l : sorted) n.kids.push_back(ctx[l.idx]);
        double r = pl[n.kids[0]].est;
        for (size_t i = 1; i < n.kids.size(); ++i) r = (r * pl[n.kids[i]].est) / db_size;
        n.est = db_size ? (uint32_t)(r + 0.5) : 0u;
        double m = 0.0;
        for (int k : n.kids) m += pl[k].maxw;
        n.maxw = m;
        std::vector<Est> ke;
        for (int k : n.kids) ke.push_back(pl[k].e);
        n.e = est_mand(ke, db_size);
        return add(std::move(n));
    }
    int or2(int l, int r) {
        PNode n; n.type = P_OR; n.kids = {l, r};
        n.e = est_or(pl[l].e, pl[r].e, db_size);
        n.est = n.e.est;
        n.maxw = pl[l].maxw + pl[r].maxw;
        return add(std::move(n));
    }
    int or_tree(const std::vector<int>& ctx) {
        if (ctx.empty()) return -1;
        if (ctx.size() == 1) return ctx[0];
        std::vector<HeapItem> heap;
        for (int c : ctx) heap.push_back(HeapItem{pl[c].est, c});
        heap_make(heap);
        while (true) {
            HeapItem r = heap.front();
            heap_pop(heap);
            HeapItem l = heap.front();
            const int id = or2(l.node, r.node);
            if (heap.size() == 1) return id;
            heap[0].node = id;
            heap[0].tf = l.tf + r.tf;
            heap_sift_down(heap, heap.size(), 0);
        }
    }
    void sub_and_like(int a, std::vector<int>& ctx, double factor) {
        const ANode& n = ast[a];
        if (n.kind == XGM_T_AND) { for (int k : n.kids) sub_and_like(k, ctx, factor); return; }
        if (n.kind == XGM_T_FILTER) { for (int k : n.kids) { sub_and_like(k, ctx, factor); factor = 0.0; } return; }
        ctx.push_back(postlist(a, factor));
    }
    void sub_or_like(int a, std::vector<int>& ctx, double factor) {
        const ANode& n = ast[a];
        if (n.kind == XGM_T_OR) { for (int k : n.kids) sub_or_like(k, ctx, factor); return; }
        ctx.push_back(postlist(a, factor));
    }
    int postlist(int a, double factor) {
        if (rc) return -1;
        const ANode& n = ast[a];
        switch (n.kind) {
        case XGM_T_TERM: return leaf(n.term, factor);
        case XGM_T_SCALE: return postlist(n.kids[0], factor * n.scale);
        case XGM_T_AND: { std::vector<int> ctx; sub_and_like(a, ctx, factor); return rc ? -1 : mand(ctx); }
        case XGM_T_FILTER: { const int l = postlist(n.kids[0], factor); const int r = postlist(n.kids[1], 0.0); return rc ? -1 : mand({l, r}); }
        case XGM_T_OR: { std::vector<int> ctx; sub_or_like(a, ctx, factor); return rc ? -1 : or_tree(ctx); }
        case XGM_T_AND_NOT: {
            const int l = postlist(n.kids[0], factor);
            std::vector<int> ctx;
            for (size_t i = 1; i < n.kids.size(); ++i) sub_or_like(n.kids[i], ctx, 0.0);
            if (rc) return -1;
            const int r = or_tree(ctx);
            PNode p; p.type = P_ANDNOT; p.kids = {l, r};
            double e = pl[l].est;
            e = (e * (db_size - (double)pl[r].est)) / db_size;
            p.est = db_size ? (uint32_t)(e + 0.5) : 0u;
            p.maxw = pl[l].maxw;
            p.e = est_andnot(pl[l].e, pl[r].e, db_size);
            return add(std::move(p));
        }
        case XGM_T_AND_MAYBE: {
            const int l = postlist(n.kids[0], factor);
            if (factor == 0.0) return l;
            std::vector<int> ctx;
            for (size_t i = 1; i < n.kids.size(); ++i) sub_or_like(n.kids[i], ctx, factor);
            if (rc) return -1;
            const int r = or_tree(ctx);
            PNode p; p.type = P_MAYBE; p.kids = {l, r}; p.est = pl[l].est; p.maxw = pl[l].maxw + pl[r].maxw; p.e = pl[l].e;
            return add(std::move(p));
        }
        case XGM_T_SYNONYM: {
            std::vector<int> terms;
            for (int k : n.kids) terms.push_back(ast[k].term);
            if (terms.size() == 1) return leaf(terms[0], factor);          /* QuerySynonym::done */
            return synonym(terms, factor);
        }
        case XGM_T_WILDCARD: {                                             /* QueryWildcard::postlist, OP_SYNONYM combiner: always a SynonymPostList */
            std::vector<int> terms;
            for (int k : n.kids) terms.push_back(ast[k].term);
            return synonym(terms, factor);
        }
        case XGM_T_WILDCARD_OR: {                                          /* ... OP_OR combiner: an OrContext of its own */
            std::vector<int> ctx;
            for (int k : n.kids) ctx.push_back(leaf(ast[k].term, factor));
            return rc ? -1 : or_tree(ctx);
        }
        }
        rc = XGM_UNSUPPORTED;
        return -1;
    }
    /* lowered tree → the device's binary nodes, children first; returns the operand id */
    int emit(int x) {
        PNode& n = pl[x];
        if (n.dev >= 0) return n.dev;
        if (n.type == P_GROUP) return n.dev = n.group;
        auto node = [&](int op, int a, int b) {
            if (out->tree_len >= XGM_MAX_TREE) { rc = XGM_UNSUPPORTED; return 0; }
            const uint32_t j = out->tree_len++;
            out->tree_op[j] = (uint8_t)op; out->tree_a[j] = (uint8_t)a; out->tree_b[j] = (uint8_t)b;
            return (int)(XGM_MAX_TERMS + j);          /* provisional id: nodes are renumbered to n_groups + j at the end */
        };
        if (n.type == P_MAND) {
            int acc = emit(n.kids[0]);                /* 0.0 + w0 == w0 exactly */
            for (size_t i = 1; i < n.kids.size(); ++i) acc = node(XGM_N_AND, acc, emit(n.kids[i]));
            return n.dev = acc;
        }
        const int a = emit(n.kids[0]), b = emit(n.kids[1]);
        return n.dev = node(n.type == P_OR ? XGM_N_OR : n.type == P_ANDNOT ? XGM_N_ANDNOT : XGM_N_MAYBE, a, b);
    }
};

int plan_tree(const xgm_index* idx, const xgm_query_desc* d, const xgm_global_stats* gs, xgm_query* out) {
    const uint32_t n = d->n_terms;
    if (n == 0 || n > XGM_MAX_TERMS || d->n_tree == 0 || d->n_tree > XGM_MAX_TREE) return XGM_UNSUPPORTED;
    TreePlanner tp;
    tp.idx = idx; tp.d = d; tp.gs = gs; tp.out = out;
    /* post-order program → AST (QueryAndLike / OrLike::done: one subquery is that subquery) */
    std::vector<int> stack;
    for (uint32_t i = 0; i < d->n_tree; ++i) {
        ANode nd; nd.kind = d->tree[i].kind;
        if (nd.kind == XGM_T_TERM) {
            if (d->tree[i].term >= n || tp.used[d->tree[i].term]) return XGM_UNSUPPORTED;     /* a term twice: wqf merging / shared postlists */
            tp.used[d->tree[i].term] = true;
            nd.term = d->tree[i].term;
        } else {
            const uint32_t ar = nd.kind == XGM_T_SCALE ? 1u : d->tree[i].arity;
            if (nd.kind > XGM_T_WILDCARD_OR || ar == 0 || stack.size() < ar) return xgm_set_error(XGM_E_INVALID, "malformed query tree");
            if (nd.kind == XGM_T_FILTER && ar != 2) return xgm_set_error(XGM_E_INVALID, "FILTER takes two subqueries");
            if ((nd.kind == XGM_T_AND_NOT || nd.kind == XGM_T_AND_MAYBE) && ar < 2) return xgm_set_error(XGM_E_INVALID, "AND_NOT / AND_MAYBE take a left and a right side");
            nd.kids.assign(stack.end() - ar, stack.end());
            stack.resize(stack.size() - ar);
            if (nd.kind == XGM_T_SCALE) { nd.scale = d->tree_scale[i]; if (!(nd.scale > 0.0)) return XGM_UNSUPPORTED; }   /* a zero scale makes the subtree boolean */
            if (nd.kind == XGM_T_SYNONYM || nd.kind == XGM_T_WILDCARD || nd.kind == XGM_T_WILDCARD_OR)
                for (int k : nd.kids) if (tp.ast[k].kind != XGM_T_TERM) return XGM_UNSUPPORTED;
        }
        tp.ast.push_back(nd);
        stack.push_back((int)tp.ast.size() - 1);
        if ((nd.kind == XGM_T_AND || nd.kind == XGM_T_OR) && nd.kids.size() == 1) stack.back() = nd.kids[0];
    }
    if (stack.size() != 1) return xgm_set_error(XGM_E_INVALID, "malformed query tree");
    for (uint32_t i = 0; i < n; ++i) {
        if (!d->terms[i] || d->term_len[i] == 0 || !tp.used[i]) return XGM_UNSUPPORTED;
        for (uint32_t j = 0; j < i; ++j)
            if (d->term_len[i] == d->term_len[j] && memcmp(d->terms[i], d->terms[j], d->term_len[i]) == 0) return XGM_UNSUPPORTED;
    }
    tp.N = gs ? gs->collection_size : idx->hdr.doccount;
    tp.db_size = idx->hdr.doccount;
    tp.len_factor = out->len_factor;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t id = UINT32_MAX;
        xgm_lookup_term_id(idx, d->terms[i], d->term_len[i], &id);
        tp.local_id[i] = id;
        tp.tf_local[i] = id == UINT32_MAX ? 0u : idx->term_df[id];
        tp.tf_global[i] = gs ? gs->termfreq[i] : tp.tf_local[i];
        if (tp.tf_global[i] < tp.tf_local[i] || tp.tf_global[i] > tp.N) return xgm_set_error(XGM_E_INVALID, "inconsistent global statistics");
        out->terms[i].term_id = id;
        out->terms[i].phrase_index = i;
    }
    out->op = XGM_OP_TREE;
    out->n_terms = n;
    const int root = tp.postlist(stack[0], 1.0);
    if (tp.rc) return tp.rc;
    const int dev_root = tp.emit(root);
    if (tp.rc) return tp.rc;
    /* renumber the provisional node ids now that the number of groups is known */
    const uint32_t G = out->n_groups;
    auto fix = [&](int x) { return (uint8_t)(x >= (int)XGM_MAX_TERMS ? G + (uint32_t)(x - (int)XGM_MAX_TERMS) : (uint32_t)x); };
    for (uint32_t j = 0; j < out->tree_len; ++j) { out->tree_a[j] = fix(out->tree_a[j]); out->tree_b[j] = fix(out->tree_b[j]); }
    out->tree_root = fix(dev_root);
    out->max_possible = tp.pl[root].maxw;
    out->est_min = tp.pl[root].e.mn; out->est_est = tp.pl[root].e.est; out->est_max = tp.pl[root].e.mx;
    for (uint32_t i = 0; i < n; ++i) out->terms[i].termweight = out->group_weight[out->group_of[i]];
    return XGM_OK;
}

}  // namespace

extern "C" int xgm_plan_query(const xgm_index* idx, const xgm_query_desc* d, const xgm_global_stats* gs, xgm_query* out) {
    if (!idx || !d || !out) return xgm_set_error(XGM_E_INVALID, "null argument");
    memset(out, 0, sizeof *out);
    out->replay = d->replay;
    const uint32_t n = d->n_terms;
    if (d->op < XGM_OP_AND || d->op > XGM_OP_TREE) return XGM_UNSUPPORTED;
    if (n == 0 || n > XGM_MAX_TERMS) return XGM_UNSUPPORTED;
    const bool is_tree = d->op == XGM_OP_TREE;
    /* AND_NOT / AND_MAYBE / FILTER: left = AND of the first nr terms, right = the others
     * (QueryAndNot / QueryAndMaybe / QueryFilter::postlist, api/queryinternal.cc:2208-2283) */
    const bool sided = d->op == XGM_OP_AND_NOT || d->op == XGM_OP_AND_MAYBE || d->op == XGM_OP_FILTER;
    const uint32_t nr = sided ? d->n_required : n;
    if (sided && (nr == 0 || nr >= n)) return xgm_set_error(XGM_E_INVALID, "n_required must leave terms on both sides");
    if (d->k2 != 0.0) return XGM_UNSUPPORTED;          /* would need ExtraWeightPostList (localsubmatch.cc:183-193) */
    if (!(d->k1 >= 0.0) || !(d->b >= 0.0 && d->b <= 1.0) || !(d->k3 >= 0.0) || !(d->min_normlen >= 0.0))
        return xgm_set_error(XGM_E_INVALID, "bad BM25 parameters");
    for (uint32_t i = 0; i < n && !is_tree; ++i) {
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

    if (is_tree) return plan_tree(idx, d, gs, out);

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
        tw[i] = bm25_termweight(collection_size, global_tf, d->k1, d->k3, d->wqf[i] ? d->wqf[i] : 1u);
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

    /* match-count bounds / estimate of the tree the reference builds for this shape */
    {
        const uint32_t db = idx->hdr.doccount;
        auto leaf_q = [&](uint32_t i) { return est_leaf(local_tf[i]); };           /* by QUERY index */
        auto and_of = [&](uint32_t lo, uint32_t hi) {                              /* MultiAnd over plan positions [lo, hi) */
            std::vector<Est> k;
            for (uint32_t p = lo; p < hi; ++p) k.push_back(leaf_q(order[p]));
            return k.size() == 1 ? k[0] : est_mand(k, db);
        };
        auto or_of = [&](uint32_t lo, uint32_t hi) {                               /* Huffman OR over query positions [lo, hi) */
            std::vector<Est> k;
            for (uint32_t i = lo; i < hi; ++i) k.push_back(leaf_q(i));
            return est_or_tree(k, db);
        };
        Est e;
        if (d->op == XGM_OP_OR) e = or_of(0, n);
        else if (d->op == XGM_OP_AND_NOT) e = est_andnot(and_of(0, nr), or_of(nr, n), db);
        else if (d->op == XGM_OP_AND_MAYBE) e = and_of(0, nr);
        else if (d->op == XGM_OP_FILTER) {
            /* QueryFilter::postlist: MultiAnd{l, r}, l and r being whole postlists (an inner MultiAnd each when they are ANDs) */
            std::vector<Leaf> in, sorted;
            std::vector<Est> sides;
            {
                std::vector<Est> k;
                for (uint32_t p = 0; p < n; ++p) if (order[p] < nr) k.push_back(leaf_q(order[p]));      /* plan order restricted to the left side = its MultiAnd order */
                sides.push_back(k.size() == 1 ? k[0] : est_mand(k, db));
                std::vector<Leaf> rin, rs;
                for (uint32_t i = nr; i < n; ++i) rin.push_back(Leaf{local_tf[i], i});
                rs.resize(rin.size());
                std::partial_sort_copy(rin.begin(), rin.end(), rs.begin(), rs.end(), TfAscending());
                std::vector<Est> rk;
                for (const Leaf& l : rs) rk.push_back(leaf_q(l.idx));
                sides.push_back(rk.size() == 1 ? rk[0] : est_mand(rk, db));
            }
            in = {Leaf{sides[0].est, 0}, Leaf{sides[1].est, 1}};
            sorted.resize(2);
            std::partial_sort_copy(in.begin(), in.end(), sorted.begin(), sorted.end(), TfAscending());
            e = est_mand({sides[sorted[0].idx], sides[sorted[1].idx]}, db);
        } else {
            e = and_of(0, n);
            if (phrase_active) {
                const uint32_t div = d->op == XGM_OP_NEAR ? 2u : (out->window == n ? 4u : 3u);
                e.mn = 0; e.est /= div;
            }
        }
        out->est_min = e.mn; out->est_est = e.est; out->est_max = e.mx;
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

/* ProtoMSet::finalise (protomset.h:484-619) without collapsing / decider / percent cut-off, given the matcher's known_matching_docs
 * (`known`; `known_exact`: it is the exact number of matching documents, i.e. the matcher never skipped a document by weight). */
static void mset_bounds_from(const xgm_query* plan, const xgm_result_hdr* hdr, uint64_t known, bool known_exact, uint32_t* lower, uint32_t* estimated,
                             uint32_t* upper) {
    uint32_t lb = plan->est_min, est = plan->est_est, ub = plan->est_max;
    const uint32_t want = plan->first + plan->maxitems;
    if (hdr->n_hits < want) {
        /* ProtoMSet not full: we got all there are (protomset.h:497-503) */
        lb = est = ub = hdr->n_hits;
    } else if (known_exact && known < plan->check_at_least) {
        /* full, but fewer matching documents than the caller asked to have looked at: the matcher saw every one of them before any
         * weight pruning could start (min_weight stays 0 until checked_enough(), protomset.h:122-126) — the count is exact
         * (protomset.h:515-519) */
        lb = est = ub = (uint32_t)known;
    } else {
        const uint32_t k32 = known > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)known;
        if (k32 > lb) lb = k32;
        if (k32 > est) est = k32;
        if (est < lb) est = lb;
        if (ub < est) ub = est;
    }
    if (lower) *lower = lb;
    if (estimated) *estimated = est;
    if (upper) *upper = ub;
}

extern "C" void xgm_mset_bounds(const xgm_query* plan, const xgm_result_hdr* hdr, uint32_t* lower, uint32_t* estimated, uint32_t* upper) {
    const bool exact = !(hdr->matches_exact & XGM_MATCHES_LOWER_BOUND);
    const uint64_t m = XGM_MATCHES_COUNT(hdr->matches_exact);
    if (exact && m >= hdr->n_hits && m < plan->check_at_least) { mset_bounds_from(plan, hdr, m, true, lower, estimated, upper); return; }
    mset_bounds_from(plan, hdr, hdr->n_hits, false, lower, estimated, upper);         /* stand-in for known_matching_docs (see xgm.h) */
}

extern "C" void xgm_mset_bounds_known(const xgm_query* plan, const xgm_result_hdr* hdr, uint64_t known_matching_docs, uint32_t* lower,
                                      uint32_t* estimated, uint32_t* upper) {
    mset_bounds_from(plan, hdr, known_matching_docs, true, lower, estimated, upper);
}

extern "C" uint64_t xgm_known_matching_docs(const double* w, uint64_t n, uint32_t max_size, uint32_t check_at_least) {
    if (!w || max_size == 0) return n;                                            /* (nothing is kept: min_weight never moves) */
    std::vector<double> heap;                                                     /* min-heap of the max_size best weights so far */
    heap.reserve(max_size);
    uint64_t known = 0;
    double min_weight = 0.0;
    bool heap_built = false;
    for (uint64_t i = 0; i < n; ++i) {
        if (w[i] < min_weight) continue;                    /* the matcher's loop drops it before ProtoMSet sees it (matcher.cc:500-505) */
        ++known;                                            /* ProtoMSet::add */
        if (heap.size() < max_size) { heap.push_back(w[i]); std::push_heap(heap.begin(), heap.end(), std::greater<double>()); continue; }
        if (!heap_built) {                                  /* the (max_size + 1)-th document: min_heap is made, min_weight set if checked_enough() */
            heap_built = true;
            if (known >= check_at_least) min_weight = heap.front();
        }
        if (!(w[i] > heap.front())) continue;               /* not better than the worst kept (equal weight: the larger docid loses): add() returns here */
        std::pop_heap(heap.begin(), heap.end(), std::greater<double>());
        heap.back() = w[i];
        std::push_heap(heap.begin(), heap.end(), std::greater<double>());
        if (known >= check_at_least) min_weight = heap.front();      /* only a replacement moves min_weight (protomset.h:392-398) */
    }
    return known;
}

/* MSet::get_matches_estimated (api/mset.cc:146-153 → api/roundestimate.h:36-69): the estimate rounded to the significant figures the
 * bounds justify.  This is the number Xapiand's HTTP API returns as "total" (src/server/http_client.cc:2554, 2684). */
extern "C" uint32_t xgm_round_estimate(uint32_t m, uint32_t M, uint32_t e) {
    const uint32_t D = M - m;
    if (D == 0 || e == 0) return e;
    uint32_t r = (uint32_t)(pow(10.0, (double)(int)log10((double)D)) + 0.5);       /* exp10(int(log10(D))) */
    while (r > e) r /= 10;
    uint32_t R = e / r * r;
    if (R < m) R += r;
    else if (R > M) R -= r;
    else if (R < e && r % 2 == 0 && e - R == r / 2) { if (e - m < M - e) R += r; }  /* round towards the centre of the range */
    if (R < m || R > M) R = e;
    return R;
}
