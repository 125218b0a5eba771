/* Seam B1: see xgm_matcher_hook.h.  Built only together with the reference's sources (oracle/ref_build/Makefile
 * target xapian_hook_b1); a maintainer adds this file and the patch to src/xapian/matcher/. */
#include "config.h"          /* as every translation unit of the library: the rare() / usual() macros etc. */

#include "xgm_matcher_hook.h"

#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

#include "xapian/error.h"
#include "xapian/api/msetinternal.h"
#include "xapian/api/queryinternal.h"
#include "xapian/api/result.h"
#include "xapian/common/serialise-double.h"
#include "xapian/weight/weightinternal.h"

namespace xgm_hook {

namespace {

struct Shard { xgm_index* idx; Xapian::rev revision; };
std::mutex g_mu;
std::map<std::string, Shard> g_shards;
std::atomic<bool> g_enabled{true}, g_decline_positional{false};
std::atomic<uint64_t> g_answered{0}, g_shape{0}, g_unreg{0}, g_rev{0}, g_dev{0};

struct Lowered {
    xgm_query_desc d;
    std::vector<std::string> terms;     /* owns the bytes d.terms[] points at */
    uint32_t total_subqs = 0;            /* weighted leaves: what QueryOptimiser::inc_total_subqs counts */
};

/* a leaf the device path takes: a term with wqf 1 (MatchAll and scaled leaves are declined) */
bool leaf_term(const Xapian::Query& q, std::string* term) {
    if (q.get_type() != Xapian::Query::LEAF_TERM) return false;
    const auto* t = static_cast<const Xapian::Internal::QueryTerm*>(q.internal.get());
    if (!t || t->get_wqf() != 1 || t->get_term().empty()) return false;
    *term = t->get_term();
    return true;
}

/* a term, or `op` over >= 1 terms → appended to out */
bool terms_of(const Xapian::Query& q, Xapian::Query::op op, std::vector<std::string>* out) {
    std::string t;
    if (leaf_term(q, &t)) { out->push_back(t); return true; }
    if (q.get_type() != op || q.get_num_subqueries() == 0) return false;
    for (size_t i = 0; i < q.get_num_subqueries(); ++i) {
        if (!leaf_term(q.get_subquery(i), &t)) return false;
        out->push_back(t);
    }
    return true;
}

/* A nested query → the post-order program of xgm_query_desc (XGM_OP_TREE): AND / OR / AND_NOT / AND_MAYBE / FILTER over
 * sub-trees, OP_SYNONYM of terms, OP_SCALE_WEIGHT, leaves with their wqf.  false = a shape the device path declines. */
bool lower_tree_node(const Xapian::Query& q, Lowered* L) {
    xgm_query_desc& d = L->d;
    const Xapian::Query::op op = q.get_type();
    if (op == Xapian::Query::LEAF_TERM) {
        const auto* t = static_cast<const Xapian::Internal::QueryTerm*>(q.internal.get());
        if (!t || t->get_term().empty() || L->terms.size() >= XGM_MAX_TERMS || d.n_tree >= XGM_MAX_TREE) return false;
        d.wqf[L->terms.size()] = t->get_wqf();
        d.tree[d.n_tree].kind = XGM_T_TERM; d.tree[d.n_tree].arity = 0; d.tree[d.n_tree].term = (uint16_t)L->terms.size();
        ++d.n_tree;
        L->terms.push_back(t->get_term());
        return true;
    }
    uint8_t kind;
    switch (op) {
    case Xapian::Query::OP_AND: kind = XGM_T_AND; break;
    case Xapian::Query::OP_OR: kind = XGM_T_OR; break;
    case Xapian::Query::OP_AND_NOT: kind = XGM_T_AND_NOT; break;
    case Xapian::Query::OP_AND_MAYBE: kind = XGM_T_AND_MAYBE; break;
    case Xapian::Query::OP_FILTER: kind = XGM_T_FILTER; break;
    case Xapian::Query::OP_SYNONYM: kind = XGM_T_SYNONYM; break;
    case Xapian::Query::OP_SCALE_WEIGHT: kind = XGM_T_SCALE; break;
    default: return false;
    }
    const size_t n = q.get_num_subqueries();
    if (n == 0 || n > 255) return false;
    for (size_t i = 0; i < n; ++i) {
        const Xapian::Query sub = q.get_subquery(i);
        if (kind == XGM_T_SYNONYM && sub.get_type() != Xapian::Query::LEAF_TERM) return false;
        if (!lower_tree_node(sub, L)) return false;
    }
    if (d.n_tree >= XGM_MAX_TREE) return false;
    d.tree[d.n_tree].kind = kind; d.tree[d.n_tree].arity = (uint8_t)n; d.tree[d.n_tree].term = 0;
    if (kind == XGM_T_SCALE) {
        /* QueryScaleWeight keeps its factor private; its serialisation is one tag byte + serialise_double(factor)
         * (api/queryinternal.cc: QueryScaleWeight::serialise) */
        std::string ser;
        q.internal->serialise(ser);
        if (ser.size() < 2) return false;
        const char* p = ser.data() + 1;
        d.tree_scale[d.n_tree] = unserialise_double(&p, ser.data() + ser.size());
    }
    ++d.n_tree;
    return true;
}

/* Xapian::Query → xgm_query_desc for the FLAT shapes of SURVEY §8 (a3) and (f).2 (each has its own fast kernel) */
bool lower_flat(const Xapian::Query& q, Lowered* L) {
    memset(&L->d, 0, sizeof L->d);
    L->terms.clear();
    const Xapian::Query::op op = q.get_type();
    std::string t;
    if (leaf_term(q, &t)) {
        L->d.op = XGM_OP_AND;
        L->terms.push_back(t);
        L->total_subqs = 1;
    } else if (op == Xapian::Query::OP_AND || op == Xapian::Query::OP_OR) {
        if (!terms_of(q, op, &L->terms)) return false;
        L->d.op = op == Xapian::Query::OP_AND ? XGM_OP_AND : XGM_OP_OR;
        L->total_subqs = (uint32_t)L->terms.size();
    } else if (op == Xapian::Query::OP_PHRASE || op == Xapian::Query::OP_NEAR) {
        if (g_decline_positional.load(std::memory_order_relaxed)) return false;
        const size_t n = q.get_num_subqueries();
        for (size_t i = 0; i < n; ++i) {
            const Xapian::Query s = q.get_subquery(i);
            if (!leaf_term(s, &t)) return false;         /* phrase offsets are the subquery ORDER (exactphrasepostlist.cc:75-133) */
            L->terms.push_back(t);
        }
        L->d.op = op == Xapian::Query::OP_PHRASE ? XGM_OP_PHRASE : XGM_OP_NEAR;
        const auto* w = static_cast<const Xapian::Internal::QueryWindowed*>(q.internal.get());
        L->d.window = (uint32_t)w->get_window();
        L->total_subqs = (uint32_t)n;
    } else if (op == Xapian::Query::OP_AND_NOT || op == Xapian::Query::OP_AND_MAYBE || op == Xapian::Query::OP_FILTER) {
        if (q.get_num_subqueries() != 2) return false;
        if (!terms_of(q.get_subquery(0), Xapian::Query::OP_AND, &L->terms)) return false;
        L->d.n_required = (uint32_t)L->terms.size();
        if (!terms_of(q.get_subquery(1), op == Xapian::Query::OP_FILTER ? Xapian::Query::OP_AND : Xapian::Query::OP_OR, &L->terms)) return false;
        L->d.op = op == Xapian::Query::OP_AND_NOT ? XGM_OP_AND_NOT : op == Xapian::Query::OP_AND_MAYBE ? XGM_OP_AND_MAYBE : XGM_OP_FILTER;
        L->total_subqs = op == Xapian::Query::OP_AND_MAYBE ? (uint32_t)L->terms.size() : L->d.n_required;
    } else {
        return false;
    }
    if (L->terms.empty() || L->terms.size() > XGM_MAX_TERMS) return false;
    L->d.n_terms = (uint32_t)L->terms.size();
    for (size_t i = 0; i < L->terms.size(); ++i) { L->d.terms[i] = L->terms[i].data(); L->d.term_len[i] = (uint32_t)L->terms[i].size(); }
    return true;
}

/* flat shape, else the general tree (nested operators, OP_SYNONYM, OP_SCALE_WEIGHT, wqf != 1); false = decline */
bool lower(const Xapian::Query& q, Lowered* L) {
    if (lower_flat(q, L)) return true;
    memset(&L->d, 0, sizeof L->d);
    L->terms.clear();
    if (!lower_tree_node(q, L)) return false;
    L->d.op = XGM_OP_TREE;
    L->total_subqs = 0;                     /* comes back from the planner (xgm_query.total_subqs) */
    L->d.n_terms = (uint32_t)L->terms.size();
    for (size_t i = 0; i < L->terms.size(); ++i) { L->d.terms[i] = L->terms[i].data(); L->d.term_len[i] = (uint32_t)L->terms[i].size(); }
    return true;
}

}  // namespace

void register_shard(const Xapian::Database& db, xgm_index* idx) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_shards[db.get_uuid()] = Shard{idx, db.get_revision()};
}

void unregister_shard(const Xapian::Database& db) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_shards.erase(db.get_uuid());
}

void set_enabled(bool on) { g_enabled.store(on); }
void set_decline_positional(bool on) { g_decline_positional.store(on); }
Counters counters() { return Counters{g_answered.load(), g_shape.load(), g_unreg.load(), g_rev.load(), g_dev.load()}; }

bool try_get_mset(const Xapian::Database& db, const Xapian::Query& query, const Xapian::Weight::Internal& stats,
                  const Xapian::Weight& wtscheme, bool full_db_has_positions, Xapian::doccount first,
                  Xapian::doccount maxitems, Xapian::doccount check_at_least, const Xapian::MatchDecider* mdecider,
                  const Xapian::KeyMaker* sorter, Xapian::doccount collapse_max, int percent_threshold,
                  double weight_threshold, Xapian::Enquire::docid_order order, bool sort_by_rel, double time_limit,
                  size_t n_matchspies, Xapian::MSet& out) {
    if (!g_enabled.load(std::memory_order_relaxed)) return false;
    /* eligibility */
    if (db.size() != 1 || !sort_by_rel || order == Xapian::Enquire::DESCENDING || collapse_max != 0 || percent_threshold != 0 ||
        weight_threshold != 0.0 || mdecider || sorter || n_matchspies != 0 || stats.rset_size != 0 || time_limit != 0.0 ||
        wtscheme.name() != "Xapian::BM25Weight") {
        ++g_shape;
        return false;
    }
    Lowered L;
    if (!lower(query, &L)) { ++g_shape; return false; }
    Shard sh;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_shards.find(db.get_uuid());
        if (it == g_shards.end()) { ++g_unreg; return false; }
        sh = it->second;
    }
    if (sh.revision != db.get_revision()) { ++g_rev; return false; }      /* the segment is of another revision: CPU until refreshed */

    /* BM25 parameters: the scheme's own serialisation (bm25weight.cc:145-153) */
    {
        const std::string ser = wtscheme.serialise();
        const char* p = ser.data();
        const char* end = p + ser.size();
        L.d.k1 = unserialise_double(&p, end); L.d.k2 = unserialise_double(&p, end); L.d.k3 = unserialise_double(&p, end);
        L.d.b = unserialise_double(&p, end); L.d.min_normlen = unserialise_double(&p, end);
    }
    L.d.first = first; L.d.maxitems = maxitems; L.d.check_at_least = check_at_least;

    /* merged statistics, exactly what the CPU matcher would weigh with (weightinternal.h:72-111) */
    xgm_global_stats gs;
    memset(&gs, 0, sizeof gs);
    gs.total_length = stats.total_length;
    gs.collection_size = stats.collection_size;
    gs.full_db_has_positions = full_db_has_positions ? 1u : 0u;
    for (size_t i = 0; i < L.terms.size(); ++i) {
        auto it = stats.termfreqs.find(L.terms[i]);
        if (it == stats.termfreqs.end()) { ++g_shape; return false; }
        gs.termfreq[i] = it->second.termfreq;
    }

    const uint32_t k = first + maxitems;
    std::vector<xgm_hit> hits(k ? k : 1);
    xgm_result_hdr hdr;
    memset(&hdr, 0, sizeof hdr);
    xgm_query plan;
    int rc = xgm_plan_query(sh.idx, &L.d, &gs, &plan);
    if (rc == XGM_OK) rc = xgm_search_batch(sh.idx, &plan, 1, k ? k : 1, hits.data(), &hdr);
    if (rc > 0) { ++g_dev; return false; }                                   /* declined by the planner: CPU matcher */
    if (rc < 0) throw Xapian::DatabaseError(std::string("xgm: ") + xgm_last_error());
    if (L.d.op == XGM_OP_TREE) L.total_subqs = plan.total_subqs;

    /* the MSet, as ProtoMSet::finalise builds it (protomset.h:466-471, 484-682).  matches_*: exact counts
     * (the reference's are estimates; documented exception). */
    std::vector<Result> items;
    const uint32_t skip = std::min<uint32_t>(first, hdr.n_hits);
    items.reserve(hdr.n_hits - skip);
    for (uint32_t i = skip; i < hdr.n_hits; ++i) items.emplace_back(hits[i].weight, hits[i].docid);
    double percent_scale = 0.0;
    if (hdr.n_hits && hdr.max_attained > 0.0 && L.total_subqs) {
        percent_scale = hdr.max_weight_subqs_matched / double(L.total_subqs);
        percent_scale /= hdr.max_attained;
    }
    /* bounds and estimate as ProtoMSet::finalise derives them from the tree's static termfreq bounds (xgm_mset_bounds: the
     * upper bound is the reference's; the lower bound and the estimate use the number of documents returned where the
     * reference uses how many its matcher happened to weigh — DESIGN.md) */
    uint32_t lb = 0, est = 0, ub = 0;
    xgm_mset_bounds(&plan, &hdr, &lb, &est, &ub);
    out = Xapian::MSet(new Xapian::MSet::Internal(first, ub, lb, est, ub, lb, est, hdr.max_possible, hdr.max_attained, std::move(items),
                                                   percent_scale * 100.0));
    ++g_answered;
    return true;
}

}  // namespace xgm_hook
