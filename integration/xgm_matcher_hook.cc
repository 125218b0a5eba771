/* Seam B1: see xgm_matcher_hook.h.  Built only together with the reference's sources (oracle/ref_build/Makefile
 * target xapian_hook_b1); a maintainer adds this file and the patch to src/xapian/matcher/. */
#include "config.h"          /* as every translation unit of the library: the rare() / usual() macros etc. */

#include "xgm_matcher_hook.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>

#include "xapian/error.h"
#include "xapian/api/enquireinternal.h"
#include "xapian/api/msetinternal.h"
#include "xapian/api/postlist.h"
#include "xapian/api/queryinternal.h"
#include "xapian/api/result.h"
#include "xapian/common/pack.h"
#include "xapian/common/serialise-double.h"
#include "xapian/weight/weightinternal.h"

namespace xgm_hook {

namespace {

/* A device column (xgm_index_attach_column_ordinals) of one shard revision and the strings its ordinals stand for */
struct Column { uint32_t slot_id = 0; std::vector<std::string> values; };
struct ShardColumns {
    std::mutex mu;                                               /* one builder at a time per shard */
    std::map<std::string, std::shared_ptr<Column>> by_key;       /* "v<slot>" | "k<KeyMaker name>\0<serialised>" */
    uint32_t next_slot = 0x40000000u;                            /* synthetic slot numbers of the index's column map */
};
/* own: shared ownership of the device index — a search copies the Shard under g_mu and so keeps the index alive for its whole call, whatever
 * the registry does meanwhile (a commit replacing the revision, the shard closing: ADVICE r5); idx = own.get() */
struct Shard { std::shared_ptr<xgm_index> own; xgm_index* idx; Xapian::rev revision; std::shared_ptr<ShardColumns> cols; uint32_t batch = 0; };
std::mutex g_mu;
std::map<std::string, Shard> g_shards;
std::map<std::string, std::shared_ptr<const SpyAdapter>> g_spy_adapters;      /* (shared: a search keeps its adapter although the name is registered again meanwhile) */
std::atomic<bool> g_enabled{true}, g_exact_bounds{false}, g_near_colocated{false}, g_replay{false};
std::atomic<int> g_positional{POSITIONAL_DECLINE}, g_collapse{COLLAPSE_DECLINE};
std::atomic<uint64_t> g_replay_limit{4u * 1000u * 1000u};       /* matches up to which a replay downloads the match (set_replay_limit) */
std::atomic<uint32_t> g_column_limit{50u * 1000u * 1000u};     /* documents per shard up to which a column is built on the search thread */
std::atomic<uint64_t> g_answered{0}, g_shape{0}, g_unreg{0}, g_rev{0}, g_dev{0}, g_sorted{0}, g_spied{0}, g_collapsed{0}, g_columns{0}, g_replayed{0}, g_combined{0}, g_combined_launches{0};

struct Lowered {
    xgm_query_desc d;
    std::vector<std::string> terms;     /* owns the bytes d.terms[] points at */
    std::vector<bool> lazy;              /* terms[i] came out of a wildcard expansion (LocalSubMatch::open_post_list, lazy_weight) */
    uint32_t total_subqs = 0;            /* weighted leaves: what QueryOptimiser::inc_total_subqs counts */
    const xgm_index* idx = nullptr;      /* the shard: wildcards expand over ITS dictionary, as qopt->db.open_allterms does */
};

/* OP_WILDCARD / OP_EDIT_DISTANCE → an OP_SYNONYM (or OP_OR) group over the shard's terms the expansion selects, as
 * Context<T>::expand_wildcard / expand_edit_distance do (api/queryinternal.cc:246-315, 319-383): the terms under the pattern's
 * fixed prefix in term order (with an empty prefix, none that starts with A-Z), each put to the reference's OWN test —
 * QueryWildcard::test_prefix_known for '?' / '*' patterns, QueryEditDistance::test — then the limit.  `test` null: every term
 * under the prefix is taken (the 1.4-style "prefix*").  false = leave the query to the CPU matcher: OP_MAX, an expansion the
 * reference would refuse (WILDCARD_LIMIT_ERROR), none or more than the device's leaves, or a most-frequent cut through a tie of
 * term frequencies (std::nth_element leaves the choice unspecified). */
template <class Test>
bool lower_expansion(Lowered* L, const std::string& pfx, Xapian::termcount max_expansion, int max_type, Xapian::Query::op combiner, const Test* test) {
    if (!L->idx || (combiner != Xapian::Query::OP_SYNONYM && combiner != Xapian::Query::OP_OR)) return false;
    uint32_t n_total = 0;
    std::vector<uint32_t> ids(XGM_MAX_TERMS + 1);
    if (xgm_expand_prefix(L->idx, pfx.data(), pfx.size(), (uint32_t)ids.size(), ids.data(), &n_total) != XGM_OK || n_total == 0) return false;
    const bool filtered = test != nullptr || pfx.empty();
    if (!filtered && (max_expansion == 0 || n_total <= max_expansion || max_type == Xapian::Query::WILDCARD_LIMIT_ERROR) && n_total > XGM_MAX_TERMS) return false;
    if (filtered || (max_expansion != 0 && n_total > max_expansion && max_type == Xapian::Query::WILDCARD_LIMIT_MOST_FREQUENT)) {
        if (n_total > ids.size()) {
            ids.resize(n_total);
            if (xgm_expand_prefix(L->idx, pfx.data(), pfx.size(), n_total, ids.data(), &n_total) != XGM_OK) return false;
        }
    }
    ids.resize(std::min<size_t>(ids.size(), n_total));
    if (filtered) {
        size_t kept = 0;
        std::string cand;
        for (size_t i = 0; i < ids.size(); ++i) {
            const char* b = nullptr; size_t bl = 0;
            if (xgm_term_info(L->idx, ids[i], &b, &bl, nullptr, nullptr) != XGM_OK || bl == 0) return false;
            if (pfx.empty() && b[0] >= 'A' && b[0] <= 'Z') continue;            /* skip_ucase: prefixed terms */
            if (test) { cand.assign(b, bl); if (!(*test)(cand)) continue; }
            ids[kept++] = ids[i];
            /* no limit will cut it down and it no longer fits: stop walking the dictionary */
            if (kept > XGM_MAX_TERMS && (max_expansion == 0 || max_expansion > XGM_MAX_TERMS)) return false;
            if (max_type == Xapian::Query::WILDCARD_LIMIT_FIRST && max_expansion != 0 && kept > max_expansion) break;   /* the rest is cut anyway */
        }
        ids.resize(kept);
        n_total = (uint32_t)kept;
        if (n_total == 0) return false;
    }
    uint32_t n = n_total;
    if (max_expansion != 0 && n_total > max_expansion) {
        if (max_type == Xapian::Query::WILDCARD_LIMIT_FIRST) {
            n = max_expansion;
        } else if (max_type == Xapian::Query::WILDCARD_LIMIT_MOST_FREQUENT) {
            if (combiner != Xapian::Query::OP_SYNONYM) return false;       /* the OR tree's tie order would follow nth_element's permutation */
            std::vector<std::pair<uint32_t, uint32_t>> by_tf;              /* (termfreq, id) */
            for (uint32_t i = 0; i < n_total; ++i) { uint32_t tf = 0; xgm_term_info(L->idx, ids[i], nullptr, nullptr, &tf, nullptr); by_tf.emplace_back(tf, ids[i]); }
            std::sort(by_tf.begin(), by_tf.end(), [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) { return a.first > b.first; });
            if (by_tf[max_expansion - 1].first == by_tf[max_expansion].first) return false;
            n = max_expansion;
            for (uint32_t i = 0; i < n; ++i) ids[i] = by_tf[i].second;
        } else {
            return false;                                                  /* WILDCARD_LIMIT_ERROR: the CPU matcher throws WildcardError */
        }
    }
    xgm_query_desc& d = L->d;
    if (n > XGM_MAX_TERMS || L->terms.size() + n > XGM_MAX_TERMS || d.n_tree + n + 1 > XGM_MAX_TREE || n > 255) return false;
    for (uint32_t i = 0; i < n; ++i) {
        const char* b = nullptr; size_t bl = 0;
        if (xgm_term_info(L->idx, ids[i], &b, &bl, nullptr, nullptr) != XGM_OK) return false;
        d.wqf[L->terms.size()] = 1;
        d.tree[d.n_tree].kind = XGM_T_TERM; d.tree[d.n_tree].arity = 0; d.tree[d.n_tree].term = (uint16_t)L->terms.size();
        ++d.n_tree;
        L->terms.emplace_back(b, bl);
        L->lazy.resize(L->terms.size(), false);
        L->lazy.back() = true;
    }
    d.tree[d.n_tree].kind = combiner == Xapian::Query::OP_SYNONYM ? XGM_T_WILDCARD : XGM_T_WILDCARD_OR;
    d.tree[d.n_tree].arity = (uint8_t)n; d.tree[d.n_tree].term = 0;
    ++d.n_tree;
    return true;
}

/* (max_expansion, combiner) of a QueryWildcard / QueryEditDistance: both keep them private and serialise them alike —
 * tag, pack_uint(max_expansion), flags byte, combiner byte, ... (api/queryinternal.cc:1507-1514, 1617-1626) */
bool expansion_header(const Xapian::Query::Internal* qi, Xapian::termcount* max_expansion, Xapian::Query::op* combiner) {
    std::string ser;
    qi->serialise(ser);
    const char* p = ser.data() + 1;
    const char* end = ser.data() + ser.size();
    if (ser.size() < 4 || !unpack_uint(&p, end, max_expansion) || end - p < 2) return false;
    *combiner = Xapian::Query::op((unsigned char)p[1]);
    return true;
}

bool lower_wildcard(const Xapian::Query& q, Lowered* L) {
    const auto* w = static_cast<const Xapian::Internal::QueryWildcard*>(q.internal.get());
    Xapian::termcount max_expansion;
    Xapian::Query::op combiner;
    if (!w || !expansion_header(w, &max_expansion, &combiner)) return false;
    struct Test { const Xapian::Internal::QueryWildcard* w; bool operator()(const std::string& c) const { return w->test_prefix_known(c); } } test{w};
    return lower_expansion(L, w->get_fixed_prefix(), max_expansion, w->get_max_type(), combiner, w->get_just_flags() != 0 ? &test : (const Test*)nullptr);
}

bool lower_edit_distance(const Xapian::Query& q, Lowered* L) {
    const auto* e = static_cast<const Xapian::Internal::QueryEditDistance*>(q.internal.get());
    Xapian::termcount max_expansion;
    Xapian::Query::op combiner;
    if (!e || !expansion_header(e, &max_expansion, &combiner)) return false;
    struct Test { const Xapian::Internal::QueryEditDistance* e; bool operator()(const std::string& c) const { return e->test(c) != 0; } } test{e};
    return lower_expansion(L, std::string(e->get_pattern(), 0, e->get_fixed_prefix_len()), max_expansion, e->get_max_type(), combiner, &test);
}

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
    if (op == Xapian::Query::OP_WILDCARD) return lower_wildcard(q, L);
    if (op == Xapian::Query::OP_EDIT_DISTANCE) return lower_edit_distance(q, L);
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
        if (g_positional.load(std::memory_order_relaxed) == POSITIONAL_DECLINE) return false;
        /* NEAR where terms may share a position: the device runs NearPostList's procedure per document from QUERY order; the reference's
         * `terms` vector keeps the order its previous test_doc left (nearpostlist.cc:80 sorts the member in place), which decides
         * which of two coinciding heads moves — its answer depends on the documents tested before.  Answered on the device
         * as INTENDED; the byte-compatible mode keeps such queries on the CPU matcher. */
        if (op == Xapian::Query::OP_NEAR && g_near_colocated.load(std::memory_order_relaxed) &&
            g_positional.load(std::memory_order_relaxed) != POSITIONAL_INTENDED) return false;
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
    L->lazy.clear();
    if (!lower_tree_node(q, L)) return false;
    L->d.op = XGM_OP_TREE;
    L->total_subqs = 0;                     /* comes back from the planner (xgm_query.total_subqs) */
    L->d.n_terms = (uint32_t)L->terms.size();
    for (size_t i = 0; i < L->terms.size(); ++i) { L->d.terms[i] = L->terms[i].data(); L->d.term_len[i] = (uint32_t)L->terms[i].size(); }
    return true;
}

}  // namespace


namespace {

/* The column of `key` for this shard revision: built on first use, attached to the index under a synthetic slot number, kept with its
 * distinct strings (ordinal o > 0 stands for values[o - 1]; 0 = no value / empty key).  `walk(emit)` calls emit(docid, string) for every
 * document that has one, and is run TWICE — first to collect the distinct strings, then to assign ordinals — so that nothing is held
 * per document but its 4-byte ordinal (one std::string per document of a large shard was gigabytes of host memory).  The
 * build holds the shard's column mutex: other sorted / spied / collapsed searches of that shard wait for it once per (revision, key);
 * shards beyond set_column_build_limit() documents are declined instead, and a build that failed is remembered (a negative entry) so
 * that later searches decline at once instead of rebuilding. */
template <class Walk>
std::shared_ptr<Column> ensure_column(const Shard& sh, const Xapian::Database& db, const std::string& key, Walk walk) {
    ShardColumns& sc = *sh.cols;
    std::lock_guard<std::mutex> lk(sc.mu);
    auto it = sc.by_key.find(key);
    if (it != sc.by_key.end()) return it->second;                /* (nullptr: a remembered failure) */
    const Xapian::docid last = db.get_lastdocid();
    if (last > g_column_limit.load(std::memory_order_relaxed)) return nullptr;      /* not cached: the limit may be raised */
    auto col = std::make_shared<Column>();
    try {
        walk([&](Xapian::docid, const std::string& v) { if (!v.empty()) col->values.push_back(v); });
        std::sort(col->values.begin(), col->values.end());
        col->values.erase(std::unique(col->values.begin(), col->values.end()), col->values.end());
        col->values.shrink_to_fit();
        std::vector<uint32_t> ord(size_t(last) + 1, 0u);
        walk([&](Xapian::docid did, const std::string& v) {
            if (!v.empty() && did <= last) ord[did] = uint32_t(std::lower_bound(col->values.begin(), col->values.end(), v) - col->values.begin()) + 1u;
        });
        col->slot_id = sc.next_slot++;
        if (xgm_index_attach_column_ordinals(sh.idx, col->slot_id, ord.data(), (uint32_t)ord.size(), (uint32_t)col->values.size()) != XGM_OK) col.reset();
    } catch (const Xapian::Error&) {
        col.reset();
    } catch (const std::bad_alloc&) {
        col.reset();
    }
    sc.by_key.emplace(key, col);
    if (col) ++g_columns;
    return col;
}

/* value slot → column, through the shard's value stream (ValueIterator walks the slot's chunks in docid order) */
std::shared_ptr<Column> value_column(const Shard& sh, const Xapian::Database& db, Xapian::valueno slot) {
    return ensure_column(sh, db, "v" + std::to_string(slot), [&](const std::function<void(Xapian::docid, const std::string&)>& emit) {
        for (Xapian::ValueIterator it = db.valuestream_begin(slot); it != db.valuestream_end(slot); ++it) emit(it.get_docid(), *it);
    });
}

/* KeyMaker → column of its keys.  The identity of a key maker is its class name + serialisation (Xapiand's
 * Multi_MultiValueKeyMaker implements both, src/multivalue/keymaker.h:366-372); one that does not serialise is declined. */
std::shared_ptr<Column> key_column(const Shard& sh, const Xapian::Database& db, const Xapian::KeyMaker& sorter) {
    std::string key = "k";
    try {
        key += sorter.name();
        key += '\0';
        key += sorter.serialise();
    } catch (const Xapian::Error&) {
        return nullptr;
    }
    if (key.size() == 2) return nullptr;                       /* no name, no serialisation: nothing to recognise it by */
    return ensure_column(sh, db, key, [&](const std::function<void(Xapian::docid, const std::string&)>& emit) {
        for (Xapian::PostingIterator p = db.postlist_begin(std::string()); p != db.postlist_end(std::string()); ++p) emit(*p, sorter(db.get_document(*p)));
    });
}

/* which value slot a spy counts: Xapian::ValueCountMatchSpy serialises exactly its slot (api/matchspy.cc: pack_uint_last) */
bool spy_slot_of(const Xapian::MatchSpy& spy, Xapian::valueno* slot, std::shared_ptr<const SpyAdapter>* adapter) {
    adapter->reset();
    std::string name;
    try { name = spy.name(); } catch (const Xapian::Error&) { return false; }
    if (name == "Xapian::ValueCountMatchSpy") {
        const std::string ser = spy.serialise();
        const char* p = ser.data();
        Xapian::valueno v;
        if (!unpack_uint_last(&p, p + ser.size(), &v)) return false;
        *slot = v;
        return true;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_spy_adapters.find(name);
    if (it == g_spy_adapters.end() || !it->second || !it->second->slot_of || !it->second->feed) return false;
    *adapter = it->second;
    return (*adapter)->slot_of(spy, slot);
}


/* What maybe_replay needs from the Matcher::get_mset call that is about to run get_local_mset on this thread */
struct ReplayCtx { bool armed = false; const Xapian::Weight::Internal* stats = nullptr; const Xapian::Weight* wt = nullptr; bool full_db_has_positions = false; };
thread_local ReplayCtx tl_replay;

/* ALL matching documents of a search, in docid order, as a PostList for Matcher::get_local_mset's loop (see set_replay) */
class ReplayPostList : public Xapian::Internal::PostList {
    std::unique_ptr<xgm_hit[]> hits;           /* ascending docid (xgm_search_all) */
    size_t n_hits;
    size_t pos = size_t(-1);                   /* before the first */
    Xapian::doccount tf_min, tf_est, tf_max;
    double max_weight;
  public:
    ReplayPostList(std::unique_ptr<xgm_hit[]>&& h, size_t n, Xapian::doccount mn, Xapian::doccount est, Xapian::doccount mx, double mw)
        : hits(std::move(h)), n_hits(n), tf_min(mn), tf_est(est), tf_max(mx), max_weight(mw) {}
    Xapian::doccount get_termfreq_min() const override { return tf_min; }
    Xapian::doccount get_termfreq_max() const override { return tf_max; }
    Xapian::doccount get_termfreq_est() const override { return tf_est; }
    Xapian::docid get_docid() const override { return hits[pos].docid; }
    double get_weight(Xapian::termcount, Xapian::termcount) const override { return hits[pos].weight; }
    bool at_end() const override { return pos != size_t(-1) && pos >= n_hits; }
    double recalc_maxweight() override { return max_weight; }
    PostList* next(double) override { ++pos; return nullptr; }
    PostList* skip_to(Xapian::docid did, double) override {
        if (pos == size_t(-1)) pos = 0;
        if (pos < n_hits && hits[pos].docid < did)
            pos = size_t(std::lower_bound(hits.get() + pos, hits.get() + n_hits, did, [](const xgm_hit& h, Xapian::docid d) { return h.docid < d; }) - hits.get());
        return nullptr;
    }
    Xapian::termcount count_matching_subqs() const override { return hits[pos].subqs_matched; }
    std::string get_description() const override { return "XgmReplay(" + std::to_string(n_hits) + ")"; }
};

/* ---- searches under a sort, with a spy or collapsed, from many threads: ONE launch for those that wait together (round 6) -----------------------
 * The plain searches of concurrent threads meet in the library's dispatcher (xgm_index_set_batching).  These have their own entry points —
 * xgm_search_sorted_batch / _sorted_spy_batch / _collapsed_batch: nq searches under ONE sort spec, spy slot or collapse key — so the hook
 * combines them itself: searches of the same index and the same spec form a lane; the first to arrive launches at once, those that arrive
 * while a launch of the lane is in flight wait together and go out as ONE launch when it returns (no linger, no timer: the batch is what
 * accumulated behind the launch before it).  Every search gets exactly what its single call would have given it
 * (tests/test_gpu_sorted.py::test_sorted_batch_equals_single_searches_and_the_oracle); a batch the library declines as a whole is run one by one. */
struct ComboKey {
    xgm_index* idx; uint32_t kind;                     /* 0 sorted, 1 sorted + spy, 2 collapsed */
    uint32_t has_sort, sort_by, sort_slot, reverse, aux_slot, aux_max;
    bool operator<(const ComboKey& o) const {
        return std::tie(idx, kind, has_sort, sort_by, sort_slot, reverse, aux_slot, aux_max) <
               std::tie(o.idx, o.kind, o.has_sort, o.sort_by, o.sort_slot, o.reverse, o.aux_slot, o.aux_max);
    }
};
struct ComboReq {
    const xgm_query* plan; uint32_t k;
    xgm_hit* hits; uint32_t* ord; uint32_t* cord; uint32_t* ccount; xgm_result_hdr* hdr; uint64_t* clb; uint32_t* counts; uint32_t n_counts;
    int rc; std::string error;
};
struct ComboGroup { std::vector<ComboReq*> reqs; std::condition_variable cv; bool done = false; };
struct ComboLane { bool running = false; std::deque<std::shared_ptr<ComboGroup>> waiting; };
constexpr size_t kComboMax = 256;
/* (never destroyed: threads of the host may still be searching when the process exits) */
std::mutex& g_combo_mu = *new std::mutex;
std::map<ComboKey, ComboLane>& g_combo = *new std::map<ComboKey, ComboLane>;

int combo_single(const ComboKey& K, const xgm_sort_spec* sort, ComboReq& r) {
    int rc;
    if (K.kind == 2u) rc = xgm_search_collapsed(K.idx, r.plan, sort, K.aux_slot, K.aux_max, r.hits, r.ord, r.cord, r.ccount, r.hdr, r.clb);
    else if (K.kind == 1u) rc = xgm_search_sorted_spy(K.idx, r.plan, sort, r.hits, r.ord, r.hdr, K.aux_slot, r.counts, r.n_counts);
    else rc = xgm_search_sorted(K.idx, r.plan, sort, r.hits, r.ord, r.hdr);
    if (rc < 0) r.error = xgm_last_error();            /* (the library's message is per thread: the leader's, handed to the request's own thread) */
    return rc;
}

void combo_run(const ComboKey& K, ComboGroup& g) {
    xgm_sort_spec spec;
    memset(&spec, 0, sizeof spec);
    spec.sort_by = K.sort_by; spec.slot = K.sort_slot; spec.reverse = K.reverse;
    const xgm_sort_spec* sort = K.has_sort ? &spec : nullptr;
    const size_t n = g.reqs.size();
    if (n > 1) {
        uint32_t ks = 1;
        for (const ComboReq* r : g.reqs) ks = std::max(ks, r->k);
        const uint32_t nc = g.reqs[0]->n_counts;
        std::vector<xgm_query> plans(n);
        for (size_t i = 0; i < n; ++i) plans[i] = *g.reqs[i]->plan;
        std::vector<xgm_hit> hits(n * ks);
        std::vector<uint32_t> ord(sort ? n * ks : 0), cord(K.kind == 2u ? n * ks : 0), ccount(K.kind == 2u ? n * ks : 0), counts(K.kind == 1u ? n * (size_t)nc : 0);
        std::vector<xgm_result_hdr> hdrs(n);
        std::vector<uint64_t> clb(n, 0);
        int rc;
        if (K.kind == 2u) rc = xgm_search_collapsed_batch(K.idx, plans.data(), (uint32_t)n, sort, K.aux_slot, K.aux_max, ks, hits.data(), sort ? ord.data() : nullptr,
                                                          cord.data(), ccount.data(), hdrs.data(), clb.data());
        else if (K.kind == 1u) rc = xgm_search_sorted_spy_batch(K.idx, plans.data(), (uint32_t)n, sort, ks, hits.data(), sort ? ord.data() : nullptr, hdrs.data(),
                                                                K.aux_slot, counts.data(), nc);
        else rc = xgm_search_sorted_batch(K.idx, plans.data(), (uint32_t)n, sort, ks, hits.data(), ord.data(), hdrs.data());
        if (rc == XGM_OK) {
            for (size_t i = 0; i < n; ++i) {
                ComboReq& r = *g.reqs[i];
                const uint32_t m = std::min(hdrs[i].n_hits, r.k);
                std::copy(hits.begin() + i * ks, hits.begin() + i * ks + m, r.hits);
                if (r.ord && sort) std::copy(ord.begin() + i * ks, ord.begin() + i * ks + m, r.ord);
                if (r.cord) std::copy(cord.begin() + i * ks, cord.begin() + i * ks + m, r.cord);
                if (r.ccount) std::copy(ccount.begin() + i * ks, ccount.begin() + i * ks + m, r.ccount);
                if (r.counts) std::copy(counts.begin() + i * (size_t)nc, counts.begin() + (i + 1) * (size_t)nc, r.counts);
                if (r.clb) *r.clb = clb[i];
                *r.hdr = hdrs[i];
                r.rc = XGM_OK;
            }
            g_combined += n;
            ++g_combined_launches;
            return;
        }
        /* declined (one query of the batch is of a shape the batch entry does not take) or failed as a whole: each on its own */
    }
    for (ComboReq* r : g.reqs) r->rc = combo_single(K, sort, *r);
}

/* the calling thread's search `r`, alone or in a launch with those of other threads */
int combo_search(const ComboKey& K, ComboReq& r) {
    std::unique_lock<std::mutex> lk(g_combo_mu);
    ComboLane& lane = g_combo[K];
    if (lane.waiting.empty() || lane.waiting.back()->reqs.size() >= kComboMax) lane.waiting.push_back(std::make_shared<ComboGroup>());
    std::shared_ptr<ComboGroup> grp = lane.waiting.back();
    grp->reqs.push_back(&r);
    const bool first = grp->reqs.size() == 1;
    while (!grp->done) {
        if (first && !lane.running && lane.waiting.front() == grp) {
            lane.running = true;
            lane.waiting.pop_front();                   /* closed: later arrivals start the next group */
            lk.unlock();
            try {
                combo_run(K, *grp);
            } catch (const std::exception& e) {         /* (out of memory for the group's buffers: every member is told, the lane goes on) */
                for (ComboReq* q : grp->reqs) { q->rc = XGM_E_INVALID; q->error = e.what(); }
            }
            lk.lock();
            lane.running = false;
            grp->done = true;
            grp->cv.notify_all();
            if (!lane.waiting.empty()) lane.waiting.front()->cv.notify_all();      /* its first member launches it */
            else g_combo.erase(K);
            break;
        }
        grp->cv.wait(lk);
    }
    return r.rc;
}

/* EVERY matching document of a planned query in ascending docid order with its weight (xgm_search_all): what the byte-compatible
 * modes replay — of any size since round 4 (rounds 1-3: at most XGM_MAX_K documents, fetched as one page and sorted here).  The
 * buffer is sized by the tree's own upper bound (plan.est_max = PostList::get_termfreq_max of the tree the reference builds), left
 * uninitialised: only the matches are ever touched.  Returns the library's code. */
int fetch_all(xgm_index* idx, const xgm_query& plan, Xapian::doccount doccount, std::unique_ptr<xgm_hit[]>* out, uint64_t* n, xgm_result_hdr* hdr) {
    /* sized by the match itself, not by the plan's upper bound (a frequent-term OR on 10 M documents would allocate 240 MB per call
     * for a few thousand matches): a first call with no room only counts; beyond the configured ceiling the search is left to the
     * CPU matcher (XGM_UNSUPPORTED) — set_replay_limit */
    (void)doccount;
    int rc = xgm_search_all(idx, &plan, nullptr, 0, n, hdr);
    if (rc != XGM_OK || *n == 0) return rc;
    if (*n > g_replay_limit.load(std::memory_order_relaxed)) return XGM_UNSUPPORTED;
    const uint64_t cap = *n;
    out->reset(new xgm_hit[cap]);
    rc = xgm_search_all(idx, &plan, out->get(), cap, n, hdr);
    if (rc == XGM_OK && *n > cap) rc = XGM_UNSUPPORTED;        /* (the shard changed between the two calls: cannot happen under the shard lock) */
    return rc;
}

}  // namespace

static void register_ptr(const std::string& uuid, Xapian::rev revision, std::shared_ptr<xgm_index> own, uint32_t batch) {
    xgm_index* idx = own.get();
    if (batch) xgm_index_set_batching(idx, batch);
    if (g_near_colocated.load(std::memory_order_relaxed)) xgm_index_set_near_colocated(idx, 1);
    Shard replaced;                                                /* (its index — if the registry's was the last reference — is closed outside the lock) */
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_shards.find(uuid);
        if (it != g_shards.end()) replaced = std::move(it->second);
        g_shards[uuid] = Shard{std::move(own), idx, revision, std::make_shared<ShardColumns>(), batch};
    }
}

void register_shard(const Xapian::Database& db, xgm_index* idx, uint32_t batch) {
    register_ptr(db.get_uuid(), db.get_revision(), std::shared_ptr<xgm_index>(idx, [](xgm_index*) {}), batch);      /* not owned: the caller closes it */
}

void register_shard_owned(const std::string& uuid, uint64_t revision, xgm_index* idx, uint32_t batch, std::function<void()> released) {
    register_ptr(uuid, (Xapian::rev)revision, std::shared_ptr<xgm_index>(idx, [released](xgm_index* p) { xgm_index_close(p); if (released) released(); }), batch);
}

void register_shard_owned(const Xapian::Database& db, xgm_index* idx, uint32_t batch, std::function<void()> released) {
    register_shard_owned(db.get_uuid(), db.get_revision(), idx, batch, std::move(released));
}

void unregister_shard(const Xapian::Database& db) {
    Shard gone;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_shards.find(db.get_uuid());
        if (it == g_shards.end()) return;
        gone = std::move(it->second);
        g_shards.erase(it);
    }
}

void set_enabled(bool on) { g_enabled.store(on); }
void set_positional_mode(PositionalMode m) { g_positional.store(int(m)); }
void set_collapse_mode(CollapseMode m) { g_collapse.store(int(m)); }
void set_exact_bounds(bool on) { g_exact_bounds.store(on); }
void set_near_colocated_terms(bool may_exist) {
    /* the shards run NearPostList's procedure in full from here on (xgm_index_set_near_colocated); POSITIONAL_INTENDED answers such
     * queries on the device, POSITIONAL_REFERENCE keeps them on the CPU matcher (lower_query) */
    g_near_colocated.store(may_exist);
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& kv : g_shards) xgm_index_set_near_colocated(kv.second.idx, may_exist ? 1 : 0);
}
void set_replay(bool on) { g_replay.store(on); }
void set_replay_limit(uint64_t max_matches) { g_replay_limit.store(max_matches); }
void set_column_build_limit(uint32_t max_documents) { g_column_limit.store(max_documents); }
void register_spy_adapter(const std::string& spy_class_name, SpyAdapter adapter) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_spy_adapters[spy_class_name] = std::make_shared<const SpyAdapter>(std::move(adapter));
}
Counters counters() {
    return Counters{g_answered.load(), g_shape.load(), g_unreg.load(), g_rev.load(), g_dev.load(), g_sorted.load(), g_spied.load(),
                    g_collapsed.load(), g_columns.load(), g_replayed.load(), g_combined.load(), g_combined_launches.load()};
}

bool try_get_mset(const Xapian::Database& db, const Xapian::Query& query, const Xapian::Weight::Internal& stats,
                  const Xapian::Weight& wtscheme, bool full_db_has_positions, Xapian::doccount first,
                  Xapian::doccount maxitems, Xapian::doccount check_at_least, const Xapian::MatchDecider* mdecider,
                  const Xapian::KeyMaker* sorter, Xapian::valueno collapse_key, Xapian::doccount collapse_max, int percent_threshold,
                  double weight_threshold, Xapian::Enquire::docid_order order, Xapian::valueno sort_key, int sort_by,
                  bool sort_val_reverse, double time_limit,
                  const std::vector<Xapian::Internal::opt_intrusive_ptr<Xapian::MatchSpy>>& matchspies, Xapian::MSet& out) {
    typedef Xapian::Enquire::Internal EI;
    tl_replay.armed = false;
    if (!g_enabled.load(std::memory_order_relaxed)) return false;
    /* searches whose exact semantics live in the reference's own collation (set_replay): left to get_local_mset, which will ask
     * maybe_replay for the device's match list */
    {
        const bool basic = db.size() == 1 && !mdecider && time_limit == 0.0 && stats.rset_size == 0 && order != Xapian::Enquire::DESCENDING &&
                           wtscheme.name() == "Xapian::BM25Weight";
        const bool by_collapse = collapse_max != 0 && g_collapse.load(std::memory_order_relaxed) == COLLAPSE_REFERENCE;
        const bool by_cutoff = (percent_threshold != 0 || weight_threshold != 0.0) && g_replay.load(std::memory_order_relaxed);
        if (basic && (by_collapse || by_cutoff)) {
            tl_replay = ReplayCtx{true, &stats, &wtscheme, full_db_has_positions};
            return false;
        }
    }
    /* eligibility (widened by row (f).3) */
    const bool by_rel = sort_by == int(EI::REL);
    const bool by_value = sort_by == int(EI::VAL) || sort_by == int(EI::VAL_REL) || sort_by == int(EI::REL_VAL);
    if (db.size() != 1 || !(by_rel || by_value) || order == Xapian::Enquire::DESCENDING || percent_threshold != 0 || weight_threshold != 0.0 ||
        mdecider || stats.rset_size != 0 || time_limit != 0.0 || wtscheme.name() != "Xapian::BM25Weight" ||
        (collapse_max != 0 && (g_collapse.load(std::memory_order_relaxed) == COLLAPSE_DECLINE || !matchspies.empty())) || (by_rel && sorter)) {
        ++g_shape;
        return false;
    }
    /* spies: only classes the hook can feed */
    struct SpyPlan { Xapian::MatchSpy* spy; Xapian::valueno slot; std::shared_ptr<const SpyAdapter> adapter; std::shared_ptr<Column> col; };
    std::vector<SpyPlan> spies;
    for (const auto& sp : matchspies) {
        if (!sp.get()) continue;
        SpyPlan pl{sp.get(), 0, nullptr, nullptr};
        if (!spy_slot_of(*sp, &pl.slot, &pl.adapter)) { ++g_shape; return false; }
        spies.push_back(pl);
    }
    Shard sh;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_shards.find(db.get_uuid());
        if (it == g_shards.end()) { ++g_unreg; return false; }
        sh = it->second;
    }
    if (sh.revision != db.get_revision()) { ++g_rev; return false; }      /* the segment is of another revision: CPU until refreshed */
    Lowered L;
    L.idx = sh.idx;
    if (!lower(query, &L)) { ++g_shape; return false; }

    /* BM25 parameters: the scheme's own serialisation (bm25weight.cc:145-153) */
    {
        const std::string ser = wtscheme.serialise();
        const char* p = ser.data();
        const char* end = p + ser.size();
        L.d.k1 = unserialise_double(&p, end); L.d.k2 = unserialise_double(&p, end); L.d.k3 = unserialise_double(&p, end);
        L.d.b = unserialise_double(&p, end); L.d.min_normlen = unserialise_double(&p, end);
    }
    L.d.first = first; L.d.maxitems = maxitems; L.d.check_at_least = check_at_least;
    /* (reference mode needs exact positional semantics first: the pruned search's page is the intended one and only a hint whether the
     * page fills; the replay below replaces it) */

    /* merged statistics, exactly what the CPU matcher would weigh with (weightinternal.h:72-111) */
    xgm_global_stats gs;
    memset(&gs, 0, sizeof gs);
    gs.total_length = stats.total_length;
    gs.collection_size = stats.collection_size;
    gs.full_db_has_positions = full_db_has_positions ? 1u : 0u;
    for (size_t i = 0; i < L.terms.size(); ++i) {
        auto it = stats.termfreqs.find(L.terms[i]);
        if (it != stats.termfreqs.end()) { gs.termfreq[i] = it->second.termfreq; continue; }
        /* a term out of a wildcard expansion that the query does not name: the matcher registers it with THIS shard's
         * frequencies when it opens the posting list (LocalSubMatch::open_post_list, lazy_weight: localsubmatch.cc:283-292) */
        uint32_t tf = 0;
        if (!(i < L.lazy.size() && L.lazy[i]) || xgm_lookup_term(sh.idx, L.terms[i].data(), L.terms[i].size(), nullptr, &tf, nullptr, nullptr) != XGM_OK) { ++g_shape; return false; }
        gs.termfreq[i] = tf;
    }

    const uint32_t k = first + maxitems;
    std::vector<xgm_hit> hits(k ? k : 1);
    std::vector<uint32_t> hit_ord, hit_cord, hit_ccount;
    xgm_result_hdr hdr;
    memset(&hdr, 0, sizeof hdr);
    xgm_query plan;
    int rc = xgm_plan_query(sh.idx, &L.d, &gs, &plan);
    if (rc > 0) { ++g_dev; return false; }
    if (rc < 0) throw Xapian::DatabaseError(std::string("xgm: ") + xgm_last_error());
    if (L.d.op == XGM_OP_TREE) L.total_subqs = plan.total_subqs;

    const bool plain = by_rel && spies.empty() && collapse_max == 0;
    std::shared_ptr<Column> sort_col, collapse_col;
    uint64_t collapsed_lb = 0;
    /* The byte-compatible modes ride in the SAME call as every other search (round 6; round 5: a second, one-query-at-a-time call behind it):
     * the plan carries XGM_REPLAY_BATCH_* bits, the library answers with the reference's own collation — and, with xgm_index_set_batching on, in
     * the dispatcher's shared launches.  POSITIONAL_REFERENCE: the page ProtoMSet + SelectPostList's frozen weight leave (selectpostlist.cc:28-55)
     * and its known_matching_docs; exact bounds: known_matching_docs of any other shape (protomset.h:340-400). */
    const bool positional = L.d.op == XGM_OP_PHRASE || L.d.op == XGM_OP_NEAR;
    const bool pos_ref = plain && positional && plan.phrase_active && g_positional.load(std::memory_order_relaxed) == POSITIONAL_REFERENCE && k > 0;
    const bool want_count = plain && !positional && g_exact_bounds.load(std::memory_order_relaxed) && k > 0;
    uint64_t known_raw = 0;
    std::string combo_error;                        /* the library's message when the search failed on another thread (the leader of a combined launch) */
    if (plain) {
        /* (POSITIONAL_REFERENCE alone: the reference's page — ranks, docids, weights; with set_exact_bounds its match-count figures too, for which every
         * document of the conjunction has to be tested: ~9 x the time on frequent-term phrases) */
        plan.replay = pos_ref ? (XGM_REPLAY_BATCH_FROZEN | (g_exact_bounds.load(std::memory_order_relaxed) ? XGM_REPLAY_BATCH_COUNT : 0u)) : want_count ? XGM_REPLAY_BATCH_COUNT : 0u;
        rc = plan.replay ? xgm_search_batch_known(sh.idx, &plan, 1, k ? k : 1, hits.data(), &hdr, &known_raw)
                         : xgm_search_batch(sh.idx, &plan, 1, k ? k : 1, hits.data(), &hdr);
    } else {
        if (k == 0 || plan.max_possible == 0.0) { ++g_shape; return false; }      /* (max_possible == 0: the matcher renormalises the sort, matcher.cc:421-434) */
        /* the columns this search ranks, counts and collapses by */
        xgm_sort_spec spec;
        memset(&spec, 0, sizeof spec);
        if (by_value) {
            sort_col = sorter ? key_column(sh, db, *sorter) : value_column(sh, db, sort_key);
            if (!sort_col) { ++g_shape; return false; }
            spec.sort_by = sort_by == int(EI::VAL) ? XGM_SORT_VALUE : sort_by == int(EI::VAL_REL) ? XGM_SORT_VALUE_RELEVANCE : XGM_SORT_RELEVANCE_VALUE;
            spec.slot = sort_col->slot_id;
            spec.reverse = sort_val_reverse ? 1u : 0u;
            hit_ord.resize(k);
        }
        for (SpyPlan& sp : spies) {
            sp.col = value_column(sh, db, sp.slot);
            if (!sp.col) { ++g_shape; return false; }
        }
        const xgm_sort_spec* sp_sort = by_value ? &spec : nullptr;
        /* one search of this thread — in a launch with those other threads issue under the same sort / spy slot / collapse key meanwhile when the
         * shard was registered with batching on (combo_search above), else on its own */
        auto search = [&](uint32_t kind, uint32_t aux_slot, uint32_t aux_max, xgm_hit* h, uint32_t* o, uint32_t* co, uint32_t* cc, xgm_result_hdr* hd,
                          uint64_t* clb, uint32_t* cnt, uint32_t n_cnt) {
            const ComboKey K{sh.idx, kind, sp_sort ? 1u : 0u, spec.sort_by, spec.slot, spec.reverse, aux_slot, aux_max};
            ComboReq r{&plan, k, h, o, co, cc, hd, clb, cnt, n_cnt, 0, std::string()};
            static const bool no_combine = getenv("XGM_HOOK_NO_COMBINE") != nullptr;        /* A/B switch: every search its own launch */
            const int rc_ = (sh.batch > 1u && !no_combine) ? combo_search(K, r) : (r.rc = combo_single(K, sp_sort, r));
            if (rc_ < 0) combo_error = r.error;
            return rc_;
        };
        if (collapse_max != 0) {
            collapse_col = value_column(sh, db, collapse_key);
            if (!collapse_col) { ++g_shape; return false; }
            hit_cord.resize(k); hit_ccount.resize(k);
            rc = search(2u, collapse_col->slot_id, collapse_max, hits.data(), by_value ? hit_ord.data() : nullptr, hit_cord.data(), hit_ccount.data(), &hdr,
                        &collapsed_lb, nullptr, 0);
        } else if (spies.empty()) {
            rc = search(0u, 0u, 0u, hits.data(), hit_ord.data(), nullptr, nullptr, &hdr, nullptr, nullptr, 0);
        } else {
            /* one pass per spy (each counts one column); the first pass's page is the answer */
            std::vector<std::vector<uint32_t>> counts(spies.size());
            for (size_t i = 0; i < spies.size() && rc == XGM_OK; ++i) {
                counts[i].assign(spies[i].col->values.size() + 1, 0u);
                std::vector<xgm_hit> h2(i ? k : 0);
                xgm_result_hdr hd2;
                rc = search(1u, spies[i].col->slot_id, 0u, i ? h2.data() : hits.data(), i ? nullptr : (by_value ? hit_ord.data() : nullptr), nullptr, nullptr,
                            i ? &hd2 : &hdr, nullptr, counts[i].data(), (uint32_t)counts[i].size());
            }
            /* A spy is shown every matching document when the value leads the sort (ProtoMSet::early_reject, protomset.h:249-283:
             * min_weight stays 0) or when the match does not exceed check_at_least (min_weight is raised only once checked_enough());
             * otherwise what it sees depends on the CPU matcher's traversal: leave the search to it. */
            const bool value_leads = sort_by == int(EI::VAL) || sort_by == int(EI::VAL_REL);
            if (rc == XGM_OK && !value_leads && XGM_MATCHES_COUNT(hdr.matches_exact) > plan.check_at_least) {
                if (g_replay.load(std::memory_order_relaxed) && sort_by == int(EI::REL)) tl_replay = ReplayCtx{true, &stats, &wtscheme, full_db_has_positions};
                else ++g_shape;
                return false;
            }
            if (rc == XGM_OK) {
                for (size_t i = 0; i < spies.size(); ++i) {
                    const std::vector<std::string>& vals = spies[i].col->values;
                    const Xapian::doccount total = (Xapian::doccount)XGM_MATCHES_COUNT(hdr.matches_exact);
                    if (spies[i].adapter) {
                        std::vector<std::pair<std::string, Xapian::doccount>> cv;
                        for (size_t o = 1; o < counts[i].size(); ++o) if (counts[i][o]) cv.emplace_back(vals[o - 1], counts[i][o]);
                        spies[i].adapter->feed(*spies[i].spy, total, cv);
                    } else {
                        /* ValueCountMatchSpy::merge_results (api/matchspy.cc:381-403): total, then (value, frequency) pairs */
                        std::string ser;
                        pack_uint(ser, total);
                        for (size_t o = 1; o < counts[i].size(); ++o) if (counts[i][o]) { pack_string(ser, vals[o - 1]); pack_uint(ser, (Xapian::doccount)counts[i][o]); }
                        spies[i].spy->merge_results(ser);
                    }
                }
            }
        }
    }
    if (rc > 0) { ++g_dev; return false; }                                   /* declined by the device path: CPU matcher */
    if (rc < 0) throw Xapian::DatabaseError(std::string("xgm: ") + (combo_error.empty() ? std::string(xgm_last_error()) : combo_error));

    /* (the figures of a row that carried replay bits: ProtoMSet's own count, exact unless flagged) */
    const bool replayed = plain && plan.replay != 0u && !(known_raw & XGM_KNOWN_LOWER_BOUND);
    const uint64_t replay_known = known_raw & ~XGM_KNOWN_LOWER_BOUND;
    if (replayed && hdr.n_hits == k) ++g_replayed;

    /* the MSet, as ProtoMSet::finalise builds it (protomset.h:466-471, 484-682) */
    std::vector<Result> items;
    const uint32_t skip = std::min<uint32_t>(first, hdr.n_hits);
    items.reserve(hdr.n_hits - skip);
    for (uint32_t i = skip; i < hdr.n_hits; ++i) {
        items.emplace_back(hits[i].weight, hits[i].docid);
        if (sort_col && hit_ord[i]) items.back().set_sort_key(sort_col->values[hit_ord[i] - 1]);
        if (collapse_col && hit_cord[i]) { items.back().set_collapse_key(collapse_col->values[hit_cord[i] - 1]); items.back().set_collapse_count(hit_ccount[i]); }
    }
    double percent_scale = 0.0;
    if (hdr.n_hits && hdr.max_attained > 0.0 && L.total_subqs) {
        percent_scale = hdr.max_weight_subqs_matched / double(L.total_subqs);
        percent_scale /= hdr.max_attained;
    }
    /* bounds and estimate as ProtoMSet::finalise derives them from the tree's static termfreq bounds and known_matching_docs
     * (protomset.h:484-619).  Where the reference's main loop shows ProtoMSet every matching document — a sort the value leads —
     * known_matching_docs IS the match count and the three figures are the reference's; by relevance the count of documents
     * returned stands in for it (what known_matching_docs is under weight pruning). */
    uint32_t lb = 0, est = 0, ub = 0;
    const uint64_t m_all = XGM_MATCHES_COUNT(hdr.matches_exact);
    if (replayed) {
        /* known_matching_docs is a function of the match in docid order with its weights — for the operators that visit every match (a term,
         * AND, FILTER, AND_NOT, PHRASE, NEAR) and for those whose tree prunes by weight (OR, AND_MAYBE, nested trees: it only ever skips documents
         * the matcher's loop would drop anyway, matcher.cc:500-505) alike: the device counted as ProtoMSet would */
        xgm_mset_bounds_known(&plan, &hdr, replay_known, &lb, &est, &ub);
    } else if (sort_by == int(EI::VAL) || sort_by == int(EI::VAL_REL)) {
        xgm_mset_bounds_known(&plan, &hdr, m_all, &lb, &est, &ub);
    } else {
        xgm_mset_bounds(&plan, &hdr, &lb, &est, &ub);
    }
    uint32_t ulb = lb, uest = est, uub = ub;
    if (collapse_max != 0) {
        /* ProtoMSet::finalise with a collapser that considered every matching document (protomset.h:497-619): docs_considered = the
         * match, dups_ignored = the documents beyond collapse_max of their key, lower bound = what stays (collapser.cc:212-218) */
        uint32_t slb = plan.est_min, sest = plan.est_est, sub = plan.est_max;
        ulb = std::max<uint32_t>(slb, (uint32_t)std::min<uint64_t>(m_all, 0xFFFFFFFFu)); uest = sest; uub = sub;
        if (hdr.n_hits < k) {
            lb = est = ub = hdr.n_hits;
            ulb = slb;
        } else {
            const uint64_t dups = m_all - collapsed_lb;
            lb = (uint32_t)collapsed_lb;
            ub = sub - (uint32_t)std::min<uint64_t>(dups, sub);
            const double unique_rate = m_all ? double(m_all - dups) / double(m_all) : 1.0;
            est = unique_rate != 1.0 ? Xapian::doccount(sest * unique_rate + 0.5) : sest;
            if (est < lb) est = lb;
            est = std::min(std::max(est, lb), std::max(ub, lb));
            if (ub < lb) ub = lb;
        }
        if (lb > ulb) ulb = lb;
        uest = std::min(std::max(uest, ulb), std::max(uub, ulb));
        ++g_collapsed;
    }
    out = Xapian::MSet(new Xapian::MSet::Internal(first, ub, lb, est, uub, ulb, uest, hdr.max_possible, hdr.max_attained, std::move(items),
                                                   percent_scale * 100.0));
    ++g_answered;
    if (by_value) ++g_sorted;
    if (!spies.empty()) ++g_spied;
    return true;
}

Xapian::Internal::PostList* maybe_replay(const Xapian::Database& db, const Xapian::Query& query, Xapian::Internal::PostList* pl) {
    const ReplayCtx ctx = tl_replay;
    tl_replay.armed = false;
    if (!ctx.armed || !pl) return pl;
    Shard sh;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_shards.find(db.get_uuid());
        if (it == g_shards.end()) { ++g_unreg; return pl; }
        sh = it->second;
    }
    if (sh.revision != db.get_revision()) { ++g_rev; return pl; }
    Lowered L;
    L.idx = sh.idx;
    if (!lower(query, &L)) { ++g_shape; return pl; }
    const bool positional = L.d.op == XGM_OP_PHRASE || L.d.op == XGM_OP_NEAR;
    if (positional && g_positional.load(std::memory_order_relaxed) != POSITIONAL_INTENDED) { ++g_shape; return pl; }   /* the frozen weight lives in the tree */
    {
        const std::string ser = ctx.wt->serialise();
        const char* p = ser.data();
        const char* end = p + ser.size();
        L.d.k1 = unserialise_double(&p, end); L.d.k2 = unserialise_double(&p, end); L.d.k3 = unserialise_double(&p, end);
        L.d.b = unserialise_double(&p, end); L.d.min_normlen = unserialise_double(&p, end);
    }
    xgm_global_stats gs;
    memset(&gs, 0, sizeof gs);
    gs.total_length = ctx.stats->total_length;
    gs.collection_size = ctx.stats->collection_size;
    gs.full_db_has_positions = ctx.full_db_has_positions ? 1u : 0u;
    for (size_t i = 0; i < L.terms.size(); ++i) {
        auto it = ctx.stats->termfreqs.find(L.terms[i]);
        if (it != ctx.stats->termfreqs.end()) { gs.termfreq[i] = it->second.termfreq; continue; }
        uint32_t tf = 0;
        if (!(i < L.lazy.size() && L.lazy[i]) || xgm_lookup_term(sh.idx, L.terms[i].data(), L.terms[i].size(), nullptr, &tf, nullptr, nullptr) != XGM_OK) { ++g_shape; return pl; }
        gs.termfreq[i] = tf;
    }
    /* the whole match in docid order (positional: every candidate's positions tested) */
    L.d.first = 0; L.d.maxitems = 1; L.d.check_at_least = 0xFFFFFFFFu;
    xgm_query plan;
    xgm_result_hdr hdr;
    memset(&hdr, 0, sizeof hdr);
    int rc = xgm_plan_query(sh.idx, &L.d, &gs, &plan);
    std::unique_ptr<xgm_hit[]> all;
    uint64_t m = 0;
    if (rc == XGM_OK) rc = fetch_all(sh.idx, plan, db.get_doccount(), &all, &m, &hdr);
    if (rc != XGM_OK || m == 0) { ++g_dev; return pl; }  /* declined, nothing to gain, or a device failure — which costs nothing here: the tree is still there, the CPU matcher runs */
    /* the static figures of the tree this list stands in for */
    Xapian::Internal::PostList* r = new ReplayPostList(std::move(all), (size_t)m, pl->get_termfreq_min(), pl->get_termfreq_est(), pl->get_termfreq_max(), pl->recalc_maxweight());
    delete pl;
    ++g_replayed;
    ++g_answered;
    return r;
}

}  // namespace xgm_hook
