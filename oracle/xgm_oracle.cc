/* xgm_oracle — CPU restatement of the reference's match/rank path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (xapiand_amd/, include/) never does.  It is pinned against the REAL reference
 * (oracle/_ref/xapian_ref, built from /root/reference) by tests/test_oracle_vs_reference.py and by
 * the committed fixtures under tests/golden/ which that binary generated.
 *
 * What is restated, with the reference lines each part follows (paths under src/xapian/):
 *   - glass posting-list chunk format + iteration:   backends/glass/glass_postlist.cc:677-695 (format),
 *       :110-124 (read_did_increase/read_wdf), :768-829 (next_in_chunk/next_chunk), :933-991 (skip_to),
 *       CHUNKSIZE 2000 (:224); varints common/pack.h:296-310, 325-389
 *   - doclen lookups through the doclen list:          glass_postlist.cc:194-205, 994-1021
 *   - BM25 term weight and per-document weight:        weight/bm25weight.cc:46-130, 170-181
 *   - AND: MultiAndPostList ordering + leapfrog + sum: matcher/multiandpostlist.h:117-130,
 *                                                      multiandpostlist.cc:150-160, 180-207
 *   - OR: Huffman tree of OrPostLists, l + r weights:  api/queryinternal.cc:440-489 with common/heap.h,
 *                                                      matcher/orpostlist.cc:94-103
 *         (exhaustive merge: the reference's MaxScore-style decay, orpostlist.cc:35-78, never changes
 *          the top-k set or the weights, only how many documents get scored)
 *   - PHRASE: ExactPhrasePostList / PhrasePostList:    matcher/exactphrasepostlist.cc:75-133,
 *                                                      matcher/phrasepostlist.cc:60-90
 *   - AND_NOT / AND_MAYBE / FILTER (left AND of terms, right terms): api/queryinternal.cc:2208-2283,
 *                                                      matcher/andmaybepostlist.cc:57-64, andnotpostlist.cc
 *   - top-k: ProtoMSet::add min-heap + final sort:      matcher/protomset.h:340-400, 657;
 *                                                      order matcher/msetcmp.cc:55-62
 *   - multi-shard protocol: merged stats, unshard, merge: api/enquire.cc:385-394, backends/multi.h:69-73,
 *                                                      matcher/matcher.cc:653-743
 *   - value sorts (widening, not on the device yet):   matcher/msetcmp.cc:64-101 (by value / value then relevance / relevance
 *       then value, both directions), matcher/matcher.cc:482-536 + protomset.h:249-283 (every matching document reaches
 *       update_max_weight; with the value leading there is no weight pruning): xgo_search_sorted (one shard) / _sorted_g (a shard
 *       of several, merged statistics); collapse (matcher/collapser.cc) with the INTENDED semantics — the snapshot's own is a
 *       quirk, DESIGN.md 7.3
 *   - a ValueCountMatchSpy over the whole match:       api/matchspy.cc:307-313, fed by matcher/matcher.cc:519-527 and
 *       protomset.h:268-275: xgo_search_spy
 * Plus the deterministic synthetic corpus of tools/xgm_corpus.h and its inversion to raw postings.
 *
 * Build: g++ -O2 -ffp-contract=off -shared -fPIC (oracle/Makefile).  C ABI for ctypes.
 */
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../tools/xgm_corpus.h"

namespace {

/* ----------------------------------------------------------------------------- corpus ---------- */

struct Corpus {
    uint32_t lastdocid = 0, doccount = 0;
    uint64_t total_length = 0;
    bool has_positions = false;
    std::vector<uint32_t> doclen;               /* [lastdocid+1] */
    std::vector<std::string> terms;             /* sorted bytewise */
    std::vector<uint32_t> term_len;
    std::vector<const char*> term_ptr;
    std::vector<uint32_t> df;
    std::vector<uint64_t> term_start;           /* [n_terms+1] into did/wdf */
    std::vector<uint32_t> did, wdf;
    std::vector<uint64_t> pos_off;
    std::vector<uint32_t> pos;
};

Corpus* corpus_build(uint64_t seed, uint64_t n_docs_global, uint32_t vocab, uint32_t len_lo, uint32_t len_hi,
                     uint32_t n_shards, uint32_t shard, bool with_positions) {
    xgm_corpus_params cp{seed, vocab, len_lo, len_hi};
    std::vector<uint64_t> thr(vocab);
    xgm_zipf_thresholds(vocab, thr.data());
    /* tokens as (rank, local doc, pos) triples, sorted → postings */
    struct Tok { uint32_t rank, doc; uint32_t pos; };
    std::vector<Tok> toks;
    Corpus* c = new Corpus();
    c->has_positions = with_positions;
    uint32_t local = 0;
    c->doclen.push_back(0);
    for (uint64_t g = 1; g <= n_docs_global; ++g) {
        if ((g - 1) % n_shards != shard) continue;
        ++local;
        uint32_t len = xgm_doc_len(&cp, g);
        c->doclen.push_back(len);
        c->total_length += len;
        for (uint32_t p = 1; p <= len; ++p) toks.push_back(Tok{xgm_token(&cp, thr.data(), g, p), local, p});
    }
    c->lastdocid = c->doccount = local;
    /* term order = bytewise order of "t<rank>" */
    std::vector<uint32_t> ranks;
    {
        std::vector<uint8_t> seen(vocab + 1, 0);
        for (auto& t : toks) seen[t.rank] = 1;
        for (uint32_t r = 1; r <= vocab; ++r) if (seen[r]) ranks.push_back(r);
    }
    std::vector<std::string> names(ranks.size());
    for (size_t i = 0; i < ranks.size(); ++i) names[i] = "t" + std::to_string(ranks[i]);
    std::vector<uint32_t> perm(ranks.size());
    for (size_t i = 0; i < perm.size(); ++i) perm[i] = (uint32_t)i;
    std::sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return names[a] < names[b]; });
    std::vector<uint32_t> rank2id(vocab + 1, UINT32_MAX);
    for (size_t i = 0; i < perm.size(); ++i) {
        c->terms.push_back(names[perm[i]]);
        rank2id[ranks[perm[i]]] = (uint32_t)i;
    }
    std::sort(toks.begin(), toks.end(), [&](const Tok& a, const Tok& b) {
        uint32_t ia = rank2id[a.rank], ib = rank2id[b.rank];
        if (ia != ib) return ia < ib;
        if (a.doc != b.doc) return a.doc < b.doc;
        return a.pos < b.pos;
    });
    c->df.assign(c->terms.size(), 0);
    c->term_start.assign(c->terms.size() + 1, 0);
    if (with_positions) c->pos_off.push_back(0);
    for (size_t i = 0; i < toks.size();) {
        size_t j = i;
        while (j < toks.size() && toks[j].rank == toks[i].rank && toks[j].doc == toks[i].doc) ++j;
        uint32_t id = rank2id[toks[i].rank];
        c->did.push_back(toks[i].doc);
        c->wdf.push_back((uint32_t)(j - i));
        c->df[id]++;
        if (with_positions) {
            for (size_t q = i; q < j; ++q) c->pos.push_back(toks[q].pos);
            c->pos_off.push_back(c->pos.size());
        }
        i = j;
    }
    for (size_t t = 0; t < c->terms.size(); ++t) c->term_start[t + 1] = c->term_start[t] + c->df[t];
    for (auto& s : c->terms) { c->term_len.push_back((uint32_t)s.size()); }
    for (auto& s : c->terms) c->term_ptr.push_back(s.data());
    return c;
}

/* The same corpus inverted on the HOST for a handful of terms only — what lets a test at a configuration's full size (10 M documents) feed the oracle
 * from tools/xgm_corpus.h directly instead of from postings read back from the device.  Every document is generated (doclen and
 * statistics are the whole shard's), only the tokens of `ranks` are kept: two passes over the documents on n_threads threads — count per (thread, term),
 * then place — so that no global sort of a billion tokens is needed: a thread's documents ascend, the threads' slices follow each other. */
Corpus* corpus_build_terms(uint64_t seed, uint64_t n_docs_global, uint32_t vocab, uint32_t len_lo, uint32_t len_hi, uint32_t n_shards, uint32_t shard,
                           bool with_positions, const uint32_t* ranks, uint32_t n_ranks, uint32_t n_threads) {
    xgm_corpus_params cp{seed, vocab, len_lo, len_hi};
    std::vector<uint64_t> thr(vocab);
    xgm_zipf_thresholds(vocab, thr.data());
    /* term order = bytewise order of "t<rank>" */
    std::vector<uint32_t> rk(ranks, ranks + n_ranks);
    std::sort(rk.begin(), rk.end());
    rk.erase(std::unique(rk.begin(), rk.end()), rk.end());
    std::vector<std::string> names(rk.size());
    for (size_t i = 0; i < rk.size(); ++i) names[i] = "t" + std::to_string(rk[i]);
    std::vector<uint32_t> perm(rk.size());
    for (size_t i = 0; i < perm.size(); ++i) perm[i] = (uint32_t)i;
    std::sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return names[a] < names[b]; });
    std::vector<uint32_t> rank2id(vocab + 1, UINT32_MAX);
    Corpus* c = new Corpus();
    c->has_positions = with_positions;
    for (size_t i = 0; i < perm.size(); ++i) { c->terms.push_back(names[perm[i]]); rank2id[rk[perm[i]]] = (uint32_t)i; }
    const size_t NT = c->terms.size();
    const uint64_t n_local = n_docs_global > shard ? (n_docs_global - shard + n_shards - 1) / n_shards : 0;
    c->lastdocid = c->doccount = (uint32_t)n_local;
    c->doclen.assign(n_local + 1, 0);
    n_threads = std::max(1u, std::min<uint32_t>(n_threads, (uint32_t)std::max<uint64_t>(1, n_local / 1024)));
    /* thread t owns local documents (lo_t, hi_t]: global g = (local - 1) * n_shards + shard + 1 */
    auto lo_of = [&](uint32_t t) { return n_local * t / n_threads; };
    /* pass 1: per (thread, term) postings and positions; document lengths */
    std::vector<std::vector<uint64_t>> n_post(n_threads, std::vector<uint64_t>(NT, 0)), n_pos(n_threads, std::vector<uint64_t>(NT, 0));
    std::vector<uint64_t> len_sum(n_threads, 0);
    auto walk = [&](uint32_t t, auto&& on_doc_term) {
        std::vector<uint32_t> cnt(NT, 0), touched;
        std::vector<std::vector<uint32_t>> ppos(NT);
        for (uint64_t local = lo_of(t) + 1; local <= lo_of(t + 1); ++local) {
            const uint64_t g = (local - 1) * n_shards + shard + 1;
            const uint32_t len = xgm_doc_len(&cp, g);
            c->doclen[local] = len;
            touched.clear();
            for (uint32_t p = 1; p <= len; ++p) {
                const uint32_t id = rank2id[xgm_token(&cp, thr.data(), g, p)];
                if (id == UINT32_MAX) continue;
                if (cnt[id]++ == 0) touched.push_back(id);
                if (with_positions) ppos[id].push_back(p);
            }
            for (uint32_t id : touched) { on_doc_term(id, (uint32_t)local, cnt[id], ppos[id]); cnt[id] = 0; ppos[id].clear(); }
        }
    };
    {
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < n_threads; ++t)
            th.emplace_back([&, t]() {
                uint64_t ls = 0;
                walk(t, [&](uint32_t id, uint32_t, uint32_t wdf, const std::vector<uint32_t>&) { ++n_post[t][id]; n_pos[t][id] += wdf; });
                for (uint64_t local = lo_of(t) + 1; local <= lo_of(t + 1); ++local) ls += c->doclen[local];
                len_sum[t] = ls;
            });
        for (auto& x : th) x.join();
    }
    for (uint32_t t = 0; t < n_threads; ++t) c->total_length += len_sum[t];
    /* offsets: term-major, then thread */
    c->df.assign(NT, 0);
    c->term_start.assign(NT + 1, 0);
    std::vector<std::vector<uint64_t>> o_post(n_threads, std::vector<uint64_t>(NT, 0)), o_pos(n_threads, std::vector<uint64_t>(NT, 0));
    uint64_t np = 0, npos = 0;
    for (size_t id = 0; id < NT; ++id) {
        c->term_start[id] = np;
        for (uint32_t t = 0; t < n_threads; ++t) { o_post[t][id] = np; o_pos[t][id] = npos; np += n_post[t][id]; npos += n_pos[t][id]; c->df[id] += (uint32_t)n_post[t][id]; }
    }
    c->term_start[NT] = np;
    c->did.assign(np, 0); c->wdf.assign(np, 0);
    if (with_positions) { c->pos_off.assign(np + 1, 0); c->pos.assign(npos, 0); }
    /* pass 2: place */
    {
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < n_threads; ++t)
            th.emplace_back([&, t]() {
                std::vector<uint64_t> at = o_post[t], pat = o_pos[t];
                walk(t, [&](uint32_t id, uint32_t local, uint32_t wdf, const std::vector<uint32_t>& pp) {
                    const uint64_t i = at[id]++;
                    c->did[i] = local; c->wdf[i] = wdf;
                    if (with_positions) { c->pos_off[i] = pat[id]; for (uint32_t x : pp) c->pos[pat[id]++] = x; }
                });
            });
        for (auto& x : th) x.join();
    }
    if (with_positions) c->pos_off[np] = npos;
    for (auto& s : c->terms) { c->term_len.push_back((uint32_t)s.size()); }
    for (auto& s : c->terms) c->term_ptr.push_back(s.data());
    return c;
}

/* ----------------------------------------------------------------------------- glass lists ----- */

void put_varint(std::vector<uint8_t>& out, uint64_t v) {         /* pack_uint, common/pack.h:296-310 */
    while (v >= 128) { out.push_back((uint8_t)(v | 0x80)); v >>= 7; }
    out.push_back((uint8_t)v);
}
inline uint32_t get_varint(const uint8_t*& p) {                  /* unpack_uint, common/pack.h:325-389 */
    uint32_t v = 0; int sh = 0;
    while (true) { uint8_t b = *p++; v |= (uint32_t)(b & 0x7F) << sh; if (!(b & 0x80)) break; sh += 7; }
    return v;
}

/* One posting list in glass's chunk layout: each chunk is `varint(wdf0) {varint(gap-1) varint(wdf)}*`
 * with first/last docid kept beside it (the reference keeps them in the B-tree key and chunk header). */
struct GlassList {
    struct Chunk { uint32_t first, last; uint32_t off, end; };
    std::vector<Chunk> chunks;
    std::vector<uint8_t> bytes;
    uint32_t termfreq = 0;
    static constexpr size_t CHUNKSIZE = 2000;                    /* glass_postlist.cc:224 */
    void build(const uint32_t* did, const uint32_t* wdf, uint32_t n) {
        termfreq = n;
        uint32_t i = 0;
        while (i < n) {
            Chunk c;
            c.first = did[i];
            c.off = (uint32_t)bytes.size();
            put_varint(bytes, wdf[i]);
            uint32_t prev = did[i];
            ++i;
            while (i < n && bytes.size() - c.off < CHUNKSIZE) {
                put_varint(bytes, did[i] - prev - 1);
                put_varint(bytes, wdf[i]);
                prev = did[i];
                ++i;
            }
            c.last = prev;
            c.end = (uint32_t)bytes.size();
            chunks.push_back(c);
        }
    }
};

struct GlassIter {
    const GlassList* l = nullptr;
    size_t ci = 0;
    const uint8_t *p = nullptr, *e = nullptr;
    uint32_t did = 0, wdf = 0;
    bool ended = true, started = false;
    void init(const GlassList* list) { l = list; ended = list->chunks.empty(); started = false; ci = 0; did = 0; }
    void open_chunk(size_t i) {
        ci = i;
        const auto& c = l->chunks[i];
        p = l->bytes.data() + c.off; e = l->bytes.data() + c.end;
        did = c.first;
        wdf = get_varint(p);
    }
    bool at_end() const { return ended; }
    void next() {                                                /* GlassPostList::next, :853-872 */
        if (ended) return;
        if (!started) { started = true; open_chunk(0); return; }
        if (p < e) { did += get_varint(p) + 1; wdf = get_varint(p); return; }
        if (ci + 1 < l->chunks.size()) { open_chunk(ci + 1); return; }
        ended = true;
    }
    void skip_to(uint32_t target) {                              /* GlassPostList::skip_to, :959-991 */
        if (ended) return;
        if (!started) { started = true; open_chunk(0); }
        if (target <= did) return;
        if (target > l->chunks[ci].last) {
            /* B-tree find_entry(term || target): last chunk whose first docid <= target */
            size_t lo = ci, hi = l->chunks.size();
            while (hi - lo > 1) { size_t mid = (lo + hi) / 2; if (l->chunks[mid].first <= target) lo = mid; else hi = mid; }
            if (target > l->chunks[lo].last) {
                if (lo + 1 >= l->chunks.size()) { ended = true; return; }
                ++lo;
            }
            open_chunk(lo);
        }
        while (did < target) { did += get_varint(p) + 1; wdf = get_varint(p); }   /* move_forward_in_chunk_to_at_least */
    }
};

/* ----------------------------------------------------------------------------- index ----------- */

struct Index {
    uint32_t lastdocid = 0, doccount = 0, doclen_lb = 0, wdf_ub_db = 0;
    uint64_t total_length = 0;
    bool has_positions = false;
    std::map<std::string, uint32_t> dict;
    std::vector<uint32_t> df, cf, wdf_ub;
    std::vector<uint64_t> term_start;
    const uint32_t *did = nullptr, *wdf = nullptr, *pos = nullptr;
    const uint64_t* pos_off = nullptr;
    std::vector<GlassList> lists;            /* lazily built */
    std::vector<uint8_t> built;
    GlassList doclen_list;                   /* the special doclen posting list */
    std::vector<std::vector<std::string>> values;    /* [slot][docid]: Document::get_value (empty = none) */
    std::vector<uint32_t> doclen_dense;
    GlassList& list(uint32_t id) {
        if (!built[id]) { lists[id].build(did + term_start[id], wdf + term_start[id], df[id]); built[id] = 1; }
        return lists[id];
    }
};

Index* index_from_raw(uint32_t n_terms, uint32_t lastdocid, uint32_t doccount, uint64_t total_length, const uint32_t* doclen,
                      const char* const* terms, const uint32_t* term_len, const uint32_t* df, const uint32_t* did,
                      const uint32_t* wdf, const uint64_t* pos_off, const uint32_t* pos) {
    Index* ix = new Index();
    ix->lastdocid = lastdocid; ix->doccount = doccount; ix->total_length = total_length;
    ix->has_positions = pos_off != nullptr;
    ix->df.assign(df, df + n_terms);
    ix->did = did; ix->wdf = wdf; ix->pos = pos; ix->pos_off = pos_off;
    ix->term_start.assign(n_terms + 1, 0);
    uint64_t np = 0;
    for (uint32_t t = 0; t < n_terms; ++t) { ix->term_start[t] = np; np += df[t]; ix->dict[std::string(terms[t], term_len[t])] = t; }
    ix->term_start[n_terms] = np;
    for (uint64_t i = 0; i < np; ++i) ix->wdf_ub_db = std::max(ix->wdf_ub_db, wdf[i]);
    ix->cf.resize(n_terms); ix->wdf_ub.resize(n_terms);
    for (uint32_t t = 0; t < n_terms; ++t) {
        uint64_t c = 0;
        for (uint64_t i = ix->term_start[t]; i < ix->term_start[t + 1]; ++i) c += wdf[i];
        ix->cf[t] = (uint32_t)c;
        /* GlassPostListTable::get_freqs :175-189, GlassDatabase::get_wdf_upper_bound :823-830 */
        uint32_t first_wdf = wdf[ix->term_start[t]];
        uint32_t ub = (c == 0 || df[t] == 1) ? (uint32_t)c : std::max((uint32_t)c - first_wdf, first_wdf);
        ix->wdf_ub[t] = std::min(ub, ix->wdf_ub_db);
    }
    ix->lists.resize(n_terms); ix->built.assign(n_terms, 0);
    std::vector<uint32_t> dd, dl;
    for (uint32_t d = 1; d <= lastdocid; ++d) {
        if (doclen[d] == 0) continue;              /* synthetic corpora have no empty documents */
        dd.push_back(d); dl.push_back(doclen[d]);
        if (ix->doclen_lb == 0 || doclen[d] < ix->doclen_lb) ix->doclen_lb = doclen[d];
    }
    ix->doclen_list.build(dd.data(), dl.data(), (uint32_t)dd.size());
    ix->doclen_dense.assign(doclen, doclen + lastdocid + 1);
    return ix;
}

/* ----------------------------------------------------------------------------- weights --------- */

struct BM25 {
    double k1 = 1, k2 = 0, k3 = 1, b = 0.5, min_normlen = 0.5;
    double termweight = 0, len_factor = 0;
    void init(uint32_t collection_size, uint32_t tf, double avg_len, uint32_t wqf = 1, double factor = 1.0) {   /* bm25weight.cc:46-130 */
        double tw = (collection_size - tf + 0.5) / (tf + 0.5);
        if (tw < 2) tw = tw * 0.5 + 1;
        termweight = std::log(tw) * factor;
        if (k3 != 0) { double wqf_double = wqf; termweight *= (k3 + 1) * wqf_double / (k3 + wqf_double); }
        termweight *= (k1 + 1);
        if (k2 == 0 && (b == 0 || k1 == 0)) len_factor = 0;
        else { len_factor = avg_len; if (len_factor != 0) len_factor = 1 / len_factor; }
    }
    double sumpart(uint32_t wdf, uint32_t len) const {                       /* bm25weight.cc:170-181 */
        double normlen = std::max(len * len_factor, min_normlen);
        double wdf_double = wdf;
        double denom = k1 * (normlen * b + (1 - b)) + wdf_double;
        return termweight * (wdf_double / denom);
    }
    double maxpart(uint32_t wdf_ub, uint32_t doclen_lb) const {              /* bm25weight.cc:183-207 */
        double denom = k1;
        if (k1 != 0.0 && b != 0.0) {
            double normlen_lb = std::max(std::max(wdf_ub, doclen_lb) * len_factor, min_normlen);
            denom *= (normlen_lb * b + (1 - b));
        }
        double wdf_max = wdf_ub;
        denom += wdf_max;
        return termweight * (wdf_max / denom);
    }
};

struct Hit { uint32_t did; uint32_t subqs; double weight; };

inline bool mcmp(const Hit& a, const Hit& b) {                 /* msetcmp_by_relevance<true>, msetcmp.cc:55-62 */
    if (a.weight > b.weight) return true;
    if (a.weight < b.weight) return false;
    return a.did < b.did;
}

/* ProtoMSet::add without collapsing, check_at_least <= max_size (protomset.h:340-400) */
struct ProtoMSet {
    size_t max_size;
    std::vector<Hit> results;
    std::vector<uint32_t> heap;
    double min_weight = 0.0, max_weight = 0.0;
    uint32_t max_weight_subqs = 0;
    uint64_t known_matching_docs = 0;
    explicit ProtoMSet(size_t k) : max_size(k) {}
    struct Cmp { ProtoMSet* p; bool operator()(uint32_t a, uint32_t b) const { return mcmp(p->results[a], p->results[b]); } };
    void add(const Hit& item) {
        ++known_matching_docs;
        if (item.weight > max_weight) {                      /* update_max_weight, protomset.h:174-183 */
            max_weight = item.weight;
            max_weight_subqs = item.subqs;
        }
        if (item.weight < min_weight) return;
        if (results.size() < max_size) { results.push_back(item); return; }
        if (max_size == 0) return;
        if (heap.empty()) {
            for (uint32_t i = 0; i < results.size(); ++i) heap.push_back(i);
            std::make_heap(heap.begin(), heap.end(), Cmp{this});          /* worst item on top */
            min_weight = results[heap.front()].weight;
        }
        uint32_t worst = heap.front();
        if (!mcmp(item, results[worst])) return;
        results[worst] = item;
        std::pop_heap(heap.begin(), heap.end(), Cmp{this});
        heap.back() = worst;
        std::push_heap(heap.begin(), heap.end(), Cmp{this});
        min_weight = results[heap.front()].weight;
    }
    void finalise() { std::sort(results.begin(), results.end(), mcmp); }  /* protomset.h:657 */
};

/* ----------------------------------------------------------------------------- query ----------- */

struct QueryIn {
    uint32_t op;                 /* 1 AND, 2 OR, 3 PHRASE, 4 AND_NOT, 5 AND_MAYBE, 6 FILTER, 7 NEAR (oracle only so far) */
    uint32_t n_required;         /* ops 4-6: the first n_required terms are the left-hand AND, the others the right-hand side */
    uint32_t n_terms;
    const char* const* terms; const uint32_t* term_len;
    uint32_t window, first, maxitems;
    /* merged statistics (NULL-equivalent: use_global = 0) */
    uint32_t use_global; uint64_t g_total_length; uint32_t g_collection_size; uint32_t g_has_positions;
    const uint32_t* g_termfreq;
    /* 1: reproduce the reference's SelectPostList stale cached_weight (matcher/selectpostlist.cc:28-55:
     * vet() refreshes cached_weight through pltree->get_weight(), which re-enters
     * SelectPostList::get_weight and returns the PREVIOUS cached value once it is >= 0), so that the
     * oracle can be pinned to the real reference for PHRASE with k < matches.  0: the intended
     * semantics (true weights), which is what the device path implements.  DESIGN.md §7. */
    uint32_t select_cache_bug;
    /* widening row (f).3, sort by value (Enquire::set_sort_by_value*, api/enquire.cc; comparison functions matcher/msetcmp.cc:64-101):
     * 0 = relevance (default), 1 = VAL (value, docid), 2 = VAL_REL (value, weight, docid), 3 = REL_VAL (weight, value, docid);
     * sort_reverse = the API's `reverse` flag (false: smaller keys first; a document without the value has the empty key) */
    uint32_t sort_by = 0, sort_slot = 0, sort_reverse = 0;
    /* a Xapian::ValueCountMatchSpy on slot spy_slot_plus1 - 1 (api/matchspy.cc:307-313) as it counts when the matcher shows it every
     * matching document: check_at_least >= the matches, or the value leading the sort (ProtoMSet::early_reject still calls the
     * spies, protomset.h:268-275).  With weight pruning in play what a spy sees is a property of the reference's traversal. */
    uint32_t spy_slot_plus1 = 0;
    /* Enquire::set_collapse_key(slot, collapse_max): of the documents that share a (non-empty) value in the slot only the best
     * collapse_max — under the order in force — stay in the MSet (matcher/collapser.cc).  INTENDED semantics: every matching
     * document is considered (what the reference does when check_at_least covers the whole match; with less it stops testing
     * documents that cannot rank, and its counts depend on its traversal).  collapse_max = 0: off. */
    uint32_t collapse_slot = 0, collapse_max = 0;
};

struct Leaf { uint32_t tf, idx; };
struct TfAsc { bool operator()(const Leaf& a, const Leaf& b) const { return a.tf < b.tf; } };

/* common/heap.h sift-down (libc++), comparator "a.tf > b.tf" (queryinternal.cc:140-147) */
struct HItem { uint64_t tf; int node; };
inline bool hcomp(const HItem& a, const HItem& b) { return a.tf > b.tf; }
void sift_down(std::vector<HItem>& h, size_t len, size_t start) {
    if (len < 2 || (len - 2) / 2 < start) return;
    size_t child = 2 * start + 1;
    if (child + 1 < len && hcomp(h[child], h[child + 1])) ++child;
    if (hcomp(h[child], h[start])) return;
    HItem top = h[start];
    do {
        h[start] = h[child]; start = child;
        if ((len - 2) / 2 < child) break;
        child = 2 * child + 1;
        if (child + 1 < len && hcomp(h[child], h[child + 1])) ++child;
    } while (!hcomp(h[child], top));
    h[start] = top;
}

struct PosCursor { const uint32_t* p; uint32_t n, c; bool started;
    bool skip_to(uint32_t t) { if (!started) { started = true; c = 0; } while (c < n && p[c] < t) ++c; return c < n; }
    bool next() { if (!started) { started = true; c = 0; } else ++c; return c < n; }
    uint32_t get() const { return p[c]; } };

bool exact_phrase(std::vector<PosCursor>& pl, const std::vector<uint32_t>& wdfs) {
    /* exactphrasepostlist.cc:75-133 */
    size_t n = pl.size();
    std::vector<unsigned> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = (unsigned)i;
    std::sort(order.begin(), order.end(), [&](unsigned a, unsigned b) { return wdfs[a] < wdfs[b]; });
    std::vector<PosCursor*> pls(n);
    for (size_t i = 0; i < n; ++i) pls[i] = &pl[order[i]];
    if (!pls[0]->skip_to(order[0])) return false;
    if (pls[0]->n > pls[1]->n) {
        if (!pls[1]->skip_to(order[1])) return false;
        std::swap(pls[0], pls[1]); std::swap(order[0], order[1]);
    }
    uint32_t idx0 = order[0];
    uint32_t base = pls[0]->get() - idx0;
    unsigned i = 1;
    while (true) {
        uint32_t idx = order[i];
        uint32_t required = base + idx;
        if (!pls[i]->skip_to(required)) return false;
        uint32_t got = pls[i]->get();
        if (got == required) { if (++i == n) return true; continue; }
        if (!pls[0]->skip_to(got - idx + idx0)) return false;
        base = pls[0]->get() - idx0;
        i = 1;
    }
}

bool window_phrase(std::vector<PosCursor>& pl, uint32_t window) {
    /* phrasepostlist.cc:60-90 */
    size_t n = pl.size();
    if (!pl[0].next()) return false;
    uint32_t b;
    do {
        uint32_t base = pl[0].get();
        uint32_t pos = base;
        unsigned i = 0;
        do {
            if (++i == n) return true;
            if (!pl[i].skip_to(pos + 1)) return false;
            pos = pl[i].get();
            b = pos + (uint32_t)(n - i);
        } while (b - base <= window);
    } while (pl[0].skip_to(b - window));
    return false;
}

/* NearPostList::test_doc (matcher/nearpostlist.cc:70-150), the whole of it: one occurrence of every term inside a span shorter than
 * `window` AT PAIRWISE DISTINCT POSITIONS, found the reference's way.  The lists are started lazily in ascending wdf order (TermCmp,
 * nearpostlist.cc:52-58; std::sort of <= 16 elements is an insertion sort: ties keep query order); their heads sit in a binary heap
 * with the smallest position on top — common/heap.h, i.e. libc++'s sift-up / sift-down, whose behaviour on equal keys decides which of
 * two coinciding heads is moved; while the span from the top to `last` (the largest head) is too wide the top skips to last - window + 1;
 * once every list is inside the window the heads are walked in ascending position (lines 106-140): a head on the previous head's
 * position is advanced — past the window: back to the outer loop with it as the new maximum (the heap is rebuilt), otherwise it sinks
 * to its place and the walk goes on; all heads distinct: a match.  With one term per position no two heads ever coincide and this is
 * "max(head) - min(head) < window for some alignment" — the predicate the device's wave-parallel path evaluates; a shard whose indexer
 * puts several terms at one position needs every step (xgm_index_set_near_colocated; xgm_posfilter.h near_colocated). */
/* `history`: NearPostList sorts its `terms` MEMBER in place (nearpostlist.cc:80), so between documents the vector keeps the order the
 * previous test left: equal wdf then fall in the order of the LAST TESTED documents' wdf, not in query order — and with coinciding heads
 * the order decides which list is advanced, i.e. whether a match is found (reference quirk, DESIGN.md §7.4: NEAR(ca cb) and NEAR(cb ca)
 * match different documents, and a top-k search, which tests fewer documents, can disagree with the full search of the same query).
 * history != nullptr reproduces that (the vector lives across the calls of one query, in the matcher's document order); nullptr is the
 * stateless reading the device implements: every document starts from query order. */
bool near_window(std::vector<PosCursor>& pl, const std::vector<uint32_t>& wdfs, uint32_t window, std::vector<unsigned>* history = nullptr) {
    const size_t n = pl.size();
    std::vector<unsigned> fresh;
    if (!history) { fresh.resize(n); for (size_t i = 0; i < n; ++i) fresh[i] = (unsigned)i; history = &fresh; }
    std::stable_sort(history->begin(), history->end(), [&](unsigned a, unsigned b) { return wdfs[a] < wdfs[b]; });
    const std::vector<unsigned>& ord = *history;
    std::vector<unsigned> h(n);
    auto pos = [&](unsigned id) { return pl[id].get(); };
    auto above = [&](unsigned a, unsigned b) { return pos(a) > pos(b); };          /* Cmp: a min-heap on the heads */
    auto sift_down = [&](size_t len, size_t start) {
        size_t child = start;
        if (len < 2 || (len - 2) / 2 < child) return;
        child = 2 * child + 1;
        if (child + 1 < len && above(h[child], h[child + 1])) ++child;
        if (above(h[child], h[start])) return;
        const unsigned top = h[start];
        do {
            h[start] = h[child];
            start = child;
            if ((len - 2) / 2 < child) break;
            child = 2 * child + 1;
            if (child + 1 < len && above(h[child], h[child + 1])) ++child;
        } while (!above(h[child], top));
        h[start] = top;
    };
    auto push = [&](size_t len) {
        if (len < 2) return;
        size_t p = (len - 2) / 2, last = len - 1;
        if (!above(h[p], h[last])) return;
        const unsigned t = h[last];
        do {
            h[last] = h[p];
            last = p;
            if (p == 0) break;
            p = (p - 1) / 2;
        } while (above(h[p], t));
        h[last] = t;
    };
    auto pop = [&](size_t len) { if (len > 1) { std::swap(h[0], h[len - 1]); sift_down(len - 1, 0); } };
    h[0] = ord[0];
    if (!pl[h[0]].next()) return false;
    uint32_t last = pos(h[0]);
    size_t end = 1;
    while (true) {
        if (last - pos(h[0]) < window) {
            if (end != n) {
                const unsigned id = ord[end];
                if (last < window) { if (!pl[id].next()) return false; }
                else if (!pl[id].skip_to(last - window + 1)) return false;
                if (pos(id) > last) last = pos(id);
                h[end++] = id;
                push(end);
                continue;
            }
            uint32_t p = pos(h[0]);
            pop(end);
            size_t i = end - 1;
            while (true) {
                if (pos(h[0]) == p) {
                    if (!pl[h[0]].next()) return false;
                    const uint32_t np = pos(h[0]);
                    if (np - pos(h[end - 1]) >= window) { last = np; break; }
                    sift_down(i, 0);
                    continue;
                }
                p = pos(h[0]);
                pop(i);
                if (--i == 0) return true;
            }
            if (end > 1) for (size_t s = (end - 2) / 2 + 1; s-- > 0;) sift_down(end, s);
            continue;
        }
        if (!pl[h[0]].skip_to(last - window + 1)) break;
        last = std::max(last, pos(h[0]));
        sift_down(end, 0);
    }
    return false;
}

struct Result { std::vector<Hit> hits; uint64_t matches = 0; double max_possible = 0, max_attained = 0; uint32_t max_subqs = 0;
                std::vector<std::string> sort_keys;      /* sorted searches: the items' keys, parallel to hits */
                std::vector<std::string> collapse_keys; std::vector<uint32_t> collapse_counts;   /* collapsed searches, parallel to hits */
                uint64_t collapsed_lower_bound = 0;
                uint64_t spy_total = 0; std::map<std::string, uint32_t> spy_values; };  /* documents without a key + distinct keys... kept entries (Collapser::get_matches_lower_bound) */

int run_query(Index* ix, const QueryIn& q, Result* out) {
    const uint32_t n = q.n_terms;
    /* AND_NOT / AND_MAYBE / FILTER (api/queryinternal.cc:2208-2283): l = postlist of subquery 0 (here the AND of
     * the first nr terms), r = the remaining terms — an unweighted OR whose matches are excluded
     * (AndNotPostList), a weighted OR whose weight is added where it matches (AndMaybePostList::get_weight,
     * andmaybepostlist.cc:57-64), or unweighted required terms (QueryFilter: MultiAnd of {l, r x 0}). */
    const bool sided = q.op >= 4 && q.op <= 6;
    const uint32_t nr = sided ? q.n_required : n;
    if (sided && (nr == 0 || nr >= n)) return -1;
    BM25 proto;
    const uint32_t N = q.use_global ? q.g_collection_size : ix->doccount;
    const uint64_t TL = q.use_global ? q.g_total_length : ix->total_length;
    const bool full_pos = q.use_global ? q.g_has_positions != 0 : ix->has_positions;
    const double avg = N == 0 ? 0.0 : (double)TL / N;
    /* enquire.cc:419-426 */
    uint32_t docs = ix->doccount;
    uint32_t first = std::min(q.first, docs);
    uint32_t maxitems = std::min(q.maxitems, docs - first);
    const size_t k = (size_t)first + maxitems;

    std::vector<uint32_t> id(n), tf_local(n);
    std::vector<BM25> wt(n);
    for (uint32_t i = 0; i < n; ++i) {
        auto it = ix->dict.find(std::string(q.terms[i], q.term_len[i]));
        id[i] = it == ix->dict.end() ? UINT32_MAX : it->second;
        tf_local[i] = id[i] == UINT32_MAX ? 0 : ix->df[id[i]];
        wt[i] = proto;
        wt[i].init(N, q.use_global ? q.g_termfreq[i] : tf_local[i], avg);
    }
    bool phrase = (q.op == 3 || q.op == 7) && n > 1 && full_pos;     /* positional filter over the AND: PHRASE or NEAR */
    if (phrase && !ix->has_positions) { out->hits.clear(); return 0; }
    uint32_t window = q.window ? q.window : n;

    /* plan order + tree */
    std::vector<uint32_t> order(n);
    std::vector<std::pair<int, int>> nodes;
    int root = 0;
    /* Huffman-shaped OR tree over plan positions [lo, n) (OrContext::postlist); returns its root */
    auto or_tree = [&](uint32_t lo) -> int {
        const uint32_t m = n - lo;
        if (m == 1) return (int)lo;
        std::vector<HItem> h;
        for (uint32_t p = lo; p < n; ++p) h.push_back(HItem{tf_local[order[p]], (int)p});
        for (long s = (long)((m - 2) / 2); s >= 0; --s) sift_down(h, m, (size_t)s);
        while (true) {
            HItem r = h.front();
            size_t len = h.size();
            if (len > 1) { std::swap(h[0], h[len - 1]); sift_down(h, len - 1, 0); }
            h.pop_back();
            HItem l = h.front();
            nodes.push_back({l.node, r.node});
            int nid = (int)n + (int)nodes.size() - 1;
            if (h.size() == 1) return nid;
            h[0].node = nid; h[0].tf = l.tf + r.tf;
            sift_down(h, h.size(), 0);
        }
    };
    if (sided) {
        /* left side: MultiAnd order among the required terms; right side: query order */
        std::vector<Leaf> in(nr), sorted(nr);
        for (uint32_t i = 0; i < nr; ++i) in[i] = Leaf{tf_local[i], i};
        std::partial_sort_copy(in.begin(), in.end(), sorted.begin(), sorted.end(), TfAsc());
        for (uint32_t i = 0; i < nr; ++i) order[i] = sorted[i].idx;
        for (uint32_t i = nr; i < n; ++i) order[i] = i;
        for (uint32_t p = 1; p < nr; ++p) { nodes.push_back({root, (int)p}); root = (int)n + (int)nodes.size() - 1; }
        if (q.op == 5) {
            int r_root = or_tree(nr);
            nodes.push_back({root, r_root});                    /* l + r, andmaybepostlist.cc:57-64 */
            root = (int)n + (int)nodes.size() - 1;
        }
    } else if (q.op == 2) {
        for (uint32_t i = 0; i < n; ++i) order[i] = i;
        if (n > 1) {
            std::vector<HItem> h;
            for (uint32_t i = 0; i < n; ++i) h.push_back(HItem{tf_local[i], (int)i});
            for (long s = (long)((n - 2) / 2); s >= 0; --s) sift_down(h, n, (size_t)s);
            while (true) {
                HItem r = h.front();
                size_t len = h.size();
                if (len > 1) { std::swap(h[0], h[len - 1]); sift_down(h, len - 1, 0); }
                h.pop_back();
                HItem l = h.front();
                nodes.push_back({l.node, r.node});
                int nid = (int)n + (int)nodes.size() - 1;
                if (h.size() == 1) { root = nid; break; }
                h[0].node = nid; h[0].tf = l.tf + r.tf;
                sift_down(h, h.size(), 0);
            }
        }
    } else {
        std::vector<Leaf> in(n), sorted(n);
        for (uint32_t i = 0; i < n; ++i) in[i] = Leaf{tf_local[i], i};
        std::partial_sort_copy(in.begin(), in.end(), sorted.begin(), sorted.end(), TfAsc());
        for (uint32_t i = 0; i < n; ++i) order[i] = sorted[i].idx;
        for (uint32_t p = 1; p < n; ++p) { nodes.push_back({root, (int)p}); root = (int)n + (int)nodes.size() - 1; }
    }
    {
        std::vector<double> val(2 * n);
        for (uint32_t p = 0; p < n; ++p) {
            uint32_t i = order[p];
            val[p] = wt[i].maxpart(id[i] == UINT32_MAX ? 0 : ix->wdf_ub[id[i]], ix->doclen_lb);
        }
        for (size_t j = 0; j < nodes.size(); ++j) val[n + j] = val[nodes[j].first] + val[nodes[j].second];
        out->max_possible = val[root];
    }

    std::vector<GlassIter> it(n);
    for (uint32_t p = 0; p < n; ++p) {
        uint32_t i = order[p];
        static GlassList empty;
        it[p].init(id[i] == UINT32_MAX ? &empty : &ix->list(id[i]));
    }
    GlassIter dl;
    dl.init(&ix->doclen_list);
    ProtoMSet pm(k);
    double select_cached = -HUGE_VAL;
    std::vector<unsigned> near_history(n);          /* NearPostList::terms as the previous test_doc left it (reference mode only) */
    for (uint32_t i = 0; i < n; ++i) near_history[i] = i;
    std::vector<double> val(2 * n);
    std::vector<char> present(n);

    uint32_t last_subqs = 0;
    auto weigh = [&](uint32_t did) -> double {
        dl.skip_to(did);                                      /* get_doclength → jump_to, :994-1021 */
        uint32_t len = dl.wdf;
        uint32_t subqs = 0;
        for (uint32_t p = 0; p < n; ++p) {
            if (present[p]) { val[p] = wt[order[p]].sumpart(it[p].wdf, len); ++subqs; }
        }
        /* tree sum; an absent side contributes no addition (orpostlist.cc:94-103).  For AND the chain
         * starts at 0.0 in the reference (multiandpostlist.cc:150-160): 0.0 + w0 == w0 exactly. */
        std::vector<char> pr(2 * n, 0);
        for (uint32_t p = 0; p < n; ++p) pr[p] = present[p];
        for (size_t j = 0; j < nodes.size(); ++j) {
            int a = nodes[j].first, b = nodes[j].second;
            size_t o = n + j;
            if (pr[a] && pr[b]) { val[o] = val[a] + val[b]; pr[o] = 1; }
            else if (pr[a]) { val[o] = val[a]; pr[o] = 1; }
            else if (pr[b]) { val[o] = val[b]; pr[o] = 1; }
        }
        last_subqs = subqs;
        return val[root];
    };
    uint64_t true_matches = 0;      /* every matching document, pruned or not (the reference only estimates this) */
    std::vector<Hit> every;         /* sorted searches: min_weight stays 0 when the value leads (and weight pruning is exact when the
                                       weight leads), so the MSet is the best first + maxitems of ALL matches under the chosen order */
    const std::vector<std::string>* spy_val = nullptr;
    if (q.spy_slot_plus1) { if (q.spy_slot_plus1 - 1 >= ix->values.size()) return -1; spy_val = &ix->values[q.spy_slot_plus1 - 1]; }
    auto score = [&](uint32_t did) {
        ++true_matches;
        double w = weigh(did);
        if (spy_val) { ++out->spy_total; if (did < spy_val->size() && !(*spy_val)[did].empty()) ++out->spy_values[(*spy_val)[did]]; }
        if (q.sort_by || q.collapse_max) { every.push_back(Hit{did, last_subqs, w}); return; }
        if (w < pm.min_weight) { return; }                    /* matcher.cc:496-498 */
        pm.add(Hit{did, last_subqs, w});
    };

    if (q.op == 2) {
        for (uint32_t p = 0; p < n; ++p) it[p].next();
        while (true) {
            uint32_t did = UINT32_MAX;
            for (uint32_t p = 0; p < n; ++p) if (!it[p].at_end()) did = std::min(did, it[p].did);
            if (did == UINT32_MAX) break;
            for (uint32_t p = 0; p < n; ++p) present[p] = (!it[p].at_end() && it[p].did == did);
            score(did);
            for (uint32_t p = 0; p < n; ++p) if (present[p]) it[p].next();
        }
    } else if (sided) {
        /* required: plan positions [0, nr) and, for FILTER, all the others as well */
        std::vector<uint32_t> reqp;
        for (uint32_t p = 0; p < n; ++p) if (p < nr || q.op == 6) reqp.push_back(p);
        for (uint32_t p = nr; p < n; ++p) if (q.op != 6) it[p].next();
        it[reqp[0]].next();
        while (!it[reqp[0]].at_end()) {
            uint32_t did = it[reqp[0]].did;
            bool matched = true;
            for (size_t x = 1; x < reqp.size(); ++x) {
                GlassIter& o = it[reqp[x]];
                o.skip_to(did);
                if (o.at_end()) { matched = false; did = UINT32_MAX; break; }
                if (o.did != did) { it[reqp[0]].skip_to(o.did); matched = false; break; }
            }
            if (did == UINT32_MAX) break;
            if (!matched) continue;
            std::fill(present.begin(), present.end(), 0);
            for (uint32_t p = 0; p < nr; ++p) present[p] = 1;
            bool excluded = false;
            for (uint32_t p = nr; p < n && q.op != 6; ++p) {
                if (!it[p].at_end() && it[p].did < did) it[p].skip_to(did);
                const bool here = !it[p].at_end() && it[p].did == did;
                if (q.op == 4) excluded = excluded || here; else present[p] = here;
            }
            if (!excluded) score(did);
            it[reqp[0]].next();
        }
    } else {
        /* MultiAndPostList::find_next_match, multiandpostlist.cc:180-207 */
        std::fill(present.begin(), present.end(), 1);
        it[0].next();
        while (!it[0].at_end()) {
            uint32_t did = it[0].did;
            bool matched = true;
            for (uint32_t p = 1; p < n; ++p) {
                it[p].skip_to(did);
                if (it[p].at_end()) { matched = false; did = UINT32_MAX; break; }
                if (it[p].did != did) { it[0].skip_to(it[p].did); matched = false; break; }
            }
            if (did == UINT32_MAX) break;
            if (!matched) continue;
            bool ok = true;
            double cached_weight = -HUGE_VAL;
            if (phrase) {
                if (q.select_cache_bug) {
                    /* SelectPostList::vet, selectpostlist.cc:28-46, with its stale cache */
                    double w_min = pm.min_weight;
                    if (w_min <= 0.0) {
                        select_cached = -HUGE_VAL;
                    } else {
                        if (!(select_cached >= 0)) select_cached = weigh(did);
                        if (select_cached < w_min) { it[0].next(); continue; }
                    }
                    cached_weight = select_cached;
                }
                /* position lists in PHRASE order (terms[] of the PosFilter = query order) */
                std::vector<PosCursor> pl(n);
                std::vector<uint32_t> wdfs(n);
                for (uint32_t p = 0; p < n; ++p) {
                    uint32_t i = order[p];
                    /* ordinal of the posting: recover from the raw arrays by binary search */
                    const uint32_t* b = ix->did + ix->term_start[id[i]];
                    const uint32_t* e = ix->did + ix->term_start[id[i] + 1];
                    uint64_t ord = (uint64_t)(std::lower_bound(b, e, did) - ix->did);
                    pl[i] = PosCursor{ix->pos + ix->pos_off[ord], (uint32_t)(ix->pos_off[ord + 1] - ix->pos_off[ord]), 0, false};
                    wdfs[i] = it[p].wdf;
                }
                ok = q.op == 7 ? near_window(pl, wdfs, window, q.select_cache_bug ? &near_history : nullptr) : (window == n) ? exact_phrase(pl, wdfs) : window_phrase(pl, window);
            }
            if (ok) {
                if (cached_weight >= 0) {
                    /* SelectPostList::get_weight returns the cached value (selectpostlist.cc:48-55) */
                    ++true_matches;
                    weigh(did);
                    if (!(cached_weight < pm.min_weight)) pm.add(Hit{did, last_subqs, cached_weight});
                } else {
                    score(did);
                }
            }
            it[0].next();
        }
    }
    if (q.sort_by || q.collapse_max) {
        if (q.sort_by && q.sort_slot >= ix->values.size()) return -1;
        if (q.collapse_max && q.collapse_slot >= ix->values.size()) return -1;
        static const std::vector<std::string> no_values;
        static const std::string none;
        const std::vector<std::string>& val = q.sort_by ? ix->values[q.sort_slot] : no_values;
        auto key = [&](uint32_t did) -> const std::string& { return did < val.size() ? val[did] : none; };
        /* msetcmp_by_value / _by_value_then_relevance / _by_relevance_then_value with FORWARD_DID (docid order ascending):
         * "a ranks before b"; FORWARD_VALUE = the reverse flag */
        const bool fwd = q.sort_reverse != 0;
        auto before = [&](const Hit& a, const Hit& b) {
            if (q.sort_by == 0) return mcmp(a, b);
            if (q.sort_by == 3) { if (a.weight > b.weight) return true; if (a.weight < b.weight) return false; }
            const int c = key(a.did).compare(key(b.did));
            if (c > 0) return fwd;
            if (c < 0) return !fwd;
            if (q.sort_by == 2) { if (a.weight > b.weight) return true; if (a.weight < b.weight) return false; }
            return a.did < b.did;
        };
        /* ProtoMSet::update_max_weight sees EVERY matching document (process() and early_reject(), protomset.h:174-183, 249-283):
         * max_attained and the percentages refer to the best weight of the whole match, not of the page */
        double mw = 0.0; uint32_t msub = 0;
        for (const Hit& h : every) if (h.weight > mw) { mw = h.weight; msub = h.subqs; }
        if (q.collapse_max) {
            /* the whole match in rank order; per key the first collapse_max survive, the others are counted against the key */
            const std::vector<std::string>& cval = ix->values[q.collapse_slot];
            auto ckey = [&](uint32_t did) -> const std::string& { return did < cval.size() ? cval[did] : none; };
            std::sort(every.begin(), every.end(), before);
            std::map<std::string, uint32_t> seen;
            std::vector<Hit> kept;
            uint64_t no_key = 0, entries = 0;
            for (const Hit& h : every) {
                const std::string& ck = ckey(h.did);
                if (ck.empty()) { ++no_key; kept.push_back(h); continue; }
                uint32_t& n = seen[ck];
                if (n++ < q.collapse_max) { kept.push_back(h); ++entries; }
            }
            out->collapsed_lower_bound = no_key + entries;
            if (kept.size() > k) kept.resize(k);
            for (const Hit& h : kept) {
                const std::string& ck = ckey(h.did);
                out->collapse_keys.push_back(ck);
                out->collapse_counts.push_back(ck.empty() ? 0u : (seen[ck] > q.collapse_max ? seen[ck] - q.collapse_max : 0u));
            }
            every.swap(kept);
        } else {
            const size_t keep = std::min<size_t>(k, every.size());
            std::partial_sort(every.begin(), every.begin() + keep, every.end(), before);
            every.resize(keep);
        }
        out->hits = every;
        for (const Hit& h : every) out->sort_keys.push_back(key(h.did));
        out->matches = true_matches;
        out->max_attained = mw;
        out->max_subqs = msub;
        return 0;
    }
    pm.finalise();
    out->hits = pm.results;
    out->matches = true_matches;
    out->max_attained = pm.results.empty() ? 0.0 : pm.results[0].weight;
    out->max_subqs = pm.results.empty() ? 0 : pm.results[0].subqs;
    return 0;
}

/* ----------------------------------------------------------------------------- nested queries ---------- */
/* General operator trees, restated from the reference's Query → PostList lowering (api/queryinternal.cc):
 *   QueryTerm::postlist :1049-1056, QueryScaleWeight::postlist :1076-1080 (factor multiplies down),
 *   QueryAndLike::postlist / postlist_sub_and_like :2083-2103 (nested ANDs and FILTERs flatten into one AndContext →
 *     ONE MultiAndPostList whose children are ordered by get_termfreq_est, multiandpostlist.h:117-130),
 *   QueryOr / do_or_like :1790-1820, 2193-2206 (nested ORs flatten into one OrContext → Huffman tree, :440-489),
 *   QueryAndNot::postlist :2208-2225 (right side: unweighted OR of the other subqueries),
 *   QueryAndMaybe::postlist :2248-2269 (right side: weighted OR; an unweighted AND_MAYBE is its left branch),
 *   QueryFilter::postlist :2271-2283 (MultiAnd{l, r x 0}),  QuerySynonym / do_synonym :1822-1898, 2386-2397 with
 *   LocalSubMatch::make_synonym_postlist (matcher/localsubmatch.cc:199-229) and SynonymPostList::get_weight
 *   (matcher/synonympostlist.cc:62-95): ONE BM25 weight over Σ wdf of the matching members, its termfreq the
 *   independence estimate of BoolOrPostList::get_termfreq_est_using_stats (matcher/boolorpostlist.cc:192-230).
 * Estimates that order the children: leafpostlist.cc:51-54, orpostlist.cc:365-384, multiandpostlist.cc:92-105,
 * andnotpostlist.cc:50-61, AndMaybe = its left side, Synonym = BoolOrPostList::get_termfreq_est :175-190.
 * Weights: multiandpostlist.cc:150-160, orpostlist.cc:94-103, andmaybepostlist.cc:57-64, leafpostlist.cc:57-65.
 * count_matching_subqs: leafpostlist.cc:88-91 (weighted leaves), synonym = 1, sums / left side as in each PostList.
 * Evaluation is exhaustive, document at a time over dense per-term wdf arrays: test sizes only. */
enum { T_TERM = 0, T_AND = 1, T_OR = 2, T_AND_NOT = 3, T_AND_MAYBE = 4, T_FILTER = 5, T_SYNONYM = 6, T_SCALE = 7 };

struct TreeIn {
    uint32_t n_terms; const char* const* terms; const uint32_t* term_len; const uint32_t* wqf;
    uint32_t n_ops; const uint8_t* kind; const uint8_t* arity; const uint16_t* term; const double* scale;
    uint32_t first, maxitems;
    uint32_t use_global; uint64_t g_total_length; uint32_t g_collection_size; const uint32_t* g_termfreq;
};

struct ANode { int kind; std::vector<int> kids; int term = -1; double scale = 1.0; };

enum { P_LEAF = 0, P_SYN, P_MAND, P_OR, P_ANDNOT, P_MAYBE };
struct PNode {
    int type; std::vector<int> kids; int term = -1; std::vector<int> members; double factor = 1.0;
    uint32_t est = 0; BM25 wt; bool weighted = false; double maxw = 0.0;
};

struct TreeEval {
    Index* ix; const TreeIn& q;
    std::vector<ANode> ast; std::vector<PNode> pl;
    std::vector<uint32_t> id, tf_local, tf_global;
    uint32_t N, db_size; double avg; uint32_t doclen_ub = 0; uint32_t total_subqs = 0;
    std::vector<std::vector<uint32_t>> wdf_of;          /* per term: wdf + 1 by docid, 0 = absent */
    TreeEval(Index* ix_, const TreeIn& q_) : ix(ix_), q(q_) {}

    int add(PNode&& n) { pl.push_back(std::move(n)); return (int)pl.size() - 1; }
    int leaf(int t, double factor) {
        PNode n; n.type = P_LEAF; n.term = t; n.factor = factor; n.weighted = factor != 0.0;
        n.est = tf_local[t];
        n.wt.init(N, tf_global[t], avg, q.wqf ? std::max(1u, q.wqf[t]) : 1u, factor);
        if (n.weighted) { ++total_subqs; n.maxw = n.wt.maxpart(id[t] == UINT32_MAX ? 0 : ix->wdf_ub[id[t]], ix->doclen_lb); }
        return add(std::move(n));
    }
    int synonym(const std::vector<int>& terms, double factor) {
        PNode n; n.type = P_SYN; n.members = terms; n.factor = factor; n.weighted = factor != 0.0;
        {   /* BoolOrPostList::get_termfreq_est on the shard, :175-190 */
            const double scale = db_size ? 1.0 / db_size : 0.0;
            double P = tf_local[terms[0]] * scale;
            for (size_t i = 1; i < terms.size(); ++i) { double Pi = tf_local[terms[i]] * scale; P += Pi - P * Pi; }
            n.est = db_size ? (uint32_t)(P * db_size + 0.5) : 0u;
        }
        uint32_t tf_syn = 0;
        if (N) {  /* get_termfreq_est_using_stats with the merged statistics, :192-230 */
            const double scale = 1.0 / N;
            double P = tf_global[terms[0]] * scale;
            for (size_t i = 1; i < terms.size(); ++i) { double Pi = tf_global[terms[i]] * scale; P += Pi - P * Pi; }
            tf_syn = (uint32_t)(P * N + 0.5);
        }
        n.wt.init(N, tf_syn, avg, 1u, factor);
        if (n.weighted) { ++total_subqs; n.maxw = n.wt.maxpart(doclen_ub, ix->doclen_lb); }     /* Weight::init_ synonym case: wdf bound = doclength bound */
        return add(std::move(n));
    }
    int mand(std::vector<int> ctx) {
        PNode n; n.type = P_MAND;
        std::vector<Leaf> in(ctx.size()), sorted(ctx.size());
        for (size_t i = 0; i < ctx.size(); ++i) in[i] = Leaf{pl[ctx[i]].est, (uint32_t)i};
        std::partial_sort_copy(in.begin(), in.end(), sorted.begin(), sorted.end(), TfAsc());
        for (auto& l : sorted) n.kids.push_back(ctx[l.idx]);
        double r = pl[n.kids[0]].est;
        for (size_t i = 1; i < n.kids.size(); ++i) r = (r * pl[n.kids[i]].est) / db_size;
        n.est = db_size ? (uint32_t)(r + 0.5) : 0u;
        double m = 0.0;
        for (int k : n.kids) m += pl[k].maxw;                     /* MultiAndPostList::recalc_maxweight, :169-179 */
        n.maxw = m;
        return add(std::move(n));
    }
    int or2(int l, int r) {
        PNode n; n.type = P_OR; n.kids = {l, r};
        const double a = pl[l].est, b = pl[r].est, nn = db_size;
        n.est = nn == 0.0 ? 0u : (uint32_t)(a + b - (a * b / nn) + 0.5);
        n.maxw = pl[l].maxw + pl[r].maxw;
        return add(std::move(n));
    }
    int or_tree(std::vector<int> ctx) {
        if (ctx.empty()) return -1;
        if (ctx.size() == 1) return ctx[0];
        std::vector<HItem> h;
        for (int c : ctx) h.push_back(HItem{pl[c].est, c});
        const size_t m = h.size();
        for (long s2 = (long)((m - 2) / 2); s2 >= 0; --s2) sift_down(h, m, (size_t)s2);
        while (true) {
            HItem r = h.front();
            size_t len = h.size();
            if (len > 1) { std::swap(h[0], h[len - 1]); sift_down(h, len - 1, 0); }
            h.pop_back();
            HItem l = h.front();
            int nid = or2(l.node, r.node);
            if (h.size() == 1) return nid;
            h[0].node = nid; h[0].tf = l.tf + r.tf;
            sift_down(h, h.size(), 0);
        }
    }
    void sub_and_like(int a, std::vector<int>& ctx, double factor) {
        const ANode& n = ast[a];
        if (n.kind == T_AND) { for (int k : n.kids) sub_and_like(k, ctx, factor); return; }
        if (n.kind == T_FILTER) { for (int k : n.kids) { sub_and_like(k, ctx, factor); factor = 0.0; } return; }
        ctx.push_back(postlist(a, factor));
    }
    void sub_or_like(int a, std::vector<int>& ctx, double factor) {
        const ANode& n = ast[a];
        if (n.kind == T_OR) { for (int k : n.kids) sub_or_like(k, ctx, factor); return; }
        ctx.push_back(postlist(a, factor));
    }
    int postlist(int a, double factor) {
        const ANode& n = ast[a];
        switch (n.kind) {
        case T_TERM: return leaf(n.term, factor);
        case T_SCALE: return postlist(n.kids[0], factor * n.scale);
        case T_AND: { std::vector<int> ctx; sub_and_like(a, ctx, factor); return mand(ctx); }
        case T_FILTER: { int l = postlist(n.kids[0], factor); int r = postlist(n.kids[1], 0.0); return mand({l, r}); }
        case T_OR: { std::vector<int> ctx; sub_or_like(a, ctx, factor); return or_tree(ctx); }
        case T_AND_NOT: {
            int l = postlist(n.kids[0], factor);
            std::vector<int> ctx;
            for (size_t i = 1; i < n.kids.size(); ++i) sub_or_like(n.kids[i], ctx, 0.0);
            int r = or_tree(ctx);
            if (r < 0) return l;
            PNode p; p.type = P_ANDNOT; p.kids = {l, r};
            double e = pl[l].est;
            e = (e * (db_size - (double)pl[r].est)) / db_size;
            p.est = db_size ? (uint32_t)(e + 0.5) : 0u;
            p.maxw = pl[l].maxw;
            return add(std::move(p));
        }
        case T_AND_MAYBE: {
            int l = postlist(n.kids[0], factor);
            if (factor == 0.0) return l;
            std::vector<int> ctx;
            for (size_t i = 1; i < n.kids.size(); ++i) sub_or_like(n.kids[i], ctx, factor);
            int r = or_tree(ctx);
            if (r < 0) return l;
            PNode p; p.type = P_MAYBE; p.kids = {l, r}; p.est = pl[l].est; p.maxw = pl[l].maxw + pl[r].maxw;
            return add(std::move(p));
        }
        case T_SYNONYM: {
            std::vector<int> terms;
            for (int k : n.kids) terms.push_back(ast[k].term);
            if (terms.size() == 1) return leaf(terms[0], factor);          /* QuerySynonym::done: a synonym of one term is the term */
            return synonym(terms, factor);
        }
        }
        return -1;
    }
    struct Val { bool p; double w; uint32_t c; };
    Val eval(int x, uint32_t did, uint32_t len) const {
        const PNode& n = pl[x];
        switch (n.type) {
        case P_LEAF: {
            uint32_t e = wdf_of[n.term].empty() ? 0u : wdf_of[n.term][did];
            if (!e) return Val{false, 0.0, 0};
            return Val{true, n.weighted ? n.wt.sumpart(e - 1, len) : 0.0, n.weighted ? 1u : 0u};
        }
        case P_SYN: {
            bool any = false; uint32_t wdf = 0;
            for (int t : n.members) { uint32_t e = wdf_of[t].empty() ? 0u : wdf_of[t][did]; if (e) { any = true; wdf += e - 1; } }
            if (!any) return Val{false, 0.0, 0};
            return Val{true, n.weighted ? n.wt.sumpart(wdf, len) : 0.0, n.weighted ? 1u : 0u};
        }
        case P_MAND: {
            double w = 0.0; uint32_t c = 0;
            for (int k : n.kids) { Val v = eval(k, did, len); if (!v.p) return Val{false, 0.0, 0}; w += v.w; c += v.c; }
            return Val{true, w, c};
        }
        case P_OR: {
            Val l = eval(n.kids[0], did, len), r = eval(n.kids[1], did, len);
            if (l.p && r.p) return Val{true, l.w + r.w, l.c + r.c};
            if (l.p) return l;
            if (r.p) return r;
            return Val{false, 0.0, 0};
        }
        case P_ANDNOT: {
            Val l = eval(n.kids[0], did, len);
            if (!l.p) return l;
            Val r = eval(n.kids[1], did, len);
            return r.p ? Val{false, 0.0, 0} : l;
        }
        case P_MAYBE: {
            Val l = eval(n.kids[0], did, len);
            if (!l.p) return l;
            Val r = eval(n.kids[1], did, len);
            return r.p ? Val{true, l.w + r.w, l.c + r.c} : l;
        }
        }
        return Val{false, 0.0, 0};
    }
};

int run_tree(Index* ix, const TreeIn& q, Result* out, uint32_t* total_subqs_out) {
    TreeEval ev(ix, q);
    /* post-order program → AST */
    std::vector<int> stack;
    for (uint32_t i = 0; i < q.n_ops; ++i) {
        ANode n; n.kind = q.kind[i];
        if (n.kind == T_TERM) { if (q.term[i] >= q.n_terms) return -1; n.term = q.term[i]; }
        else {
            const uint32_t ar = n.kind == T_SCALE ? 1u : q.arity[i];
            if (ar == 0 || stack.size() < ar) return -1;
            n.kids.assign(stack.end() - ar, stack.end());
            stack.resize(stack.size() - ar);
            if (n.kind == T_SCALE) n.scale = q.scale[i];
            if (n.kind == T_SYNONYM) for (int k : n.kids) if (ev.ast[k].kind != T_TERM) return -1;
            if ((n.kind == T_FILTER) && ar != 2) return -1;
            if ((n.kind == T_AND_NOT || n.kind == T_AND_MAYBE) && ar < 2) return -1;
        }
        ev.ast.push_back(n);
        stack.push_back((int)ev.ast.size() - 1);
        /* QueryAndLike / OrLike::done: one subquery → that subquery */
        if ((n.kind == T_AND || n.kind == T_OR) && n.kids.size() == 1) { stack.back() = n.kids[0]; }
    }
    if (stack.size() != 1) return -1;
    const uint32_t n = q.n_terms;
    ev.N = q.use_global ? q.g_collection_size : ix->doccount;
    const uint64_t TL = q.use_global ? q.g_total_length : ix->total_length;
    ev.avg = ev.N == 0 ? 0.0 : (double)TL / ev.N;
    ev.db_size = ix->doccount;
    ev.id.resize(n); ev.tf_local.resize(n); ev.tf_global.resize(n); ev.wdf_of.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        auto it = ix->dict.find(std::string(q.terms[i], q.term_len[i]));
        ev.id[i] = it == ix->dict.end() ? UINT32_MAX : it->second;
        ev.tf_local[i] = ev.id[i] == UINT32_MAX ? 0 : ix->df[ev.id[i]];
        ev.tf_global[i] = q.use_global ? q.g_termfreq[i] : ev.tf_local[i];
        if (ev.id[i] != UINT32_MAX) {
            ev.wdf_of[i].assign((size_t)ix->lastdocid + 1, 0u);
            for (uint64_t p = ix->term_start[ev.id[i]]; p < ix->term_start[ev.id[i] + 1]; ++p) ev.wdf_of[i][ix->did[p]] = ix->wdf[p] + 1u;
        }
    }
    for (uint32_t d = 1; d <= ix->lastdocid; ++d) ev.doclen_ub = std::max(ev.doclen_ub, ix->doclen_dense[d]);
    const int root = ev.postlist(stack[0], 1.0);
    if (root < 0) return -1;
    uint32_t docs = ix->doccount;
    uint32_t first = std::min(q.first, docs);
    uint32_t maxitems = std::min(q.maxitems, docs - first);
    ProtoMSet pm((size_t)first + maxitems);
    uint64_t matches = 0;
    for (uint32_t d = 1; d <= ix->lastdocid; ++d) {
        const uint32_t len = ix->doclen_dense[d];
        if (!len && !ix->doclen_dense.empty() && ix->doclen_dense[d] == 0) { /* absent docid (or empty document): no leaf can index it */ }
        TreeEval::Val v = ev.eval(root, d, len);
        if (!v.p) continue;
        ++matches;
        if (v.w < pm.min_weight) continue;
        pm.add(Hit{d, v.c, v.w});
    }
    pm.finalise();
    out->hits = pm.results;
    out->matches = matches;
    out->max_possible = ev.pl[root].maxw;
    out->max_attained = pm.results.empty() ? 0.0 : pm.results[0].weight;
    out->max_subqs = pm.results.empty() ? 0 : pm.results[0].subqs;
    if (total_subqs_out) *total_subqs_out = ev.total_subqs;
    return 0;
}

}  // namespace

/* ----------------------------------------------------------------------------- C ABI ----------- */

extern "C" {

void* xgo_corpus_build(uint64_t seed, uint64_t n_docs_global, uint32_t vocab, uint32_t len_lo, uint32_t len_hi,
                       uint32_t n_shards, uint32_t shard, int with_positions) {
    return corpus_build(seed, n_docs_global, vocab, len_lo, len_hi, n_shards, shard, with_positions != 0);
}
void* xgo_corpus_build_terms(uint64_t seed, uint64_t n_docs_global, uint32_t vocab, uint32_t len_lo, uint32_t len_hi, uint32_t n_shards, uint32_t shard,
                             int with_positions, const uint32_t* ranks, uint32_t n_ranks, uint32_t n_threads) {
    return corpus_build_terms(seed, n_docs_global, vocab, len_lo, len_hi, n_shards, shard, with_positions != 0, ranks, n_ranks, n_threads);
}
void xgo_corpus_free(void* c) { delete (Corpus*)c; }

struct xgo_corpus_view {
    uint32_t n_terms, lastdocid, doccount, has_positions;
    uint64_t total_length, n_postings, n_positions;
    const uint32_t* doclen; const char* const* terms; const uint32_t* term_len; const uint32_t* df;
    const uint32_t* did; const uint32_t* wdf; const uint64_t* pos_off; const uint32_t* pos;
};
void xgo_corpus_get(void* cv, xgo_corpus_view* v) {
    Corpus* c = (Corpus*)cv;
    v->n_terms = (uint32_t)c->terms.size(); v->lastdocid = c->lastdocid; v->doccount = c->doccount;
    v->has_positions = c->has_positions; v->total_length = c->total_length;
    v->n_postings = c->did.size(); v->n_positions = c->pos.size();
    v->doclen = c->doclen.data(); v->terms = c->term_ptr.data(); v->term_len = c->term_len.data(); v->df = c->df.data();
    v->did = c->did.data(); v->wdf = c->wdf.data();
    v->pos_off = c->has_positions ? c->pos_off.data() : nullptr; v->pos = c->has_positions ? c->pos.data() : nullptr;
}

/* The index borrows the raw arrays: keep them alive. */
void* xgo_index_from_raw(uint32_t n_terms, uint32_t lastdocid, uint32_t doccount, uint64_t total_length, const uint32_t* doclen,
                         const char* const* terms, const uint32_t* term_len, const uint32_t* df, const uint32_t* did,
                         const uint32_t* wdf, const uint64_t* pos_off, const uint32_t* pos) {
    return index_from_raw(n_terms, lastdocid, doccount, total_length, doclen, terms, term_len, df, did, wdf, pos_off, pos);
}
void xgo_index_free(void* ix) { delete (Index*)ix; }

int xgo_index_termfreq(void* ixv, const char* term, uint32_t len) {
    Index* ix = (Index*)ixv;
    auto it = ix->dict.find(std::string(term, len));
    return it == ix->dict.end() ? 0 : (int)ix->df[it->second];
}

/* Pre-encode a term's glass list (so timing runs do not include the encoding). */
void xgo_index_warm(void* ixv, const char* term, uint32_t len) {
    Index* ix = (Index*)ixv;
    auto it = ix->dict.find(std::string(term, len));
    if (it != ix->dict.end()) ix->list(it->second);
}

struct xgo_hit { uint32_t docid, subqs; double weight; };
struct xgo_result_hdr { uint32_t n_hits, max_subqs; uint64_t matches; double max_attained, max_possible; };

int xgo_search(void* ixv, uint32_t op, uint32_t n_terms, const char* const* terms, const uint32_t* term_len, uint32_t window,
               uint32_t first, uint32_t maxitems, uint32_t use_global, uint64_t g_total_length, uint32_t g_collection_size,
               uint32_t g_has_positions, const uint32_t* g_termfreq, uint32_t select_cache_bug, xgo_hit* hits,
               xgo_result_hdr* hdr) {
    /* op: low byte = operator, bits 8.. = n_required of AND_NOT / AND_MAYBE / FILTER */
    QueryIn q{op & 0xFFu, op >> 8, n_terms, terms, term_len, window, first, maxitems, use_global, g_total_length, g_collection_size,
              g_has_positions, g_termfreq, select_cache_bug};
    Result r;
    int rc = run_query((Index*)ixv, q, &r);
    if (rc) return rc;
    hdr->n_hits = (uint32_t)r.hits.size(); hdr->max_subqs = r.max_subqs; hdr->matches = r.matches;
    hdr->max_attained = r.max_attained; hdr->max_possible = r.max_possible;
    for (size_t i = 0; i < r.hits.size(); ++i) { hits[i].docid = r.hits[i].did; hits[i].subqs = r.hits[i].subqs; hits[i].weight = r.hits[i].weight; }
    return 0;
}

/* Value slots 0..2 of the synthetic corpus (tools/xgm_corpus.h) for an index over shard `shard` of `n_shards` (local docid d is
 * global document (d - 1) * n_shards + shard + 1). */
void xgo_index_set_synthetic_values(void* ixv, uint64_t seed, uint32_t n_shards, uint32_t shard) {
    Index* ix = (Index*)ixv;
    xgm_corpus_params cp{seed, 1u, 1u, 1u};
    ix->values.assign(3, std::vector<std::string>(ix->lastdocid + 1));
    char vb[16];
    for (uint32_t d = 1; d <= ix->lastdocid; ++d) {
        const uint64_t g = (uint64_t)(d - 1) * n_shards + shard + 1;
        for (uint32_t slot = 0; slot < 3; ++slot) { const uint32_t n = xgm_doc_value(&cp, g, slot, vb); ix->values[slot][d].assign(vb, n); }
    }
}

/* One value slot of the index as the column file of include/xgm.h (xgm_glass_export_column writes the same from glass): the
 * documents' ordinals among the slot's distinct values, then the values.  For device tests that have no glass database. */
int xgo_write_value_column(void* ixv, uint32_t slot, const char* path) {
    Index* ix = (Index*)ixv;
    if (slot >= ix->values.size()) return -1;
    const std::vector<std::string>& value = ix->values[slot];
    std::vector<std::string> distinct;
    for (const std::string& v : value) if (!v.empty()) distinct.push_back(v);
    std::sort(distinct.begin(), distinct.end());
    distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
    FILE* f = fopen(path, "wb");
    if (!f) return -2;
    const uint32_t h32[4] = {slot, ix->lastdocid, (uint32_t)distinct.size(), 0u};
    fwrite("XGMCOL1", 1, 8, f); fwrite(h32, 4, 4, f);
    for (uint32_t d = 0; d <= ix->lastdocid; ++d) {
        uint32_t o = 0;
        if (d < value.size() && !value[d].empty()) o = (uint32_t)(std::lower_bound(distinct.begin(), distinct.end(), value[d]) - distinct.begin()) + 1u;
        fwrite(&o, 4, 1, f);
    }
    uint64_t off = 0;
    for (const std::string& v : distinct) { fwrite(&off, 8, 1, f); off += v.size(); }
    fwrite(&off, 8, 1, f);
    for (const std::string& v : distinct) fwrite(v.data(), 1, v.size(), f);
    fclose(f);
    return 0;
}

/* xgo_search with Enquire::set_sort_by_value* in force (sort_by 1 VAL, 2 VAL_REL, 3 REL_VAL).  keys receives the items' sort
 * keys, key_stride bytes each (NUL-padded; keys of the synthetic corpus are at most 7 bytes). */
int xgo_search_sorted_g(void* ixv, uint32_t op, uint32_t n_terms, const char* const* terms, const uint32_t* term_len, uint32_t window,
                        uint32_t first, uint32_t maxitems, uint32_t sort_by, uint32_t sort_slot, uint32_t sort_reverse,
                        xgo_hit* hits, xgo_result_hdr* hdr, char* keys, uint32_t key_stride,
                        uint32_t collapse_slot, uint32_t collapse_max, char* collapse_keys, uint32_t* collapse_counts, uint64_t* collapsed_lower_bound,
                        uint32_t use_global, uint64_t g_total_length, uint32_t g_collection_size, uint32_t g_has_positions, const uint32_t* g_termfreq);

int xgo_search_sorted(void* ixv, uint32_t op, uint32_t n_terms, const char* const* terms, const uint32_t* term_len, uint32_t window,
                      uint32_t first, uint32_t maxitems, uint32_t sort_by, uint32_t sort_slot, uint32_t sort_reverse,
                      xgo_hit* hits, xgo_result_hdr* hdr, char* keys, uint32_t key_stride,
                      uint32_t collapse_slot, uint32_t collapse_max, char* collapse_keys, uint32_t* collapse_counts, uint64_t* collapsed_lower_bound) {
    return xgo_search_sorted_g(ixv, op, n_terms, terms, term_len, window, first, maxitems, sort_by, sort_slot, sort_reverse, hits, hdr, keys, key_stride,
                               collapse_slot, collapse_max, collapse_keys, collapse_counts, collapsed_lower_bound, 0, 0, 0, 0, nullptr);
}

/* ... on one shard of several, weighted with the merged statistics (Xapiand's per-shard protocol, as xgo_search's use_global) */
int xgo_search_sorted_g(void* ixv, uint32_t op, uint32_t n_terms, const char* const* terms, const uint32_t* term_len, uint32_t window,
                        uint32_t first, uint32_t maxitems, uint32_t sort_by, uint32_t sort_slot, uint32_t sort_reverse,
                        xgo_hit* hits, xgo_result_hdr* hdr, char* keys, uint32_t key_stride,
                        uint32_t collapse_slot, uint32_t collapse_max, char* collapse_keys, uint32_t* collapse_counts, uint64_t* collapsed_lower_bound,
                        uint32_t use_global, uint64_t g_total_length, uint32_t g_collection_size, uint32_t g_has_positions, const uint32_t* g_termfreq) {
    QueryIn q{op & 0xFFu, op >> 8, n_terms, terms, term_len, window, first, maxitems, use_global, g_total_length, g_collection_size, g_has_positions, g_termfreq, 0};
    q.sort_by = sort_by; q.sort_slot = sort_slot; q.sort_reverse = sort_reverse;
    q.collapse_slot = collapse_slot; q.collapse_max = collapse_max;
    Result r;
    int rc = run_query((Index*)ixv, q, &r);
    if (rc) return rc;
    hdr->n_hits = (uint32_t)r.hits.size(); hdr->max_subqs = r.max_subqs; hdr->matches = r.matches;
    hdr->max_attained = r.max_attained; hdr->max_possible = r.max_possible;
    for (size_t i = 0; i < r.hits.size(); ++i) {
        hits[i].docid = r.hits[i].did; hits[i].subqs = r.hits[i].subqs; hits[i].weight = r.hits[i].weight;
        if (keys) { memset(keys + i * key_stride, 0, key_stride); memcpy(keys + i * key_stride, r.sort_keys[i].data(), std::min<size_t>(key_stride - 1, r.sort_keys[i].size())); }
        if (collapse_max && collapse_keys) { memset(collapse_keys + i * key_stride, 0, key_stride); memcpy(collapse_keys + i * key_stride, r.collapse_keys[i].data(), std::min<size_t>(key_stride - 1, r.collapse_keys[i].size())); }
        if (collapse_max && collapse_counts) collapse_counts[i] = r.collapse_counts[i];
    }
    if (collapsed_lower_bound) *collapsed_lower_bound = r.collapsed_lower_bound;
    return 0;
}

/* The counts of a ValueCountMatchSpy on `spy_slot` over every document the query matches (QueryIn::spy_slot_plus1): values[] receives the
 * distinct values in byte order, key_stride bytes each, counts[] how many matching documents carry each; *total = documents seen. */
int xgo_search_spy(void* ixv, uint32_t op, uint32_t n_terms, const char* const* terms, const uint32_t* term_len, uint32_t window,
                   uint32_t spy_slot, uint32_t cap, uint32_t key_stride, char* values, uint32_t* counts, uint32_t* n_values, uint64_t* total) {
    QueryIn q{op & 0xFFu, op >> 8, n_terms, terms, term_len, window, 0, 1, 0, 0, 0, 0, nullptr, 0};
    q.spy_slot_plus1 = spy_slot + 1;
    Result r;
    int rc = run_query((Index*)ixv, q, &r);
    if (rc) return rc;
    if (r.spy_values.size() > cap) return -2;
    uint32_t i = 0;
    for (const auto& kv : r.spy_values) {
        memset(values + (size_t)i * key_stride, 0, key_stride);
        memcpy(values + (size_t)i * key_stride, kv.first.data(), std::min<size_t>(key_stride - 1, kv.first.size()));
        counts[i++] = kv.second;
    }
    *n_values = i; *total = r.spy_total;
    return 0;
}

/* Throughput of the port on several host threads at once (bench.py's cpu_baseline.all_cores): thread t answers
 * queries t, t + n_threads, ... of the list, round and round, until `seconds` have passed.  The index is
 * read-only once every term's list is built, which is done here first (index-build work, not timed).
 * terms / term_len are the queries' terms back to back; n_terms[q] says how many belong to query q. */
int xgo_search_many(void* ixv, uint32_t op, uint32_t n_queries, const uint32_t* n_terms, const char* const* terms,
                    const uint32_t* term_len, uint32_t first, uint32_t maxitems, uint32_t n_threads, double seconds,
                    uint64_t* done_out, double* wall_out) {
    Index* ix = (Index*)ixv;
    if (!ix || n_queries == 0 || n_threads == 0) return -1;
    std::vector<uint64_t> start(n_queries + 1, 0);
    for (uint32_t q = 0; q < n_queries; ++q) start[q + 1] = start[q] + n_terms[q];
    for (uint64_t i = 0; i < start[n_queries]; ++i) {
        auto it = ix->dict.find(std::string(terms[i], term_len[i]));
        if (it != ix->dict.end()) ix->list(it->second);
    }
    std::vector<uint64_t> counts(n_threads, 0);
    std::vector<int> rcs(n_threads, 0);
    const auto t0 = std::chrono::steady_clock::now();
    const auto deadline = t0 + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(seconds));
    std::vector<std::thread> threads;
    for (uint32_t t = 0; t < n_threads; ++t) {
        threads.emplace_back([&, t]() {
            for (uint64_t i = t; std::chrono::steady_clock::now() < deadline; i += n_threads) {
                const uint32_t q = (uint32_t)(i % n_queries);
                QueryIn in{op & 0xFFu, op >> 8, n_terms[q], terms + start[q], term_len + start[q], 0, first, maxitems, 0, 0, 0, 0, nullptr, 0};
                Result r;
                if (run_query(ix, in, &r)) { rcs[t] = -1; return; }
                ++counts[t];
            }
        });
    }
    for (auto& th : threads) th.join();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t done = 0;
    for (uint32_t t = 0; t < n_threads; ++t) { if (rcs[t]) return rcs[t]; done += counts[t]; }
    *done_out = done;
    *wall_out = wall;
    return 0;
}

/* Every query of a list answered once, on n_threads host threads (config-scale parity tests: the expected
 * results of a few hundred 10 M-document queries without minutes of single-threaded CPU time on the GPU box).
 * ops[q] as in xgo_search (operator | n_required << 8); hits is [n_queries][cap], cap >= first + maxitems.
 * The terms' glass lists are encoded first, also in parallel (distinct terms dealt to the threads). */
int xgo_search_batch(void* ixv, uint32_t n_queries, const uint32_t* ops, const uint32_t* n_terms, const uint32_t* windows,
                     const char* const* terms, const uint32_t* term_len, uint32_t first, uint32_t maxitems, uint32_t cap,
                     uint32_t n_threads, uint32_t select_cache_bug, xgo_hit* hits, xgo_result_hdr* hdrs) {
    Index* ix = (Index*)ixv;
    if (!ix || n_threads == 0 || cap < first + maxitems) return -1;
    std::vector<uint64_t> start(n_queries + 1, 0);
    for (uint32_t q = 0; q < n_queries; ++q) start[q + 1] = start[q] + n_terms[q];
    std::vector<uint32_t> ids;
    for (uint64_t i = 0; i < start[n_queries]; ++i) {
        auto it = ix->dict.find(std::string(terms[i], term_len[i]));
        if (it != ix->dict.end()) ids.push_back(it->second);
    }
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    {
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < n_threads; ++t)
            th.emplace_back([&, t]() { for (size_t i = t; i < ids.size(); i += n_threads) ix->list(ids[i]); });
        for (auto& x : th) x.join();
    }
    std::vector<int> rcs(n_threads, 0);
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < n_threads; ++t) {
        th.emplace_back([&, t]() {
            for (uint32_t q = t; q < n_queries; q += n_threads) {
                QueryIn in{ops[q] & 0xFFu, ops[q] >> 8, n_terms[q], terms + start[q], term_len + start[q], windows ? windows[q] : 0u, first, maxitems,
                           0, 0, 0, 0, nullptr, select_cache_bug};
                Result r;
                if (run_query(ix, in, &r)) { rcs[t] = -1; return; }
                xgo_result_hdr& h = hdrs[q];
                h.n_hits = (uint32_t)r.hits.size(); h.max_subqs = r.max_subqs; h.matches = r.matches;
                h.max_attained = r.max_attained; h.max_possible = r.max_possible;
                for (size_t i = 0; i < r.hits.size(); ++i) {
                    xgo_hit& o = hits[(size_t)q * cap + i];
                    o.docid = r.hits[i].did; o.subqs = r.hits[i].subqs; o.weight = r.hits[i].weight;
                }
            }
        });
    }
    for (auto& x : th) x.join();
    for (uint32_t t = 0; t < n_threads; ++t) if (rcs[t]) return rcs[t];
    return 0;
}

/* A nested query: terms[] + a post-order program (kind / arity / term index / scale per op), see run_tree. */
int xgo_search_tree(void* ixv, uint32_t n_terms, const char* const* terms, const uint32_t* term_len, const uint32_t* wqf,
                    uint32_t n_ops, const uint8_t* kind, const uint8_t* arity, const uint16_t* term, const double* scale,
                    uint32_t first, uint32_t maxitems, uint32_t use_global, uint64_t g_total_length, uint32_t g_collection_size,
                    const uint32_t* g_termfreq, xgo_hit* hits, xgo_result_hdr* hdr, uint32_t* total_subqs) {
    TreeIn q{n_terms, terms, term_len, wqf, n_ops, kind, arity, term, scale, first, maxitems, use_global, g_total_length, g_collection_size, g_termfreq};
    Result r;
    int rc = run_tree((Index*)ixv, q, &r, total_subqs);
    if (rc) return rc;
    hdr->n_hits = (uint32_t)r.hits.size(); hdr->max_subqs = r.max_subqs; hdr->matches = r.matches;
    hdr->max_attained = r.max_attained; hdr->max_possible = r.max_possible;
    for (size_t i = 0; i < r.hits.size(); ++i) { hits[i].docid = r.hits[i].did; hits[i].subqs = r.hits[i].subqs; hits[i].weight = r.hits[i].weight; }
    return 0;
}

/* Matcher::merge_mset + unshard_docids for per-shard result lists (already sorted). */
int xgo_merge(uint32_t n_shards, const xgo_hit* const* shard_hits, const uint32_t* n_hits, uint32_t first, uint32_t maxitems,
              xgo_hit* out) {
    std::vector<Hit> all;
    for (uint32_t s = 0; s < n_shards; ++s)
        for (uint32_t i = 0; i < n_hits[s]; ++i)
            all.push_back(Hit{(shard_hits[s][i].docid - 1) * n_shards + s + 1, shard_hits[s][i].subqs, shard_hits[s][i].weight});
    std::sort(all.begin(), all.end(), mcmp);
    uint32_t n = 0;
    for (size_t i = first; i < all.size() && n < maxitems; ++i, ++n) { out[n].docid = all[i].did; out[n].subqs = all[i].subqs; out[n].weight = all[i].weight; }
    return (int)n;
}

}  // extern "C"
