/* The reference's answer to a positional query (OP_PHRASE / OP_NEAR) for a whole BATCH: what xgm_search_replay(XGM_REPLAY_FROZEN_WEIGHT)
 * gives one query at a time — ProtoMSet's collation (matcher/protomset.h:340-400) over the matches in docid order, with the weight
 * SelectPostList freezes once min_weight turns positive (matcher/selectpostlist.cc:28-55) — from the lists xgm_andw_list_kernel's units
 * leave (xgm_device.h, xgm_prefix_entry).
 *
 * Why a PREFIX of the match is enough.  With check_at_least within the page (Enquire::get_mset clamps it to >= first + maxitems = K, Xapiand
 * passes 0) the heap is made by the (K + 1)-th match and min_weight is set there: from the next document on every match is served the
 * frozen weight w* = the weight of the conjunction's next document.  A later match replaces the worst kept document only while w* beats
 * it, and every replacement removes a kept document lighter than w*: at most K of them.  So the page is a function of the query's first
 * 2 K + 1 matches in docid order, each with its successor in the conjunction — which is what every unit lists for its own docid range,
 * stopping when it has 2 (K + 1) of them.  One wave per query walks its units' lists in stripe order; the kept documents live one per
 * lane (K <= 64: the bodies that list serve pages of at most 64).  Integer / fp64-compare work on a few hundred bytes per query. */
#include <hip/hip_runtime.h>

#include "xgm_launch.h"
#include "xgm_wave.h"

namespace {

/* a ranks before b under msetcmp_by_relevance<true> (msetcmp.cc:55-62) */
__device__ __forceinline__ bool fz_before(double aw, uint32_t ad, double bw, uint32_t bd) { return aw > bw || (aw == bw && ad < bd); }

__global__ __launch_bounds__(64) void xgm_frozen_finish_kernel(const xgm_dev_query* __restrict__ queries, uint32_t nq, const uint32_t* __restrict__ goff,
                                                               const xgm_cand* __restrict__ cand, const xgm_group_hdr* __restrict__ ghdr, uint32_t k_stride_c,
                                                               const double* __restrict__ max_possible, const uint32_t* __restrict__ row_of,
                                                               xgm_hit* __restrict__ hits, xgm_result_hdr* __restrict__ hdrs, unsigned long long* __restrict__ extra,
                                                               uint32_t k_stride_out) {
    const uint32_t qi = blockIdx.x, lane = threadIdx.x;
    if (qi >= nq) return;
    const xgm_dev_query& q = queries[qi];
    const uint32_t K = rfl32(q.k), subqs = (uint32_t)__popc(rfl32(q.score_mask));
    const uint32_t g0 = goff[qi], U = goff[qi + 1] - g0;
    const uint32_t orow = row_of ? row_of[qi] : qi;

    /* ---- the units' counts and flags ---- */
    unsigned long long total = 0;
    bool incomplete = false, declined = K > 64u || K == 0u;
    for (uint32_t u0 = 0; u0 < U; u0 += 256u) {                   /* (four headers per lane in flight: a query of frequent terms has thousands of units) */
        unsigned long long hm4[4]; uint32_t hp4[4];
#pragma unroll
        for (uint32_t b = 0; b < 4u; ++b) {
            const uint32_t u = u0 + b * 64u + lane;
            hm4[b] = 0; hp4[b] = XGM_PFX_COMPLETE;
            if (u < U) { const xgm_group_hdr& h = ghdr[g0 + u]; hm4[b] = h.matches; hp4[b] = h.pad; }
        }
#pragma unroll
        for (uint32_t b = 0; b < 4u; ++b) {
            total += hm4[b] & ~XGM_MATCHES_LOWER_BOUND;
            incomplete = incomplete || !(hp4[b] & XGM_PFX_COMPLETE);
            declined = declined || (hp4[b] & XGM_PFX_DECLINED) != 0u;
        }
    }
    for (int sh = 32; sh > 0; sh >>= 1) total += (unsigned long long)__shfl_xor((long long)total, sh);
    incomplete = __ballot(incomplete) != 0ull;
    declined = __ballot(declined) != 0ull;
    if (declined) {
        /* no LIST body for this query: the host answers it (xgm_search_replay) */
        if (lane == 0u) {
            xgm_result_hdr r;
            r.n_hits = 0; r.max_weight_subqs_matched = 0; r.matches_exact = 0; r.max_attained = 0.0; r.max_possible = max_possible ? max_possible[qi] : 0.0;
            hdrs[orow] = r;
            extra[orow] = XGM_EXTRA_FALLBACK;
        }
        return;
    }

    /* ---- ProtoMSet's state: uniform scalars; the kept documents one per lane ---- */
    double kw = 0.0; uint32_t kd = 0;                             /* lane i < size: kept document i */
    uint32_t size = 0;
    bool heap_built = false, frozen = false, have_star = false, stop = false, fallback = false;
    double min_w = 0.0, worst_w = 0.0, w_star = 0.0, best_w = 0.0;
    uint32_t worst_d = 0, worst_l = 0;
    bool best_set = false;
    unsigned long long known = 0;
    bool known_is_total = false;

    auto find_worst = [&]() {                                      /* the kept document that ranks last: every lane ends with the same one */
        double w = kw; uint32_t d = kd, l = lane < size ? lane : 0xFFFFFFFFu;
        for (int sh = 32; sh > 0; sh >>= 1) {
            const double ow = __shfl_xor(w, sh); const uint32_t od = (uint32_t)__shfl_xor((int)d, sh), ol = (uint32_t)__shfl_xor((int)l, sh);
            if (ol != 0xFFFFFFFFu && (l == 0xFFFFFFFFu || fz_before(w, d, ow, od))) { w = ow; d = od; l = ol; }
        }
        worst_w = rl_f64(w, 0u); worst_d = rl32(d, 0u); worst_l = rl32(l, 0u);
    };

    /* the units' headers 64 at a time (a query of frequent terms has thousands of units, nearly all of them empty: skipped by the look-back or without a
     * match): only the units that list something — or whose list is cut short — are walked, in stripe order */
    for (uint32_t u256 = 0; u256 < U && !stop; u256 += 256u) {
        uint32_t hn4[4], hp4[4]; unsigned long long hm4[4];
#pragma unroll
        for (uint32_t b = 0; b < 4u; ++b) {                            /* four batches of headers in flight */
            hn4[b] = 0; hp4[b] = XGM_PFX_COMPLETE; hm4[b] = 0;
            if (u256 + b * 64u + lane < U) { const xgm_group_hdr& hh = ghdr[g0 + u256 + b * 64u + lane]; hn4[b] = hh.n_cand; hp4[b] = hh.pad; hm4[b] = hh.matches & ~XGM_MATCHES_LOWER_BOUND; }
        }
#pragma unroll
    for (uint32_t b4 = 0; b4 < 4u; ++b4) {
        const uint32_t u0 = u256 + b4 * 64u;
        if (u0 >= U || stop) break;
        const uint32_t hn_ = hn4[b4], hp_ = hp4[b4]; const unsigned long long hm_ = hm4[b4];
        const bool cut_ = !(hp_ & XGM_PFX_COMPLETE) || hm_ > (unsigned long long)hn_;
        uint64_t todo = __ballot(hn_ > 0u || cut_);
    while (todo && !stop) {
        const uint32_t L_ = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        const uint32_t u = u0 + L_;
        const uint32_t n = rl32(hn_, L_), pad = rl32(hp_, L_);
        const bool truncated = ((__ballot(cut_) >> L_) & 1ull) != 0ull;      /* the unit has (or may have) matches it did not list */
        const xgm_prefix_entry* ent = reinterpret_cast<const xgm_prefix_entry*>(cand + (size_t)(g0 + u) * k_stride_c);
        for (uint32_t e0 = 0; e0 < n && !stop; e0 += 64u) {
            xgm_prefix_entry my; my.wbits = 0; my.next_wbits = 0; my.did = 0; my.has_next = 0;
            if (e0 + lane < n) my = ent[e0 + lane];
            const uint32_t cnt = n - e0 < 64u ? n - e0 : 64u;
            for (uint32_t j = 0; j < cnt && !stop; ++j) {
                const uint32_t did = rl32(my.did, j);
                const double w_true = __longlong_as_double((long long)rl64(my.wbits, j));
                const double w = frozen ? w_star : w_true;
                if (w < min_w) continue;                            /* matcher.cc:500-505: never shown to ProtoMSet */
                ++known;
                if (w > best_w) { best_w = w; best_set = true; }    /* update_max_weight: only on > (protomset.h:174-183) */
                if (size < K) { if (lane == size) { kw = w; kd = did; } ++size; continue; }
                bool moved = false;
                if (!heap_built) {
                    heap_built = true;
                    find_worst();
                    if (known >= K) { min_w = worst_w; moved = true; }       /* check_at_least == K here (the planner lists such queries only) */
                }
                if (fz_before(w, did, worst_w, worst_d)) {
                    if (lane == worst_l) { kw = w; kd = did; }
                    find_worst();
                    if (known >= K) { min_w = worst_w; moved = true; }
                }
                /* min_weight turned positive: SelectPostList::vet weighs the conjunction's NEXT document and serves that weight from now on */
                if (moved && !frozen && min_w > 0.0) {
                    frozen = true;
                    if (rl32(my.has_next, j)) { w_star = __longlong_as_double((long long)rl64(my.next_wbits, j)); have_star = true; }
                    else {
                        /* the match is its unit's last document of the conjunction: the successor opens a later unit — or there is none and the loop ends */
                        for (uint32_t v0 = u + 1u; v0 < U && !have_star && !fallback; v0 += 64u) {
                            const uint32_t v = v0 + lane;
                            const uint32_t vp = v < U ? ghdr[g0 + v].pad : XGM_PFX_COMPLETE;
                            const uint64_t hm = __ballot((vp & XGM_PFX_HAS_FIRST) != 0u);
                            /* (a unit that stopped before it met a document of the conjunction says nothing about its range: the answer is not in the lists) */
                            const uint64_t um = __ballot(!(vp & XGM_PFX_HAS_FIRST) && !(vp & XGM_PFX_COMPLETE));
                            const uint32_t Lh = hm ? (uint32_t)__builtin_ctzll(hm) : 64u, Lu = um ? (uint32_t)__builtin_ctzll(um) : 64u;
                            if (Lu < Lh) fallback = true;
                            else if (hm) { w_star = __longlong_as_double((long long)ghdr[g0 + v0 + Lh].c_pos); have_star = true; }
                        }
                        if (fallback) stop = true;
                        if (!have_star) stop = true;
                    }
                }
                if (frozen && have_star && w_star < min_w) stop = true;          /* vet() rejects every later document untested */
                /* ... or the frozen weight no longer beats the worst kept (a later docid never wins a tie): the rest of the loop only counts */
                if (frozen && have_star && !stop && !fz_before(w_star, 0xFFFFFFFFu, worst_w, worst_d)) { known_is_total = true; stop = true; }
            }
        }
        /* a unit that stopped listing early, its list walked to the end with the collation still open: more of the match is needed than was listed */
        if (!stop && truncated) { fallback = true; stop = true; }
    }
    }
    }

    if (fallback) {
        if (lane == 0u) {
            xgm_result_hdr r;
            r.n_hits = 0; r.max_weight_subqs_matched = 0; r.matches_exact = 0; r.max_attained = 0.0; r.max_possible = max_possible ? max_possible[qi] : 0.0;
            hdrs[orow] = r;
            extra[orow] = XGM_EXTRA_FALLBACK;
        }
        return;
    }
    /* ---- the page in rank order (ProtoMSet::finalise sorts, protomset.h:657): ranks by counting ---- */
    uint32_t r = 0;
    for (uint32_t j = 0; j < size; ++j) {
        const double ow = rl_f64(kw, j); const uint32_t od = rl32(kd, j);
        r += (lane < size && fz_before(ow, od, kw, kd)) ? 1u : 0u;
    }
    if (lane < size) {
        xgm_hit hit; hit.docid = kd; hit.subqs_matched = subqs; hit.weight = kw;
        hits[(size_t)orow * k_stride_out + r] = hit;
    }
    if (lane == 0u) {
        xgm_result_hdr o;
        o.n_hits = size;
        o.max_weight_subqs_matched = best_set ? subqs : 0u;
        o.matches_exact = total | (incomplete ? XGM_MATCHES_LOWER_BOUND : 0ull);
        o.max_attained = best_set ? best_w : 0.0;
        o.max_possible = max_possible ? max_possible[qi] : 0.0;
        hdrs[orow] = o;
        /* known_matching_docs: what the walk counted — or, once the rest of the loop only counts, the match count (a lower bound of it when units stopped early) */
        const unsigned long long kn = known_is_total ? total : known;
        extra[orow] = kn | ((known_is_total && incomplete) ? XGM_EXTRA_LOWER_BOUND : 0ull);
    }
}

}  // namespace

int xgm_launch_frozen_finish(const xgm_dev_query* queries, uint32_t nq, const uint32_t* goff, const xgm_cand* cand, const xgm_group_hdr* ghdr, uint32_t k_stride_c,
                             const double* max_possible, const uint32_t* row_of, xgm_hit* hits, xgm_result_hdr* hdrs, unsigned long long* extra, uint32_t k_stride_out,
                             hipStream_t stream) {
    if (nq == 0) return 0;
    hipLaunchKernelGGL(xgm_frozen_finish_kernel, dim3(nq), dim3(64), 0, stream, queries, nq, goff, cand, ghdr, k_stride_c, max_possible, row_of, hits, hdrs, extra,
                       k_stride_out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xgm_launch_error("xgm_frozen_finish_kernel", (int)e, hipGetErrorString(e));
    return 0;
}
