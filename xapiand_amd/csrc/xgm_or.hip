/* Disjunction kernel (K3) for gfx950: one WAVE per work unit (a query and a contiguous range of
 * docid stripes), no workgroup barriers, a private LDS slice per wave.
 *
 * Reference semantics (OrPostList, src/xapian/matcher/orpostlist.cc:94-204): every document that
 * indexes at least one term matches; its weight is the sum of the matching leaves in the
 * association of the Huffman-shaped tree OrContext::postlist builds (api/queryinternal.cc:440-489);
 * when the running k-th best weight w_min exceeds what a subtree can still contribute, the node
 * decays to AND_MAYBE / AND (orpostlist.cc:35-78) — documents that can no longer enter the MSet are
 * skipped without being weighed.
 *
 * Here, per stripe of W docids:
 *   1. membership: the union bitmap of ALL terms (exact match count = its popcount) and the union of
 *      the ESSENTIAL terms only.  A term is non-essential when the sum of the weight upper bounds of it
 *      and of every term with a smaller bound is below the k-th best weight known so far (MaxScore —
 *      the set-at-a-time form of the reference's decay): a document matching only non-essential terms
 *      cannot reach the top k, so it is counted but never weighed.  Dense terms contribute their probe
 *      container's bitmap (one 16-byte load per lane), the others are block-decoded (K1) into the LDS
 *      bitmaps.
 *   2. candidates = set bits of the essential union, enumerated in docid order, <= kOrwCand at a
 *      time; per term the candidate's wdf comes from ONE byte of the probe container, or from a
 *      second decode of just those blocks whose docid range holds a candidate.
 *   3. BM25 (K4, fp64, bm25weight.cc:170-181) per present leaf, tree sum in the reference's
 *      association (absent leaf = -0.0, the identity of IEEE addition), top-k (K5) in the wave's LDS
 *      buffer under msetcmp_by_relevance<true> (msetcmp.cc:55-62).
 * The k-th best weight is shared between the units of a query through one 64-bit atomic max, so a
 * unit starts pruning with what the others have already seen.  Pruning never changes the result:
 * a skipped document's weight is provably below the final k-th weight (strict comparisons; ties are
 * always weighed).
 */
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>

#include "xgm_device.h"
#include "xgm_launch.h"
#include "xgm_wave.h"

namespace {

constexpr uint32_t kOrwCand = 1024;      /* candidates per scoring chunk: 8 lanes x 4 words x 32 slots */
constexpr uint32_t kOrwChunkLanes = 8;
constexpr uint32_t kOrwRegSparse = 4;    /* block-decoded terms whose headers are software-pipelined  */
constexpr uint32_t kNoDense = 0xFFFFFFFFu;

__host__ __device__ inline size_t orw_wave_bytes(uint32_t W, uint32_t T, uint32_t cap, size_t tab_elem, uint32_t spg) {
    size_t off = 0;
    off += (size_t)cap * 8;                                    /* tk_w */
    off += (size_t)T * 64 * 8;                                 /* val: per-lane leaf / node weights */
    off += (size_t)cap * 4;                                    /* tk_d */
    off += (size_t)kStageWords * 4;                            /* stage */
    off += (size_t)(W / 32u) * 4 * 2;                          /* bm_all, bm_ess */
    off += (size_t)2 * T * spg * 4;                            /* runs */
    off += (size_t)(W / 32u) * 2;                              /* rankw (u16) */
    off += (size_t)kOrwCand * 2;                               /* c_slot */
    off += (size_t)T * kOrwCand * tab_elem;                    /* c_w */
    off += (size_t)cap;                                        /* tk_m (u8) */
    return (off + 15) & ~(size_t)15;
}

/* bitonic sort of cap (power of two, >= 128) candidates by one wave; best first */
__device__ void orw_topk_sort(uint64_t* w, uint32_t* d, uint8_t* m, uint32_t cap, uint32_t lane) {
    for (uint32_t size = 2; size <= cap; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            wave_lds_fence();
            for (uint32_t i = lane; i < (cap >> 1); i += 64u) {
                const uint32_t lo = 2u * i - (i & (stride - 1u)), hi = lo + stride;
                const bool asc = ((lo & size) == 0);
                const uint64_t aw = w[lo], bw = w[hi];
                const uint32_t ad = d[lo], bd = d[hi];
                const bool swap = asc ? cand_before(bw, bd, aw, ad) : cand_before(aw, ad, bw, bd);
                if (swap) {
                    w[lo] = bw; w[hi] = aw; d[lo] = bd; d[hi] = ad;
                    const uint8_t am = m[lo], bm = m[hi];
                    m[lo] = bm; m[hi] = am;
                }
            }
        }
    }
    wave_lds_fence();
}

/* One posting block: payload (already loaded, 4 words per lane) -> LDS window -> two postings per lane.
 * SCATTER == false: set the docids' bits in bm_all (and bm_ess when the term is essential).
 * SCATTER == true : for postings that are candidates of the current chunk (bit set in bm_ess, word in
 *                   [wlo, whi)), store wdf+1 at the candidate's ordinal in `row`. */
template <typename TabT, bool SCATTER>
__device__ __forceinline__ void orw_block(const Words4& pv, uint32_t meta, uint32_t first, uint32_t* stage, uint32_t lane,
                                          uint32_t stripe_base, uint32_t* bm_all, uint32_t* bm_ess, bool ess,
                                          const uint16_t* rankw, TabT* row, uint32_t wlo, uint32_t whi) {
    if (lane * 4u < payload_words(meta)) {
        stage[lane * 4u] = pv.a; stage[lane * 4u + 1] = pv.b; stage[lane * 4u + 2] = pv.c; stage[lane * 4u + 3] = pv.d;
    }
    wave_lds_fence();
    const DecodedPair r = unpack_staged<false>(stage, first, meta, lane);
    wave_lds_fence();
#pragma unroll
    for (uint32_t h = 0; h < 2u; ++h) {
        const bool v = h ? r.v1 : r.v0;
        if (!v) continue;
        const uint32_t s = (h ? r.d1 : r.d0) - stripe_base, wd = s >> 5, bit = s & 31u;
        if (!SCATTER) {
            atomicOr(&bm_all[wd], 1u << bit);
            if (ess) atomicOr(&bm_ess[wd], 1u << bit);
        } else if (wd >= wlo && wd < whi) {
            const uint32_t bm = bm_ess[wd];
            if ((bm >> bit) & 1u) row[(uint32_t)rankw[wd] + (uint32_t)__popc(bm & ((1u << bit) - 1u))] = (TabT)((h ? r.w1 : r.w0) + 1u);
        }
    }
}

template <typename TabT>
__global__ __launch_bounds__(XGM_WG) void xgm_orw_kernel(xgm_seg_dev seg, const xgm_dev_query* __restrict__ queries,
                                                          const xgm_work* __restrict__ work, uint32_t n_work, uint32_t SPG,
                                                          uint32_t tab_terms, uint32_t cap, uint32_t k_stride,
                                                          unsigned long long* __restrict__ theta_g, int prune,
                                                          xgm_cand* __restrict__ cand_out, xgm_group_hdr* __restrict__ ghdr_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t unit = blockIdx.x * XGM_WAVES + wave;
    if (unit >= n_work) return;                                    /* no barriers below: early exit is safe */
    const xgm_work wk = work[unit];
    const xgm_dev_query& q = queries[wk.qi];
    const uint32_t SB = seg.stripe_bits, W = 1u << SB, NW = W / 32u;
    const uint32_t T = q.n_terms, k = q.k;
    const unsigned long long t_unit_start = __builtin_readcyclecounter();

    /* private LDS slice */
    unsigned char* base = smem + (size_t)wave * orw_wave_bytes(W, tab_terms, cap, sizeof(TabT), SPG);
    size_t off = 0;
    uint64_t* tk_w = reinterpret_cast<uint64_t*>(base + off); off += (size_t)cap * 8;
    double* val = reinterpret_cast<double*>(base + off); off += (size_t)tab_terms * 64 * 8;
    uint32_t* tk_d = reinterpret_cast<uint32_t*>(base + off); off += (size_t)cap * 4;
    uint32_t* stage = reinterpret_cast<uint32_t*>(base + off); off += (size_t)kStageWords * 4;
    uint32_t* bm_all = reinterpret_cast<uint32_t*>(base + off); off += (size_t)NW * 4;
    uint32_t* bm_ess = reinterpret_cast<uint32_t*>(base + off); off += (size_t)NW * 4;
    uint32_t* rs = reinterpret_cast<uint32_t*>(base + off); off += (size_t)tab_terms * SPG * 4;
    uint32_t* re = reinterpret_cast<uint32_t*>(base + off); off += (size_t)tab_terms * SPG * 4;
    uint16_t* rankw = reinterpret_cast<uint16_t*>(base + off); off += (size_t)NW * 2;
    uint16_t* c_slot = reinterpret_cast<uint16_t*>(base + off); off += (size_t)kOrwCand * 2;
    TabT* c_w = reinterpret_cast<TabT*>(base + off); off += (size_t)tab_terms * kOrwCand * sizeof(TabT);
    uint8_t* tk_m = reinterpret_cast<uint8_t*>(base + off);

    const uint32_t n_stripes = (seg.lastdocid >> SB) + 1u;
    const uint32_t s_begin = wk.s_begin, s_end = wk.s_end;
    const bool empty = (q.flags & XGM_QF_EMPTY) || s_begin >= s_end || k == 0;

    for (uint32_t i = lane; i < cap; i += 64u) { tk_w[i] = 0; tk_d[i] = 0xFFFFFFFFu; tk_m[i] = 0; }
    for (uint32_t i = lane; i < 2u * tab_terms * SPG; i += 64u) rs[i] = 0;        /* rs and re are adjacent */
    for (uint32_t i = lane; i < T * kOrwCand; i += 64u) c_w[i] = 0;
    wave_lds_fence();

    /* lane t keeps term t's payload base, dense-container index and MaxScore prefix bound */
    uint64_t tbase_reg = 0;
    uint32_t dense_reg = kNoDense;
    bool present_reg = false;
    if (!empty && lane < T) {
        const uint32_t id = q.term_id[lane];
        if (id != 0xFFFFFFFFu) {
            present_reg = true;
            tbase_reg = seg.term_word[id];
            if (sizeof(TabT) == 1 && seg.dense_id) dense_reg = seg.dense_id[id];
        }
    }
    const uint64_t present_mask = __ballot(present_reg);
    const uint64_t dense_mask = __ballot(present_reg && dense_reg != kNoDense);
    const uint64_t sparse_mask = present_mask & ~dense_mask;
    double prefix_reg = 0.0;
    if (lane < T) {
        const double my = q.ub[lane];
        for (uint32_t j = 0; j < T; ++j) {
            const double uj = q.ub[j];
            if (uj < my || (uj == my && j <= lane)) prefix_reg += uj;
        }
        prefix_reg *= 1.000000001;                                 /* covers the rounding of any summation order */
    }

    /* block ranges of every block-decoded term inside the unit's docid range -> run table */
    for (uint64_t sm = sparse_mask; sm; sm &= sm - 1u) {
        const uint32_t t = (uint32_t)__builtin_ctzll(sm);
        const uint32_t id = q.term_id[t];
        const uint32_t b0 = (uint32_t)seg.term_blk[id], b1 = (uint32_t)seg.term_blk[id + 1];
        const uint32_t c = wave_lower_bound(seg.blk_first, b0, b1, s_begin << SB, lane);
        const uint32_t e = (s_end >= n_stripes) ? b1 : wave_lower_bound(seg.blk_first, c, b1, s_end << SB, lane);
        for (uint32_t i = c + lane; i < e; i += 64u) {
            const uint32_t s = (seg.blk_first[i] >> SB) - s_begin;
            const uint32_t sp = i > c ? (seg.blk_first[i - 1] >> SB) - s_begin : 0xFFFFFFFFu;
            const uint32_t sn = i + 1 < e ? (seg.blk_first[i + 1] >> SB) - s_begin : 0xFFFFFFFFu;
            if (s != sp) rs[t * SPG + s] = i;
            if (s != sn) re[t * SPG + s] = i + 1u;
        }
    }
    wave_lds_fence();
    auto tbase = [&](uint32_t t) {
        return ((uint64_t)__builtin_amdgcn_readlane((uint32_t)(tbase_reg >> 32), t) << 32) | __builtin_amdgcn_readlane((uint32_t)tbase_reg, t);
    };
    /* the first kOrwRegSparse block-decoded terms get pipelined header registers */
    uint32_t sp_t[kOrwRegSparse];
    uint32_t n_sp = 0;
    {
        uint64_t sm = sparse_mask;
#pragma unroll
        for (uint32_t u = 0; u < kOrwRegSparse; ++u) {
            sp_t[u] = 0;
            if (sm) { sp_t[u] = (uint32_t)__builtin_ctzll(sm); sm &= sm - 1u; n_sp = u + 1u; }
        }
    }
    uint64_t slow_sparse_mask = sparse_mask;                       /* block-decoded terms beyond the pipelined ones */
    for (uint32_t u = 0; u < n_sp; ++u) slow_sparse_mask &= slow_sparse_mask - 1u;

    uint32_t tkn = 0;                                              /* wave-uniform top-k state */
    bool theta_valid = false;
    uint64_t theta_w = 0;
    uint32_t theta_d = 0;
    uint64_t theta_glob = 0;
    unsigned long long matches = 0;                                /* per lane, reduced at the end */
    const uint32_t n_local = empty ? 0u : s_end - s_begin;

    auto next_active = [&](uint32_t from) {
        if (dense_mask) return from < n_local ? from : n_local;    /* a dense term is (almost) everywhere */
        uint32_t x = from;
        for (; x < n_local; ++x) {
            bool any = false;
            for (uint64_t sm = sparse_mask; sm; sm &= sm - 1u) {
                const uint32_t t = (uint32_t)__builtin_ctzll(sm);
                any = any || (re[t * SPG + x] != rs[t * SPG + x]);
            }
            if (any) break;
        }
        return x;
    };

    /* software-pipelined per-stripe registers: lane j = block j of the pipelined terms' runs; lane t =
     * container offset of dense term t */
    uint32_t hm[kOrwRegSparse], hf[kOrwRegSparse], hw[kOrwRegSparse], hn[kOrwRegSparse];
#pragma unroll
    for (uint32_t u = 0; u < kOrwRegSparse; ++u) { hm[u] = hf[u] = hw[u] = 0; hn[u] = 0xFFFFFFFFu; }
    uint32_t hc_off = 0;
    auto issue_headers = [&](uint32_t x) {
#pragma unroll
        for (uint32_t u = 0; u < kOrwRegSparse; ++u) {
            if (u < n_sp) {
                const uint32_t rb = rs[sp_t[u] * SPG + x], nb = re[sp_t[u] * SPG + x] - rb;
                if (lane < nb) {
                    hm[u] = seg.blk_meta[rb + lane]; hf[u] = seg.blk_first[rb + lane]; hw[u] = seg.blk_word[rb + lane];
                    hn[u] = lane + 1u < nb ? seg.blk_first[rb + lane + 1u] : 0xFFFFFFFFu;
                }
            }
        }
        hc_off = 0;
        if (present_reg && dense_reg != kNoDense) hc_off = seg.dense_dir[(size_t)dense_reg * seg.n_stripes + (s_begin + x)];
    };

    uint32_t stripe_base = 0;
    uint32_t hc_cur = 0;

    /* BM25 + tree sum + top-k for the n_c candidates of the chunk; clears c_w behind itself */
    auto score_candidates = [&](uint32_t n_c) {
        for (uint32_t i0 = 0; i0 < n_c; i0 += 64u) {
            if (tkn + 64u > cap) {
                orw_topk_sort(tk_w, tk_d, tk_m, cap, lane);
                tkn = tkn < k ? tkn : k;
                if (tkn == k) {
                    theta_valid = true; theta_w = tk_w[k - 1]; theta_d = tk_d[k - 1];
                    if (prune && lane == 0) atomicMax(&theta_g[wk.qi], (unsigned long long)theta_w);
                }
                for (uint32_t i = tkn + lane; i < cap; i += 64u) { tk_w[i] = 0; tk_d[i] = 0xFFFFFFFFu; tk_m[i] = 0; }
                wave_lds_fence();
            }
            const uint32_t o = i0 + lane;
            const bool valid = o < n_c;
            const uint32_t did = stripe_base + (valid ? (uint32_t)c_slot[o] : 0u);
            const uint32_t dlen = valid ? seg.doclen[did] : 1u;
            /* BM25Weight::get_sumpart, bm25weight.cc:170-181 — same operations, same order */
            const double len = (double)dlen;
            double normlen = len * q.len_factor;
            normlen = normlen > q.min_normlen ? normlen : q.min_normlen;
            const double denom_len = q.k1 * (normlen * q.b + (1.0 - q.b));
            uint32_t subqs = 0;
            for (uint32_t t = 0; t < T; ++t) {
                const uint32_t e = valid ? (uint32_t)c_w[(size_t)t * kOrwCand + o] : 0u;
                double wt = -0.0;                                   /* absent leaf: x + (-0.0) == x */
                if (__ballot(e != 0u)) {                            /* nobody in the round has the term: skip the divide */
                    const double wdf = (double)(e - 1u);
                    const double denom = denom_len + wdf;
                    const double x = q.termweight[t] * (wdf / denom);
                    wt = e ? x : -0.0;
                    subqs += e ? 1u : 0u;
                    if (e) c_w[(size_t)t * kOrwCand + o] = 0;
                }
                val[t * 64u + lane] = wt;
            }
            /* OrPostList::get_weight: l + r up the tree (in place: node j lands in its left operand's slot) */
            for (uint32_t j = 0; j + 1u < T; ++j) {
                const uint32_t a = q.ip_a[j], b = q.ip_b[j];
                val[a * 64u + lane] = val[a * 64u + lane] + val[b * 64u + lane];
            }
            const double weight = val[(uint32_t)q.ip_root * 64u + lane];
            const uint64_t wb = (uint64_t)__double_as_longlong(weight);
            const bool take = valid && subqs != 0u && (!theta_valid || cand_before(wb, did, theta_w, theta_d)) && wb >= theta_glob;
            const uint64_t tm = __ballot(take);
            if (take) { const uint32_t p = tkn + mbcnt(tm); tk_w[p] = wb; tk_d[p] = did; tk_m[p] = (uint8_t)subqs; }
            tkn += (uint32_t)__popcll(tm);
        }
    };

    uint32_t sl = next_active(0);
    if (sl < n_local) issue_headers(sl);
    while (sl < n_local) {
        stripe_base = (s_begin + sl) << SB;
        const uint32_t sl_next = next_active(sl + 1u);
        hc_cur = hc_off;
        /* this stripe's block headers (the registers are re-used for the next stripe's prefetch) */
        uint32_t cm[kOrwRegSparse], cf[kOrwRegSparse], cw[kOrwRegSparse], cn[kOrwRegSparse], cnb[kOrwRegSparse];
#pragma unroll
        for (uint32_t u = 0; u < kOrwRegSparse; ++u) {
            cm[u] = hm[u]; cf[u] = hf[u]; cw[u] = hw[u]; cn[u] = hn[u];
            cnb[u] = u < n_sp ? re[sp_t[u] * SPG + sl] - rs[sp_t[u] * SPG + sl] : 0u;
        }

        /* ---- MaxScore: which terms are essential under the best threshold known ---- */
        uint64_t ess_mask = present_mask;
        if (prune) {
            theta_glob = __hip_atomic_load(&theta_g[wk.qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint64_t th_bits = theta_valid && theta_w > theta_glob ? theta_w : theta_glob;
            const double th = __longlong_as_double((long long)th_bits);
            ess_mask = __ballot(present_reg && !(prefix_reg < th));
        }

        /* ---- 1a. dense terms: OR of the containers' bitmaps (4 words per lane) ---- */
        uint32_t a[4] = {0, 0, 0, 0}, e[4] = {0, 0, 0, 0};
        for (uint64_t dm = dense_mask; dm;) {
            uint32_t tt[4], oo[4];
            uint32_t x[4][4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                tt[u] = 0; oo[u] = 0;
                if (dm) { tt[u] = (uint32_t)__builtin_ctzll(dm); dm &= dm - 1u; oo[u] = __builtin_amdgcn_readlane(hc_cur, tt[u]); }
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) {
                    x[u][i] = 0;
                    const uint32_t w = lane * 4u + i;
                    if (oo[u] && w < NW) x[u][i] = reinterpret_cast<const uint32_t*>(seg.dense_data + (size_t)oo[u] * 16)[w];
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                const bool es = (ess_mask >> tt[u]) & 1ull;
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) { a[i] |= x[u][i]; if (es) e[i] |= x[u][i]; }
            }
        }
        if (sparse_mask) {
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) {
                const uint32_t w = lane * 4u + i;
                if (w < NW) { bm_all[w] = a[i]; bm_ess[w] = e[i]; }
            }
            wave_lds_fence();
            /* ---- 1b. block-decoded terms: every block of the stripe ---- */
            uint64_t bmask[kOrwRegSparse];
#pragma unroll
            for (uint32_t u = 0; u < kOrwRegSparse; ++u) bmask[u] = cnb[u] >= 64u ? ~0ull : ((1ull << cnb[u]) - 1ull);
            while (true) {
                uint64_t any = 0;
#pragma unroll
                for (uint32_t u = 0; u < kOrwRegSparse; ++u) any |= bmask[u];
                if (!any) break;
                uint32_t jj[kOrwRegSparse];
                Words4 pv[kOrwRegSparse];
                bool have[kOrwRegSparse];
#pragma unroll
                for (uint32_t u = 0; u < kOrwRegSparse; ++u) {
                    have[u] = bmask[u] != 0ull;
                    jj[u] = 0; pv[u] = Words4{0, 0, 0, 0};
                    if (have[u]) {
                        jj[u] = (uint32_t)__builtin_ctzll(bmask[u]);
                        bmask[u] &= bmask[u] - 1u;
                        const uint32_t bm = __builtin_amdgcn_readlane(cm[u], jj[u]);
                        if (lane * 4u < payload_words(bm))
                            pv[u] = *reinterpret_cast<const Words4*>(seg.words + tbase(sp_t[u]) + __builtin_amdgcn_readlane(cw[u], jj[u]) + lane * 4u);
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < kOrwRegSparse; ++u) {
                    if (have[u])
                        orw_block<TabT, false>(pv[u], __builtin_amdgcn_readlane(cm[u], jj[u]), __builtin_amdgcn_readlane(cf[u], jj[u]), stage, lane,
                                               stripe_base, bm_all, bm_ess, (ess_mask >> sp_t[u]) & 1ull, rankw, c_w, 0u, 0u);
                }
            }
            for (uint64_t sm = slow_sparse_mask; sm; sm &= sm - 1u) {
                const uint32_t t = (uint32_t)__builtin_ctzll(sm);
                const uint32_t rb = rs[t * SPG + sl], nb = re[t * SPG + sl] - rb;
                for (uint32_t j = 0; j < nb; ++j) {
                    const uint32_t meta = seg.blk_meta[rb + j], first = seg.blk_first[rb + j];
                    Words4 pv = Words4{0, 0, 0, 0};
                    if (lane * 4u < payload_words(meta)) pv = *reinterpret_cast<const Words4*>(seg.words + tbase(t) + seg.blk_word[rb + j] + lane * 4u);
                    orw_block<TabT, false>(pv, meta, first, stage, lane, stripe_base, bm_all, bm_ess, (ess_mask >> t) & 1ull, rankw, c_w, 0u, 0u);
                }
            }
            wave_lds_fence();
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) {
                const uint32_t w = lane * 4u + i;
                if (w < NW) { a[i] = bm_all[w]; e[i] = bm_ess[w]; }
            }
        }

        /* ---- exact match count; candidates = essential union ---- */
        matches += (unsigned long long)(__popc(a[0]) + __popc(a[1]) + __popc(a[2]) + __popc(a[3]));
        const uint32_t cnt = (uint32_t)(__popc(e[0]) + __popc(e[1]) + __popc(e[2]) + __popc(e[3]));
        const uint32_t incl = wave_incl_scan(cnt);
        const uint32_t n_total = __builtin_amdgcn_readlane(incl, 63);
        if (n_total == 0u) {
            if (sl_next < n_local) issue_headers(sl_next);
            sl = sl_next;
            continue;
        }
        const bool single = n_total <= kOrwCand;
        const uint32_t n_chunks = single ? 1u : 64u / kOrwChunkLanes;
        for (uint32_t c = 0; c < n_chunks; ++c) {
            const uint32_t lane_lo = single ? 0u : c * kOrwChunkLanes;
            const uint32_t lane_hi = single ? 64u : lane_lo + kOrwChunkLanes;
            const uint32_t ord_base = lane_lo ? __builtin_amdgcn_readlane(incl, lane_lo - 1u) : 0u;
            const uint32_t n_c = __builtin_amdgcn_readlane(incl, lane_hi - 1u) - ord_base;
            const bool last_chunk = c + 1u == n_chunks;
            if (n_c == 0u) {
                if (last_chunk && sl_next < n_local) issue_headers(sl_next);
                continue;
            }
            const bool in_chunk = lane >= lane_lo && lane < lane_hi;
            const uint32_t wlo = lane_lo * 4u, whi = lane_hi * 4u;
            /* ---- 2a. enumerate the chunk's candidates in docid order ---- */
            if (in_chunk) {
                uint32_t o = incl - cnt - ord_base;
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) {
                    const uint32_t w = lane * 4u + i;
                    if (w < NW) rankw[w] = (uint16_t)o;
                    uint32_t m = e[i];
                    while (m) {
                        const uint32_t bit = (uint32_t)__ffs(m) - 1u;
                        c_slot[o] = (uint16_t)(w * 32u + bit);
                        m &= m - 1u;
                        ++o;
                    }
                }
            }
            const unsigned long long coarse = __ballot(in_chunk && cnt != 0u);   /* bit = 128-slot bucket with a candidate */
            wave_lds_fence();

            /* ---- 2b. wdf of the dense terms: one byte per candidate and term ---- */
            for (uint32_t c0 = 0; c0 < n_c; c0 += 64u) {
                const uint32_t o = c0 + lane;
                const bool valid = o < n_c;
                const uint32_t slot = valid ? c_slot[o] : 0u;
                for (uint64_t dm = dense_mask; dm;) {
                    uint32_t tt[4], wv[4];
                    bool on[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) {
                        tt[u] = 0; wv[u] = 0; on[u] = false;
                        if (dm) {
                            tt[u] = (uint32_t)__builtin_ctzll(dm); dm &= dm - 1u;
                            const uint32_t oo = __builtin_amdgcn_readlane(hc_cur, tt[u]);
                            on[u] = oo != 0u;
                            if (on[u] && valid) wv[u] = seg.dense_data[(size_t)oo * 16 + (size_t)NW * 4 + slot];
                        }
                    }
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u)
                        if (on[u] && valid && wv[u]) c_w[(size_t)tt[u] * kOrwCand + o] = (TabT)wv[u];
                }
            }

            /* ---- 2c. wdf of the block-decoded terms: only blocks whose buckets hold a candidate ---- */
            auto bucket_need = [&](uint32_t first, uint32_t nfirst) {
                const uint32_t lo = (first - stripe_base) >> 7;
                const uint32_t hi = ((nfirst == 0xFFFFFFFFu ? W : nfirst - stripe_base) - 1u) >> 7;
                const unsigned long long mm = (hi >= 63u ? ~0ull : ((1ull << (hi + 1u)) - 1ull)) & ~((1ull << lo) - 1ull);
                return (coarse & mm) != 0ull;
            };
            if (sparse_mask) {
                uint64_t bmask[kOrwRegSparse];
#pragma unroll
                for (uint32_t u = 0; u < kOrwRegSparse; ++u) bmask[u] = __ballot(lane < cnb[u] && bucket_need(cf[u], cn[u]));
                while (true) {
                    uint64_t any = 0;
#pragma unroll
                    for (uint32_t u = 0; u < kOrwRegSparse; ++u) any |= bmask[u];
                    if (!any) break;
                    uint32_t jj[kOrwRegSparse];
                    Words4 pv[kOrwRegSparse];
                    bool have[kOrwRegSparse];
#pragma unroll
                    for (uint32_t u = 0; u < kOrwRegSparse; ++u) {
                        have[u] = bmask[u] != 0ull;
                        jj[u] = 0; pv[u] = Words4{0, 0, 0, 0};
                        if (have[u]) {
                            jj[u] = (uint32_t)__builtin_ctzll(bmask[u]);
                            bmask[u] &= bmask[u] - 1u;
                            const uint32_t bm = __builtin_amdgcn_readlane(cm[u], jj[u]);
                            if (lane * 4u < payload_words(bm))
                                pv[u] = *reinterpret_cast<const Words4*>(seg.words + tbase(sp_t[u]) + __builtin_amdgcn_readlane(cw[u], jj[u]) + lane * 4u);
                        }
                    }
#pragma unroll
                    for (uint32_t u = 0; u < kOrwRegSparse; ++u) {
                        if (have[u])
                            orw_block<TabT, true>(pv[u], __builtin_amdgcn_readlane(cm[u], jj[u]), __builtin_amdgcn_readlane(cf[u], jj[u]), stage, lane,
                                                  stripe_base, bm_all, bm_ess, false, rankw, c_w + (size_t)sp_t[u] * kOrwCand, wlo, whi);
                    }
                }
                for (uint64_t sm = slow_sparse_mask; sm; sm &= sm - 1u) {
                    const uint32_t t = (uint32_t)__builtin_ctzll(sm);
                    const uint32_t rb = rs[t * SPG + sl], nb = re[t * SPG + sl] - rb;
                    for (uint32_t j = 0; j < nb; ++j) {
                        const uint32_t meta = seg.blk_meta[rb + j], first = seg.blk_first[rb + j];
                        const uint32_t nfirst = j + 1u < nb ? seg.blk_first[rb + j + 1u] : 0xFFFFFFFFu;
                        if (!bucket_need(first, nfirst)) continue;
                        Words4 pv = Words4{0, 0, 0, 0};
                        if (lane * 4u < payload_words(meta)) pv = *reinterpret_cast<const Words4*>(seg.words + tbase(t) + seg.blk_word[rb + j] + lane * 4u);
                        orw_block<TabT, true>(pv, meta, first, stage, lane, stripe_base, bm_all, bm_ess, false, rankw, c_w + (size_t)t * kOrwCand, wlo, whi);
                    }
                }
            }
            wave_lds_fence();

            /* headers of the next active stripe: in flight while this chunk is scored */
            if (last_chunk && sl_next < n_local) issue_headers(sl_next);

            /* ---- 3. BM25, tree sum, top-k ---- */
            score_candidates(n_c);
            wave_lds_fence();
        }
        sl = sl_next;
    }

    /* ---- unit epilogue ---- */
    orw_topk_sort(tk_w, tk_d, tk_m, cap, lane);
    for (int sh = 32; sh > 0; sh >>= 1) matches += (unsigned long long)__shfl_xor((long long)matches, sh);
    const uint32_t n_out = tkn < k ? tkn : k;
    if (prune && n_out == k && lane == 0) atomicMax(&theta_g[wk.qi], (unsigned long long)tk_w[k - 1]);
    xgm_cand* out = cand_out + (size_t)wk.slot * k_stride;
    for (uint32_t i = lane; i < n_out; i += 64u) {
        xgm_cand c;
        c.wbits = tk_w[i]; c.did = tk_d[i]; c.subqs = tk_m[i];
        out[i] = c;
    }
    if (lane == 0) {
        xgm_group_hdr h;
        h.matches = matches; h.n_cand = n_out; h.pad = 0;
        h.t_start = t_unit_start; h.t_end = __builtin_readcyclecounter();
        ghdr_out[wk.slot] = h;
    }
}

template <class K>
int orw_ensure_dyn_smem(K kern, size_t smem, std::atomic<size_t>& seen) {
    if (smem <= seen.load(std::memory_order_relaxed)) return XGM_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return xgm_launch_error("hipFuncSetAttribute(xgm_orw_kernel)", (int)e, hipGetErrorString(e));
    seen.store(smem, std::memory_order_relaxed);
    return XGM_OK;
}

}  // namespace

size_t xgm_orw_smem_bytes(uint32_t stripe_bits, uint32_t tab_terms, uint32_t cap, bool wide, uint32_t spg) {
    return XGM_WAVES * orw_wave_bytes(1u << stripe_bits, tab_terms, cap, wide ? 2 : 1, spg);
}

int xgm_launch_orw(const xgm_match_launch& L, unsigned long long* theta_g, hipStream_t stream) {
    static const bool no_prune = getenv("XGM_NO_PRUNE") != nullptr;          /* A/B switch for measurements */
    const size_t smem = xgm_orw_smem_bytes(L.seg.stripe_bits, L.tab_terms, L.cap, L.wide, L.stripes_per_group);
    const dim3 grid((L.n_work + XGM_WAVES - 1u) / XGM_WAVES), block(XGM_WG);
    static std::atomic<size_t> seen8{0}, seen16{0};
    int rc;
    if (L.wide) {
        auto kern = xgm_orw_kernel<uint16_t>;
        if ((rc = orw_ensure_dyn_smem(kern, smem, seen16))) return rc;
        hipLaunchKernelGGL(kern, grid, block, smem, stream, L.seg, L.queries, L.work, L.n_work, L.stripes_per_group, L.tab_terms, L.cap,
                           L.k_stride, theta_g, no_prune ? 0 : 1, L.cand, L.ghdr);
    } else {
        auto kern = xgm_orw_kernel<uint8_t>;
        if ((rc = orw_ensure_dyn_smem(kern, smem, seen8))) return rc;
        hipLaunchKernelGGL(kern, grid, block, smem, stream, L.seg, L.queries, L.work, L.n_work, L.stripes_per_group, L.tab_terms, L.cap,
                           L.k_stride, theta_g, no_prune ? 0 : 1, L.cand, L.ghdr);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xgm_launch_error("xgm_orw_kernel launch", (int)e, hipGetErrorString(e));
    return XGM_OK;
}
