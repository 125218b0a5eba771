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
 *   1. membership: the union bitmap of ALL terms (exact match count = its popcount).  Dense terms contribute
 *      their probe container's bitmap (one 16-byte load per lane), the others are block-decoded (K1), one
 *      term at a time, into an LDS bitmap pair: the term's documents, and those whose wdf is >= 2.
 *   2. candidates = the documents whose weight BOUND reaches the threshold (below), enumerated in docid
 *      order and QUEUED in LDS across stripes (a stripe leaves a handful); per term a candidate's wdf
 *      comes from ONE byte of the probe container of its own stripe, or — block-decoded terms — from a
 *      second decode of just those blocks whose docid range holds a candidate, scattered into the queue
 *      while the stripe's blocks are at hand.
 *   3. BM25 (K4, fp64, bm25weight.cc:170-181) per present leaf, tree sum in the reference's
 *      association (absent leaf = -0.0, the identity of IEEE addition), top-k (K5) in the wave's LDS
 *      buffer under msetcmp_by_relevance<true> (msetcmp.cc:55-62) — in full 64-lane rounds, once enough
 *      candidates wait.
 * The pruning threshold is GLOBAL per query: every weighed document is counted in a 256-bucket
 * histogram of weight bit patterns (32 buckets per octave below the query's weight upper bound) shared
 * by all units of the query through global atomics; the highest bucket with >= k documents at or above
 * it is a lower bound of the final k-th weight.
 *
 * Which documents are weighed: those whose weight BOUND reaches the threshold.  A (document, term) pair
 * is bounded by the term's weight at wdf = 1 or, when the container's second bitmap says wdf >= 2, at
 * the term's largest wdf (both with the shortest document's length); the bounds are quantised UP to
 * 1/64 of the threshold and summed for 32 documents at a time by a bit-sliced adder.  The threshold of
 * the first pass starts at the planner's GUESS of the final k-th weight (xgm_dev_query::theta_seed)
 * and follows the histogram upwards; the pass also counts the matches.  If, when a unit has finished
 * its stripes, fewer than k documents of the query are known at or above the guess, the guess was too
 * high: the unit goes over its stripes again with the threshold it has, weighing the documents between
 * that and the guess — those the first pass weighed (recognised per document by the same quantised sum)
 * are skipped, so none is counted twice.  Pruning never changes the result: a skipped document's weight
 * is provably below the final k-th weight (strict comparisons; ties are always weighed).
 */
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>

#include "xgm_device.h"
#include "xgm_launch.h"
#include "xgm_wave.h"
#include "xgm_unit_finish.h"

namespace {

#ifndef XGM_ORW_CAND
#define XGM_ORW_CAND 384            /* (512 until round 6: with 384 three workgroups' LDS slices fit a CU at C3's shape — see XGM_ORW_MINWG) */
#endif
constexpr uint32_t kOrwCand = XGM_ORW_CAND;   /* candidates per scoring chunk */
#ifndef XGM_ORW_REGSPARSE
#define XGM_ORW_REGSPARSE 2
#endif
#ifndef XGM_ORW_MINWG
#define XGM_ORW_MINWG 3             /* round 6: once the unit, the query and the threshold live in scalar registers the kernel needs 189 VGPRs; capped at 168 it
                                       spills 14 and runs three waves per SIMD: 1.50 -> 1.30 ms per launch of C3 (round 5's attempts spilled 87-136 and lost) */
#endif
#ifndef XGM_ORW_GROUP
#define XGM_ORW_GROUP 4             /* dense terms whose bitmaps (+ wdf >= 2 bitmaps) are in flight together (5 until round 6: 40 registers of bitmaps; at three
                                       waves per SIMD 4 and 3 measure 1.18 ms per launch of C3, 5 1.30, 2 1.22) */
#endif
#ifndef XGM_ORW_TIMERS
#define XGM_ORW_TIMERS 0            /* section timers cost ~30 VGPRs: A/B builds only (tools/ab_build.sh) */
#endif
constexpr uint32_t kOrwRegSparse = XGM_ORW_REGSPARSE;   /* block-decoded terms whose headers are software-pipelined */
static_assert(kOrwRegSparse >= 1u && kOrwRegSparse <= 2u, "the per-term decode of xgm_orw_kernel picks between two header register sets");
#ifndef XGM_ORW_FLPRE
#define XGM_ORW_FLPRE 2             /* flat terms whose next postings are prefetched a stripe ahead */
#endif
constexpr uint32_t kFlPre = XGM_ORW_FLPRE;
constexpr uint32_t kNoDense = 0xFFFFFFFFu;
constexpr uint32_t kQ = 64;              /* quantisation of a weight bound relative to the threshold  */
constexpr uint32_t kHistShift = 47;      /* weight bits >> 47: sign, exponent, 5 mantissa bits         */

__host__ __device__ inline size_t orw_wave_bytes(uint32_t W, uint32_t T, uint32_t cap, size_t tab_elem, uint32_t spg) {
    size_t off = 0;
    off += (size_t)cap * 8;                                    /* tk_w */
    off += T > 8u ? (size_t)T * 64 * 8 : 0;                    /* val: per-lane leaf / node weights (queries of > 8 terms only: fewer sum in registers) */
    off += (size_t)cap * 4;                                    /* tk_d */
    off += (size_t)kStageWords * 4;                            /* stage */
    off += (size_t)(W / 32u) * 4 * 2;                          /* bm_ess, bm_ne */
    off += (size_t)XGM_OR_HIST * 4;                            /* lh: the wave's pending histogram counts */
    off += (size_t)2 * T * spg * 4;                            /* runs */
    off += (size_t)T * spg * 4;                                /* dir_tab: container offsets of the dense terms in the unit's stripes */
    off += (size_t)(W / 32u) * 2;                              /* rankw (u16) */
    off += (size_t)kOrwCand * 4;                               /* c_did */
    off += (size_t)T * kOrwCand * tab_elem;                    /* c_w */
    off += (size_t)cap;                                        /* tk_m (u8) */
    return (off + 15) & ~(size_t)15;
}

/* bitonic sort of cap (power of two, >= 128) candidates by one wave; best first */
__device__ void orw_topk_sort(uint64_t* w_, uint32_t* d_, uint8_t* m_, uint32_t cap, uint32_t lane) {
    XGM_AS_LDS uint64_t* w = (XGM_AS_LDS uint64_t*)w_;                               /* (the wave's top-k buffer lives in its LDS slice) */
    XGM_AS_LDS uint32_t* d = (XGM_AS_LDS uint32_t*)d_;
    XGM_AS_LDS uint8_t* m = (XGM_AS_LDS uint8_t*)m_;
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
 * SCATTER == false: set the docids' bits in bm_a and, for postings whose wdf is >= 2, in bm_b (bm_c: unused).
 * SCATTER == true : for postings that are candidates of the current chunk (bit set in bm_a, word in
 *                   [wlo, whi)), store wdf+1 at the candidate's ordinal in `row`. */
template <typename TabT, bool SCATTER>
__device__ __forceinline__ void orw_block(const Words4& pv, uint32_t meta, uint32_t first, uint32_t* stage, uint32_t lane,
                                          uint32_t stripe_base, uint32_t* bm_a, uint32_t* bm_b, uint32_t* bm_c,
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
            atomicOr(&bm_a[wd], 1u << bit);
            if ((h ? r.w1 : r.w0) >= 2u) atomicOr(&bm_b[wd], 1u << bit);
        } else if (wd >= wlo && wd < whi) {
            const uint32_t bm = bm_a[wd];
            if ((bm >> bit) & 1u) row[(uint32_t)rankw[wd] + (uint32_t)__popc(bm & ((1u << bit) - 1u))] = (TabT)((h ? r.w1 : r.w0) + 1u);
        }
    }
}

typedef double orw_d8 __attribute__((ext_vector_type(8)));
typedef uint32_t orw_u4 __attribute__((ext_vector_type(4)));

/* (mask & a) | (~mask & b): v_bfi_b32.  With mask = x ^ y: the majority of (x, y, a) when b = x — the carry of a full adder. */
__device__ __forceinline__ uint32_t orw_bfi(uint32_t mask, uint32_t a, uint32_t b) {
    return (mask & a) | (~mask & b);
}

/* FLAT (round 5): every term without a container is read from its flat posting array (docids + one wdf byte, decoded once when the index was
 * opened: xgm_dense.hip) — a cursor per term, 64 postings per coalesced load — instead of decoding its blocks twice per stripe (once for the
 * bitmaps, once more for the candidates' wdf): no run table, no block headers, no payload staging, no unpack.  plan_batch selects the
 * instantiation when every such term of the batch has an array (XGM_NO_OR_FLAT=1: A/B switch, the variant tests). */
/* PL (round 6): planes of the bit-sliced bound sum = how finely the weight bounds are quantised, kq = 2^PL steps of the threshold.  6 for every batch
 * until round 6; 4 where every query of the batch has 4-8 terms (plan_batch) — measured per launch of 256 queries at 10 M documents, 6 / 5 / 4 planes:
 * OR-2 top-10 0.748 / 0.781 / 1.174 ms, OR-3 top-10 0.794 / 0.766 / 0.782, OR-5 top-10 1.034 / 0.988 / 0.950, OR-5 top-100 (C3) 1.175 / 1.122 / 1.087,
 * OR-8 top-100 2.116 / 2.023 / 1.934; 3 planes: worse everywhere.  A plane less is a sixth of the sum's instructions and four registers saved on every
 * stripe, paid for with the documents the coarser rounding lets through — few where many terms share the threshold, many where two do.  A run-time
 * plane count over the 6-plane code kept a third of the gain (1.163 ms): the planes' code and registers have to go.  Pruning is exact either way: the
 * bounds are rounded UP. */
template <typename TabT, bool TALLY, bool FLAT, uint32_t PL>
__global__ __launch_bounds__(XGM_WG, XGM_ORW_MINWG) void xgm_orw_kernel(xgm_seg_dev seg, const xgm_dev_query* __restrict__ queries,
                                                          const xgm_work* __restrict__ work, uint32_t n_work, uint32_t SPG,
                                                          uint32_t tab_terms, uint32_t cap, uint32_t k_stride,
                                                          uint32_t* __restrict__ hist_all, int prune_flags,
                                                          xgm_cand* __restrict__ cand_out, xgm_group_hdr* __restrict__ ghdr_out,
                                                          unsigned long long* __restrict__ phase_cycles, const xgm_fuse* __restrict__ fuse) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    /* (the wave's index through readfirstlane: the compiler cannot prove threadIdx.x >> 6 uniform, and with it the unit, the query and every loop
     *  bound read from them would live in vector registers — loops over the query's terms under exec masks, the query's fields fetched with vector
     *  loads, copies at every join: round 6, as xgm_dense_unit since round 3) */
    const uint32_t lane = threadIdx.x & 63u, wave = rfl32(threadIdx.x >> 6);
    const uint32_t unit = blockIdx.x * XGM_WAVES + wave;
    if (unit >= n_work) return;                                    /* no barriers below: early exit is safe */
    const xgm_work wk = work[unit];
    const xgm_dev_query& q = queries[wk.qi];
    const uint32_t SB = seg.stripe_bits, W = 1u << SB, NW = W / 32u;
    const uint32_t T = q.n_terms, k = q.k;
    constexpr uint32_t pl_q = PL, kq = 1u << PL;
    const unsigned long long t_unit_start = __builtin_readcyclecounter();
    /* traffic tallies (xgm_group_hdr): wave-uniform, kept in scalar registers */
    uint32_t cn_bmpw = 0, cn_probe = 0, cn_blkw = 0, cn_hdr = 0, cn_dl = 0, cn_aux = 0, cn_probe_raw = 0, cn_dl_raw = 0;
    uint32_t cn_first = 0, cn_fixw = 0;
    /* lanes hold ascending keys: how many distinct (key >> sh) values = memory sectors does one gather round touch? */
    auto tally_sectors = [&](bool valid, uint32_t key, uint32_t sh) {
        const uint32_t prev = (uint32_t)__shfl_up((int)key, 1);
        return (uint32_t)__popcll(__ballot(valid && (lane == 0u || (prev >> sh) != (key >> sh))));
    };
#define XGM_SU(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
    /* the in-place summation program of queries with <= 8 terms, in scalar registers */
    const uint32_t* ipa32 = reinterpret_cast<const uint32_t*>(q.ip_a);
    const uint32_t* ipb32 = reinterpret_cast<const uint32_t*>(q.ip_b);
    const uint64_t prog_a = ((uint64_t)rfl32(ipa32[1]) << 32) | rfl32(ipa32[0]);
    const uint64_t prog_b = ((uint64_t)rfl32(ipb32[1]) << 32) | rfl32(ipb32[0]);
    const uint32_t prog_root = __builtin_amdgcn_readfirstlane(q.ip_root);

    /* private LDS slice */
    unsigned char* base = smem + (size_t)wave * orw_wave_bytes(W, tab_terms, cap, sizeof(TabT), SPG);
    size_t off = 0;
    uint64_t* tk_w = reinterpret_cast<uint64_t*>(base + off); off += (size_t)cap * 8;
    double* val = reinterpret_cast<double*>(base + off); off += tab_terms > 8u ? (size_t)tab_terms * 64 * 8 : 0;
    uint32_t* tk_d = reinterpret_cast<uint32_t*>(base + off); off += (size_t)cap * 4;
    uint32_t* stage = reinterpret_cast<uint32_t*>(base + off); off += (size_t)kStageWords * 4;
    uint32_t* bm_ess = reinterpret_cast<uint32_t*>(base + off); off += (size_t)NW * 4;
    uint32_t* bm_ne = reinterpret_cast<uint32_t*>(base + off); off += (size_t)NW * 4;
    uint32_t* lh = reinterpret_cast<uint32_t*>(base + off); off += (size_t)XGM_OR_HIST * 4;
    uint32_t* rs = reinterpret_cast<uint32_t*>(base + off); off += (size_t)tab_terms * SPG * 4;
    uint32_t* re = reinterpret_cast<uint32_t*>(base + off); off += (size_t)tab_terms * SPG * 4;
    uint32_t* dir_tab = reinterpret_cast<uint32_t*>(base + off); off += (size_t)tab_terms * SPG * 4;
    uint16_t* rankw = reinterpret_cast<uint16_t*>(base + off); off += (size_t)NW * 2;
    uint32_t* c_did = reinterpret_cast<uint32_t*>(base + off); off += (size_t)kOrwCand * 4;   /* the candidate queue: docids (of several stripes) awaiting their weights */
    TabT* c_w = reinterpret_cast<TabT*>(base + off); off += (size_t)tab_terms * kOrwCand * sizeof(TabT);
    uint8_t* tk_m = reinterpret_cast<uint8_t*>(base + off);

    const uint32_t n_stripes = (seg.lastdocid >> SB) + 1u;
    const uint32_t s_begin = wk.s_begin, s_end = wk.s_end;
    const bool empty = (q.flags & XGM_QF_EMPTY) || s_begin >= s_end || k == 0;

    for (uint32_t i = lane; i < cap; i += 64u) { tk_w[i] = 0; tk_d[i] = 0xFFFFFFFFu; tk_m[i] = 0; }
    for (uint32_t i = lane; i < 2u * tab_terms * SPG; i += 64u) rs[i] = 0;        /* rs and re are adjacent */
    for (uint32_t i = lane; i < T * kOrwCand; i += 64u) c_w[i] = 0;
    for (uint32_t i = lane; i < XGM_OR_HIST; i += 64u) lh[i] = 0;
    wave_lds_fence();

    /* lane t keeps term t's payload base, dense-container index and MaxScore prefix bound */
    uint64_t tbase_reg = 0;
    uint32_t dense_reg = kNoDense;
    bool present_reg = false;
    uint32_t fl_cur = 0, fl_end = 0, fl_c0 = 0, fl_n = 0;           /* FLAT, lane t: term t's cursor into flat_did / end of its slice; this stripe's first posting and count */
    if (!empty && lane < T) {
        const uint32_t id = q.term_id[lane];
        if (id != 0xFFFFFFFFu) {
            present_reg = true;
            tbase_reg = seg.term_word[id];
            if (sizeof(TabT) == 1 && seg.dense_id) dense_reg = seg.dense_id[id];
            if (FLAT && dense_reg == kNoDense) { fl_cur = (uint32_t)seg.flat_off[id]; fl_end = (uint32_t)seg.flat_off[id + 1]; }
        }
    }
    const uint64_t present_mask = __ballot(present_reg);
    const uint64_t dense_mask = __ballot(present_reg && dense_reg != kNoDense);
    const uint64_t sparse_mask = present_mask & ~dense_mask;
    /* MaxScore order: terms by ascending weight upper bound; prefix_reg = sum of the bounds up to and
     * including this term's.  rank_reg = position in that order. */
    double prefix_reg = 0.0, ub_reg = 0.0, ub1_reg = 0.0;
    uint32_t rank_reg = 0;
    const bool no_sum = (prune_flags & 4) != 0;                    /* A/B: term-level MaxScore only */
    if (lane < T) {
        const double my = q.ub[lane];
        ub_reg = my;
        /* without a wdf >= 2 bitmap every document of the term is bounded by the term's maximum */
        ub1_reg = (dense_reg != kNoDense && !seg.dense_plane) ? my : q.ub1[lane];     /* (block-decoded terms: the decode itself tells wdf = 1 from wdf >= 2) */
        for (uint32_t j = 0; j < T; ++j) {
            const double uj = q.ub[j];
            if (uj < my || (uj == my && j <= lane)) { prefix_reg += uj; ++rank_reg; }
        }
        prefix_reg *= 1.000000001;                                 /* covers the rounding of any summation order */
    }
    /* the dense terms in ascending order of their bounds, 4 bits each (the bound sum adds them in this order) */
    uint64_t dense_ord = 0;
    uint32_t n_dense_ord = 0;
    for (uint32_t p = 1; p <= T; ++p) {
        const uint64_t m = __ballot(lane < T && rank_reg == p);
        if (m) {
            const uint32_t t = (uint32_t)__builtin_ctzll(m);
            if ((dense_mask >> t) & 1ull) { dense_ord |= (uint64_t)t << (4u * n_dense_ord); ++n_dense_ord; }
        }
    }
    /* the prefix of the term with the largest bound is the bound of any document's weight */
    const uint64_t top_mask = __ballot(present_reg && rank_reg == T);
    const uint32_t r_term = top_mask ? (uint32_t)__builtin_ctzll(top_mask) : 0u;
    const uint64_t mp_bits = top_mask ? rl64((uint64_t)__double_as_longlong(prefix_reg), r_term) : 0ull;
    const int hbase = (int)(mp_bits >> kHistShift) - (int)(XGM_OR_HIST - 1u);
    const bool prune = (prune_flags & 1) && top_mask != 0ull && hbase > 0 && !empty;
    uint32_t* hist_g = hist_all + (size_t)wk.qi * XGM_OR_HIST;
    /* the planner's guess of the final k-th weight, rounded DOWN to a histogram bucket edge (k documents at or above it then lift the
     * histogram's bound to exactly it); none when it falls below the histogram's range */
    uint64_t seed_bits = 0;
    if (prune && (prune_flags & 2) && !no_sum) {
        const uint64_t sb = (uint64_t)__double_as_longlong(q.theta_seed) >> kHistShift;
        const uint32_t sb_lo = __builtin_amdgcn_readfirstlane((uint32_t)sb);
        const int sbk = (int)sb_lo - hbase;
        if (sbk >= 1) seed_bits = (uint64_t)(sbk > (int)XGM_OR_HIST - 1 ? (uint32_t)hbase + XGM_OR_HIST - 1u : sb_lo) << kHistShift;
    }

    if (FLAT) {
        /* the first posting at or after the unit's first docid: a 64-ary search of the term's slice, once per unit */
        for (uint64_t sm = sparse_mask; sm; sm &= sm - 1u) {
            const uint32_t t = (uint32_t)__builtin_ctzll(sm);
            const uint32_t lo = rl32(fl_cur, t), hi = rl32(fl_end, t);
            const uint32_t c = wave_lower_bound(seg.flat_did + lo, 0u, hi - lo, s_begin << SB, lane);
            if (lane == t) fl_cur = lo + c;
            if (TALLY) { cn_aux += 64u; }
        }
    }
    const uint32_t fl_start = fl_cur;                              /* (the repair pass walks the slices again) */
    /* FLAT, round 6: the next 64 postings from the cursor of the first kFlPre flat terms are requested as soon as the cursor has moved — a stripe
     * ahead of their use — instead of when the stripe's term loop gets to them (one exposed memory round trip per flat term and stripe: the
     * section took 20 % of the kernel's cycles on C3) */
    uint32_t pre_t[kFlPre], pre_d[kFlPre], pre_w[kFlPre];
    uint32_t n_pre = 0;
    {
        uint64_t sm = FLAT ? sparse_mask : 0ull;
#pragma unroll
        for (uint32_t u = 0; u < kFlPre; ++u) {
            pre_t[u] = 0xFFu; pre_d[u] = 0xFFFFFFFFu; pre_w[u] = 0u;
            if (sm) { pre_t[u] = (uint32_t)__builtin_ctzll(sm); sm &= sm - 1u; n_pre = u + 1u; }
        }
    }
    auto flat_prefetch = [&](uint32_t u) {                        /* (u: compile-time after unrolling) */
        const uint32_t t = pre_t[u] & 63u;
        const uint32_t g = rl32(fl_cur, t) + lane, e_ = rl32(fl_end, t);
        const bool in_arr = u < n_pre && g < e_;
        pre_d[u] = in_arr ? seg.flat_did[g] : 0xFFFFFFFFu;
        pre_w[u] = in_arr ? (uint32_t)seg.flat_wdf[g] : 0u;
    };
    if (FLAT) {
#pragma unroll
        for (uint32_t u = 0; u < kFlPre; ++u) flat_prefetch(u);
    }
    /* block ranges of every block-decoded term inside the unit's docid range -> run table */
    for (uint64_t sm = FLAT ? 0ull : sparse_mask; sm; sm &= sm - 1u) {
        const uint32_t t = (uint32_t)__builtin_ctzll(sm);
        const uint32_t id = q.term_id[t];
        const uint32_t b0 = (uint32_t)seg.term_blk[id], b1 = (uint32_t)seg.term_blk[id + 1];
        const uint32_t c = wave_lower_bound(seg.blk_first, b0, b1, s_begin << SB, lane);
        const uint32_t e = (s_end >= n_stripes) ? b1 : wave_lower_bound(seg.blk_first, c, b1, s_end << SB, lane);
        if (TALLY) { cn_aux += e - c; }
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
        return rl64(tbase_reg, t);
    };
    /* the first kOrwRegSparse block-decoded terms get pipelined header registers */
    uint32_t sp_t[kOrwRegSparse];
    uint32_t n_sp = 0;
    {
        uint64_t sm = FLAT ? 0ull : sparse_mask;
#pragma unroll
        for (uint32_t u = 0; u < kOrwRegSparse; ++u) {
            sp_t[u] = 0;
            if (sm) { sp_t[u] = (uint32_t)__builtin_ctzll(sm); sm &= sm - 1u; n_sp = u + 1u; }
        }
    }
    uint64_t slow_sparse_mask = FLAT ? 0ull : sparse_mask;         /* block-decoded terms beyond the pipelined ones */
    for (uint32_t u = 0; u < n_sp; ++u) slow_sparse_mask &= slow_sparse_mask - 1u;

    uint32_t tkn = 0;                                              /* wave-uniform top-k state */
    bool theta_valid = false;
    uint64_t theta_w = 0;
    uint32_t theta_d = 0;
    uint64_t theta_glob = 0;                                       /* bit pattern; lower bound of the query's final k-th weight */
    unsigned long long matches = 0;                                /* per lane, reduced at the end */
    uint32_t n_scored = 0;                                         /* diagnostics */
    bool lh_dirty = false;
    const uint32_t n_local = empty ? 0u : s_end - s_begin;

    /* software-pipelined per-stripe registers: lane j = block j of the pipelined terms' runs; lane t =
     * container offset of dense term t */
    uint32_t hm[kOrwRegSparse], hf[kOrwRegSparse], hw[kOrwRegSparse], hn[kOrwRegSparse];
#pragma unroll
    for (uint32_t u = 0; u < kOrwRegSparse; ++u) { hm[u] = hf[u] = hw[u] = 0; hn[u] = 0xFFFFFFFFu; }
    /* container offsets of every dense term in the unit's stripes (one pass; later lookups — per stripe for the bitmaps, per queued
     * candidate for its wdf byte — are LDS reads) */
    for (uint64_t dm = dense_mask; dm; dm &= dm - 1u) {
        const uint32_t t = (uint32_t)__builtin_ctzll(dm);
        const uint32_t dr = rl32(dense_reg, t);
        if (TALLY) { cn_aux += n_local; }
        for (uint32_t i = lane; i < n_local; i += 64u) dir_tab[t * SPG + i] = seg.dense_dir[(size_t)dr * seg.n_stripes + (s_begin + i)];
    }
    wave_lds_fence();
    uint32_t hc_off = 0;
    auto issue_headers = [&](uint32_t x) {
#pragma unroll
        for (uint32_t u = 0; u < kOrwRegSparse; ++u) {
            if (u < n_sp) {
                const uint32_t rb = rs[sp_t[u] * SPG + x], nb = re[sp_t[u] * SPG + x] - rb;
                if (TALLY) { cn_hdr += XGM_SU(nb); }
                if (lane < nb) {
                    hm[u] = seg.blk_meta[rb + lane]; hf[u] = seg.blk_first[rb + lane]; hw[u] = seg.blk_word[rb + lane];
                    hn[u] = lane + 1u < nb ? seg.blk_first[rb + lane + 1u] : 0xFFFFFFFFu;
                }
            }
        }
        hc_off = (present_reg && dense_reg != kNoDense) ? dir_tab[lane * SPG + x] : 0u;
    };

    uint32_t stripe_base = 0;
    uint32_t hc_cur = 0;
    /* second pass (the guess was too high): the first pass's quantities, to recognise the documents it weighed */
    bool fix = false;
    uint64_t essA_mask = 0;
    uint32_t qA2_reg = 0, qA1_reg = 0;

    unsigned long long pc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};           /* diagnostics: cycles per section */
    unsigned long long tmark = (XGM_ORW_TIMERS && phase_cycles) ? __builtin_readcyclecounter() : 0ull;
#define ORW_PH(i) do { if (XGM_ORW_TIMERS && phase_cycles) { const unsigned long long n_ = __builtin_readcyclecounter(); pc[i] += n_ - tmark; tmark = n_; } } while (0)

    /* One round (64 candidates) of gathers, issued a round ahead of its use: the document length and,
     * for queries of <= 8 terms, the dense terms' wdf bytes straight from the probe containers (one
     * byte per candidate and term; they never pass through LDS). */
    const bool fast = T <= 8u;
    uint32_t pf_dl = 1u, pf_pb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto prefetch_round = [&](uint32_t i0, uint32_t n_c) {
        const uint32_t o = i0 + lane;
        const bool valid = o < n_c;
        const uint32_t cd = valid ? c_did[o] : 0u;
        const uint32_t slot = cd & (W - 1u), csl = valid ? (cd >> SB) - s_begin : 0u;      /* slot in its stripe, stripe of the unit */
        pf_dl = valid ? seg.doclen[cd] : 1u;
        const uint32_t n_valid = n_c - i0 < 64u ? n_c - i0 : 64u;
        const uint32_t sec = TALLY ? tally_sectors(valid, slot, 6u) : 0u;
        if (TALLY) { cn_dl += tally_sectors(valid, slot, 4u); cn_dl_raw += n_valid; }
#pragma unroll
        for (uint32_t t = 0; t < 8u; ++t) {
            pf_pb[t] = 0;
            if (fast && ((dense_mask >> t) & 1ull)) {
                const uint32_t oo = valid ? dir_tab[t * SPG + csl] : 0u;           /* the container of the candidate's own stripe */
                if (TALLY) { cn_probe += sec; cn_probe_raw += n_valid; }
                if (oo) pf_pb[t] = seg.dense_data[(size_t)oo * 16 + (size_t)NW * 4 + slot];
            }
        }
    };

    /* BM25 + tree sum + top-k for the n_c candidates of the chunk (round 0 already prefetched); clears
     * c_w behind itself */
    auto score_candidates = [&](uint32_t n_c) {
        n_scored += n_c;
        for (uint32_t i0 = 0; i0 < n_c; i0 += 64u) {
            ORW_PH(5);
            if (tkn + 64u > cap) {
                orw_topk_sort(tk_w, tk_d, tk_m, cap, lane);
                tkn = tkn < k ? tkn : k;
                /* (the same LDS word read by every lane is still a VECTOR value to the compiler: everything decided by the threshold — the stripe
                 *  loop's exits, the planes the bound sum touches — then runs under exec masks with copies at every join; through readfirstlane
                 *  it is the scalar it is: round 6) */
                if (tkn == k) { theta_valid = true; theta_w = rl64(tk_w[k - 1], 0u); theta_d = rl32(tk_d[k - 1], 0u); }
                for (uint32_t i = tkn + lane; i < cap; i += 64u) { tk_w[i] = 0; tk_d[i] = 0xFFFFFFFFu; tk_m[i] = 0; }
                wave_lds_fence();
            }
            ORW_PH(8);
            const uint32_t o = i0 + lane;
            const bool valid = o < n_c;
            const uint32_t did = valid ? c_did[o] : 0u;
            const uint32_t dlen = pf_dl;
            uint32_t pb[8];
#pragma unroll
            for (uint32_t t = 0; t < 8u; ++t) pb[t] = pf_pb[t];
            if (i0 + 64u < n_c) prefetch_round(i0 + 64u, n_c);       /* the next round's gathers fly while this one is weighed */
            /* BM25Weight::get_sumpart, bm25weight.cc:170-181 — same operations, same order */
            const double len = (double)dlen;
            double normlen = len * q.len_factor;
            normlen = normlen > q.min_normlen ? normlen : q.min_normlen;
            const double denom_len = q.k1 * (normlen * q.b + (1.0 - q.b));
            uint32_t subqs = 0;
            uint32_t sumA = 0;                                     /* second pass: the first pass's quantised bound sum of this document */
            double weight;
            if (T <= 8u) {
                /* leaves and tree in registers: the node program is wave-uniform (SGPRs), so an operand is an
                 * indexed register read (s_set_gpr_idx), not an LDS round trip */
                orw_d8 v;
#pragma unroll
                for (uint32_t g = 0; g < 2u; ++g) {
                    uint32_t ev[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) {
                        const uint32_t t = g * 4u + u;
                        ev[u] = ((dense_mask >> t) & 1ull) ? pb[t] : ((t < T && valid) ? (uint32_t)c_w[(size_t)t * kOrwCand + o] : 0u);
                    }
                    double wt[4] = {-0.0, -0.0, -0.0, -0.0};        /* absent leaf: x + (-0.0) == x */
                    if (g * 4u < T && __ballot((ev[0] | ev[1] | ev[2] | ev[3]) != 0u)) {   /* nobody has these terms: skip the divides */
#pragma unroll
                        for (uint32_t u = 0; u < 4u; ++u) {
                            const double wdf = (double)(ev[u] - 1u);
                            const double denom = denom_len + wdf;
                            const double x = q.termweight[g * 4u + u] * (wdf / denom);
                            wt[u] = ev[u] ? x : -0.0;
                            subqs += ev[u] ? 1u : 0u;
                            if (ev[u] && !((dense_mask >> (g * 4u + u)) & 1ull)) c_w[(size_t)(g * 4u + u) * kOrwCand + o] = 0;
                            if (fix && g * 4u + u < T) {
                                const uint32_t t = g * 4u + u;
                                const uint32_t a2_ = rl32(qA2_reg, t), a1_ = rl32(qA1_reg, t);      /* (read before the per-lane choice: no branches around readlanes) */
                                sumA += ev[u] ? (ev[u] >= 3u ? a2_ : a1_) : 0u;
                            }
                        }
                    }
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) v[g * 4u + u] = wt[u];
                }
                ORW_PH(9);
                /* OrPostList::get_weight: l + r up the tree (in place: node j lands in its left operand's slot) */
                for (uint32_t j = 0; j + 1u < T; ++j) {
                    const uint32_t a = (uint32_t)(prog_a >> (8u * j)) & 7u, b = (uint32_t)(prog_b >> (8u * j)) & 7u;
                    const double x = v[a] + v[b];
                    v[a] = x;
                }
                weight = v[prog_root & 7u];
            } else {
                for (uint32_t t0 = 0; t0 < T; t0 += 4u) {          /* four leaves at a time: independent divide chains */
                    uint32_t ev[4];
    #pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) ev[u] = (t0 + u < T && valid) ? (uint32_t)c_w[(size_t)(t0 + u) * kOrwCand + o] : 0u;
                    double wt[4] = {-0.0, -0.0, -0.0, -0.0};
                    if (__ballot((ev[0] | ev[1] | ev[2] | ev[3]) != 0u)) {
    #pragma unroll
                        for (uint32_t u = 0; u < 4u; ++u) {
                            const double wdf = (double)(ev[u] - 1u);
                            const double denom = denom_len + wdf;
                            const double x = q.termweight[(t0 + u) & (XGM_MAX_TERMS - 1u)] * (wdf / denom);
                            wt[u] = ev[u] ? x : -0.0;
                            subqs += ev[u] ? 1u : 0u;
                            if (fix && t0 + u < T) {
                                const uint32_t t = t0 + u;
                                const uint32_t a2_ = rl32(qA2_reg, t), a1_ = rl32(qA1_reg, t);      /* (read before the per-lane choice: no branches around readlanes) */
                                sumA += ev[u] ? (ev[u] >= 3u ? a2_ : a1_) : 0u;
                            }
                        }
                    }
    #pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) {
                        if (t0 + u < T) {
                            val[(t0 + u) * 64u + lane] = wt[u];
                            if (ev[u]) c_w[(size_t)(t0 + u) * kOrwCand + o] = 0;
                        }
                    }
                }
                ORW_PH(9);
                for (uint32_t j = 0; j + 1u < T; ++j) {
                    const uint32_t a = q.ip_a[j], b = q.ip_b[j];
                    val[a * 64u + lane] = val[a * 64u + lane] + val[b * 64u + lane];
                }
                weight = val[(uint32_t)q.ip_root * 64u + lane];
            }
            const uint64_t wb = (uint64_t)__double_as_longlong(weight);
            ORW_PH(10);
            /* weighed by the first pass: its bound sum reached the guess (the same quantised sum, document by document) */
            const bool in_first = fix && sumA >= kq;
            const bool live = valid && subqs != 0u && wb >= theta_glob && !in_first;
            if (prune && live) {
                int b = (int)(wb >> kHistShift) - hbase;
                b = b < 0 ? 0 : (b > (int)XGM_OR_HIST - 1 ? (int)XGM_OR_HIST - 1 : b);
                atomicAdd(&lh[b], 1u);
            }
            if (prune && __ballot(live)) lh_dirty = true;
            const bool take = live && (!theta_valid || cand_before(wb, did, theta_w, theta_d));
            const uint64_t tm = __ballot(take);
            if (take) { const uint32_t p = tkn + mbcnt(tm); tk_w[p] = wb; tk_d[p] = did; tk_m[p] = (uint8_t)subqs; }
            tkn += (uint32_t)__popcll(tm);
            ORW_PH(11);
        }
    };

    /* The candidate queue: with a threshold in force a stripe leaves a handful of candidates, far fewer than the 64 lanes of a round
     * of gathers + BM25.  They wait in c_did / c_w (block-decoded terms: their wdf, scattered while the stripe's blocks are at hand)
     * and are weighed together, in full rounds, once enough stripes have contributed. */
    uint32_t qn = 0;
    auto flush_queue = [&]() {
        if (qn == 0u) return;
        wave_lds_fence();
        prefetch_round(0u, qn);
        score_candidates(qn);
        wave_lds_fence();
        qn = 0;
    };

    /* quantised bounds for a threshold th (> 0): which terms are essential (MaxScore over all terms: only the A/B variant without the
     * bound sum uses it) and, per lane t, the quantised bound of term t at its largest wdf (q2) and at wdf = 1 (q1).
     * Quantised UP: q >= ub * kQ / th, so sum(q) >= kQ whenever sum(ub) >= th */
    /* (a threshold moves rarely — the histogram is looked at every fourth stripe and mostly says the same: the last answer is kept and the
     *  two fp64 divisions are paid only when the threshold's bits changed, round 5) */
    uint64_t qz_bits = 0, qz_ess = 0;
    uint32_t qz_q2 = 0, qz_q1 = 0;
    auto quantise = [&](uint64_t th_bits, uint64_t& ess, uint32_t& q2, uint32_t& q1) {
        if (th_bits == qz_bits) { ess = qz_ess; q2 = qz_q2; q1 = qz_q1; return; }
        const double th = __longlong_as_double((long long)th_bits);
        ess = __ballot(present_reg && !(prefix_reg < th));
        const double r2 = ub_reg * (double)kq / th, r1 = ub1_reg * (double)kq / th;
        q2 = r2 >= (double)kq ? kq : (uint32_t)r2 + 1u;
        q1 = r1 >= (double)kq ? kq : (uint32_t)r1 + 1u;
        qz_bits = th_bits; qz_ess = ess; qz_q2 = q2; qz_q1 = q1;
    };
    /* the histogram's bound of the final k-th weight: highest bucket with >= k documents at or above it */
    auto hist_bound = [&](const uint32_t* hc) {
        const uint32_t s4 = hc[0] + hc[1] + hc[2] + hc[3];
        const uint32_t P = wave_incl_scan(s4);
        const uint32_t suf = __builtin_amdgcn_readlane(P, 63) - P + s4;       /* documents in buckets >= 4 * lane */
        const uint64_t okm = __ballot(suf >= k);
        if (okm) {
            const uint32_t Lh = 63u - (uint32_t)__builtin_clzll(okm);
            const uint32_t cum = __builtin_amdgcn_readlane(suf, Lh) - __builtin_amdgcn_readlane(s4, Lh);
            uint32_t bsel = 4u * Lh;
            const uint32_t c3 = __builtin_amdgcn_readlane(hc[3], Lh), c2 = __builtin_amdgcn_readlane(hc[2], Lh), c1 = __builtin_amdgcn_readlane(hc[1], Lh);
            if (cum + c3 >= k) bsel = 4u * Lh + 3u;
            else if (cum + c3 + c2 >= k) bsel = 4u * Lh + 2u;
            else if (cum + c3 + c2 + c1 >= k) bsel = 4u * Lh + 1u;
            if (bsel > 0u) {
                const uint64_t tb = (uint64_t)((uint32_t)hbase + bsel) << kHistShift;
                theta_glob = tb > theta_glob ? tb : theta_glob;
            }
        }
    };

    for (uint32_t pass = 0; pass < 2u; ++pass) {
        if (pass == 1u) {
            /* was the guess too high?  Only if fewer than k documents of the whole query are known at or above it */
            if (!seed_bits || empty) break;
            if (lh_dirty) {
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) {
                    const uint32_t v = lh[lane * 4u + i];
                    if (v) { atomicAdd(&hist_g[lane * 4u + i], v); lh[lane * 4u + i] = 0; }
                }
                lh_dirty = false;
                wave_lds_fence();
            }
            uint32_t hc[4];
            if (TALLY) { cn_aux += XGM_OR_HIST; }
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) hc[i] = __hip_atomic_load(&hist_g[lane * 4u + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hist_bound(hc);
            const uint64_t th_now = theta_valid && theta_w > theta_glob ? theta_w : theta_glob;
            if (th_now >= seed_bits) break;
            fix = true;
            quantise(seed_bits, essA_mask, qA2_reg, qA1_reg);
            if (FLAT) {
                fl_cur = fl_start;
#pragma unroll
                for (uint32_t u = 0; u < kFlPre; ++u) flat_prefetch(u);
            }
        }

        auto next_active = [&](uint32_t from) {
            if (dense_mask || FLAT) return from < n_local ? from : n_local;    /* a dense term is (almost) everywhere (FLAT without one: every stripe is looked at — a few instructions where no term has a posting) */
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

        uint32_t sl = next_active(0);
        if (sl < n_local) issue_headers(sl);
        while (sl < n_local) {
            ORW_PH(7);
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

            /* ---- global threshold: highest histogram bucket with >= k documents at or above it ---- */
            uint32_t hc[4] = {0, 0, 0, 0};
            /* once a threshold is known it moves slowly: the query-wide histogram is read (and, below, fed) every fourth stripe only */
            const bool have_th = seed_bits || theta_valid || theta_glob;
            const bool look = prune && (fix || !have_th || (sl & 3u) == 0u);
            if (look) {
                if (TALLY) { cn_aux += XGM_OR_HIST; }
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) hc[i] = __hip_atomic_load(&hist_g[lane * 4u + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }

            /* ---- 1a. dense terms: union of the containers' bitmaps (4 words per lane) and, once a
             * threshold is known, the bit-sliced sum of the present terms' quantised weight bounds:
             * plane j of S holds bit j of min(sum, kq - 1) for 32 documents, ovf = the sum reached kq (pl_q planes in use). ---- */
            uint32_t a[4] = {0, 0, 0, 0}, e[4] = {0, 0, 0, 0};
            uint32_t S[PL][4], ovf[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t j = 0; j < PL; ++j) { S[j][0] = S[j][1] = S[j][2] = S[j][3] = 0; }
            /* S += qv on the documents of B.  Planes outermost: the addend of plane j is chosen once (wave-uniform bit j of qv) for the
             * lane's four words; planes the running maximum of the sum cannot reach are not touched (max_sum: the largest value any
             * document's sum can have so far — terms are added in ascending order of their bounds, so the early ones stay low) */
            uint32_t max_sum = 0;
            auto planes_for = [&](uint32_t add) {
                const uint32_t nm = rfl32(max_sum + add);
                max_sum = nm < kq ? nm : kq;
                return nm >= kq ? pl_q : 32u - (uint32_t)__builtin_clz(nm | 1u);
            };
            auto add_bound = [&](uint32_t qv, const uint32_t* B) {
                if (qv >= kq) {
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i) ovf[i] |= B[i];
                    return;
                }
                const uint32_t np = planes_for(qv);
                uint32_t carry[4] = {0, 0, 0, 0};
#pragma unroll
                for (uint32_t j = 0; j < PL; ++j) {
                    if (j < np) {
                        if ((qv >> j) & 1u) {
#pragma unroll
                            for (uint32_t i = 0; i < 4u; ++i) {
                                const uint32_t sj = S[j][i], xo = sj ^ B[i];
                                S[j][i] = xo ^ carry[i];
                                carry[i] = orw_bfi(xo, carry[i], sj);  /* majority(sj, B, carry): the carry of a full adder */
                            }
                        } else {
#pragma unroll
                            for (uint32_t i = 0; i < 4u; ++i) {
                                const uint32_t sj = S[j][i];
                                S[j][i] = sj ^ carry[i];
                                carry[i] &= sj;
                            }
                        }
                    }
                }
                if (np == pl_q) {
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i) ovf[i] |= carry[i];
                }
            };
            /* documents of B get q1v, those also in P (wdf >= 2) get q2v >= q1v: one ripple pass, the addend of plane j chosen by the
             * (wave-uniform) bits j of the two values */
            auto add_bound2 = [&](uint32_t q1v, uint32_t q2v, const uint32_t* B, const uint32_t* P) {
                if (q1v >= kq || q1v == q2v) { add_bound(q1v, B); return; }
                uint32_t lo[4];
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) lo[i] = B[i] & ~P[i];
                if (q2v >= kq) {
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i) ovf[i] |= P[i];
                    add_bound(q1v, lo);
                    return;
                }
                const uint32_t np = planes_for(q2v);
                uint32_t carry[4] = {0, 0, 0, 0};
#pragma unroll
                for (uint32_t j = 0; j < PL; ++j) {
                    if (j < np) {
                        const uint32_t b1 = (q1v >> j) & 1u, b2 = (q2v >> j) & 1u;
                        if (b1 | b2) {
#pragma unroll
                            for (uint32_t i = 0; i < 4u; ++i) {
                                const uint32_t ad = (b1 & b2) ? B[i] : (b1 ? lo[i] : P[i]);
                                const uint32_t sj = S[j][i], xo = sj ^ ad;
                                S[j][i] = xo ^ carry[i];
                                carry[i] = orw_bfi(xo, carry[i], sj);
                            }
                        } else {
#pragma unroll
                            for (uint32_t i = 0; i < 4u; ++i) {
                                const uint32_t sj = S[j][i];
                                S[j][i] = sj ^ carry[i];
                                carry[i] &= sj;
                            }
                        }
                    }
                }
                if (np == pl_q) {
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i) ovf[i] |= carry[i];
                }
            };
            uint64_t ess_mask = present_mask;                       /* block-decoded terms whose documents are all candidates */
            bool use_sum = false;                                   /* candidates of the dense terms come from the bound sum */
            uint32_t q2_reg = kq, q1_reg = kq;                      /* lane t: quantised bounds of term t */
            bool first_group = true;
            bool stop = false;
            for (uint32_t dp = 0; dp < n_dense_ord || first_group;) {
                uint32_t tt[XGM_ORW_GROUP], oo[XGM_ORW_GROUP];
                uint32_t x[XGM_ORW_GROUP][4], pl[XGM_ORW_GROUP][4];
                /* the wdf >= 2 bitmaps are wanted once there is a threshold; the very first group is requested before this
                 * stripe's threshold is known: decide by the last one (a threshold never falls) */
                const bool want_planes = prune && !no_sum && seg.dense_plane != 0u && (fix || seed_bits || theta_valid || theta_glob);
#pragma unroll
                for (uint32_t u = 0; u < XGM_ORW_GROUP; ++u) {
                    tt[u] = 0; oo[u] = 0;
                    if (dp < n_dense_ord) { tt[u] = (uint32_t)(dense_ord >> (4u * dp)) & 15u; ++dp; oo[u] = rl32(hc_cur, tt[u]); }
                    if (TALLY) { if (oo[u]) cn_bmpw += want_planes ? 2u * NW : NW; }
                    /* ONE 16-byte load per lane and bitmap, issued without a branch around it (NW is a multiple of 4: XGM_MIN_STRIPE_BITS; containers
                     * start on 16-byte boundaries and so do their planes) — written word by word under `w < NW` the compiler cannot merge the four
                     * loads and wraps each in an exec-mask region: 40 load instructions per stripe where 10 do (round 6) */
                    const bool lane_in = lane * 4u < NW;
                    /* (no container in this stripe — or no containers at all: dense_data is null then —: the load still goes out, to 16 bytes that
                     *  are always there (the query array), and its result is dropped.  A branch around each load measured 1.75 ms per launch of C3
                     *  against 1.50 with the ten loads back to back) */
                    const bool has = oo[u] != 0u;
                    const unsigned char* safe = reinterpret_cast<const unsigned char*>(queries);
                    const unsigned char* cp = has ? seg.dense_data + (size_t)oo[u] * 16 : safe;
                    const unsigned char* pp = has ? cp + seg.dense_plane : safe;
                    const uint32_t li = (has && lane_in) ? lane : 0u;
                    orw_u4 xv = reinterpret_cast<const orw_u4*>(cp)[li];
                    orw_u4 pv_ = orw_u4{0u, 0u, 0u, 0u};
                    if (want_planes) pv_ = reinterpret_cast<const orw_u4*>(pp)[li];
                    if (!(has && lane_in)) { xv = orw_u4{0u, 0u, 0u, 0u}; pv_ = orw_u4{0u, 0u, 0u, 0u}; }
                    x[u][0] = xv.x; x[u][1] = xv.y; x[u][2] = xv.z; x[u][3] = xv.w;
                    pl[u][0] = pv_.x; pl[u][1] = pv_.y; pl[u][2] = pv_.z; pl[u][3] = pv_.w;
                }
                if (first_group) {
                    first_group = false;
                    /* the histogram loads were issued before the bitmaps': consume them while those fly */
                    if (prune) {
                        if (look) hist_bound(hc);
                        uint64_t th_bits = theta_valid && theta_w > theta_glob ? theta_w : theta_glob;
                        if (fix && th_bits >= seed_bits) { stop = true; break; }      /* whatever is left was weighed by the first pass or cannot reach the top k */
                        if (!fix && seed_bits > th_bits) th_bits = seed_bits;
                        if (th_bits) {
                            quantise(th_bits, ess_mask, q2_reg, q1_reg);
                            use_sum = !no_sum;
                        }
                    }
                }
                ORW_PH(12);
                if (XGM_ORW_TIMERS && phase_cycles) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ORW_PH(14); }      /* (diagnostics: the wait for the bitmaps apart from the sum) */
#pragma unroll
                for (uint32_t u = 0; u < XGM_ORW_GROUP; ++u) {
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i) a[i] |= x[u][i];
                    if (use_sum && oo[u]) {
                        const uint32_t qa = __builtin_amdgcn_readlane(q1_reg, tt[u]), qb = __builtin_amdgcn_readlane(q2_reg, tt[u]);
                        if (want_planes) add_bound2(qa, qb, x[u], pl[u]);
                        else add_bound(qb, x[u]);                  /* no wdf >= 2 bitmap at hand: every document at the term's maximum */
                    }
                }
            }
            if (stop) break;
            ORW_PH(13);
            uint32_t es[4] = {0, 0, 0, 0};                         /* bound sum switched off (A/B): union of the essential block-decoded terms */
            if (sparse_mask) {
                /* ---- 1b. block-decoded terms, one at a time: every block of the stripe is decoded into two LDS bitmaps — the term's
                 * documents, and those whose wdf is >= 2 — which then join the bound sum exactly like a container's pair ---- */
                for (uint64_t sm = sparse_mask; sm; sm &= sm - 1u) {
                    const uint32_t t = (uint32_t)__builtin_ctzll(sm);
                    /* (the LDS bitmaps 16 bytes per lane at a time: their slices start on 16-byte boundaries, NW is a multiple of 4) */
                    if (lane * 4u < NW) {
                        reinterpret_cast<orw_u4*>(bm_ess)[lane] = orw_u4{0u, 0u, 0u, 0u};
                        reinterpret_cast<orw_u4*>(bm_ne)[lane] = orw_u4{0u, 0u, 0u, 0u};
                    }
                    wave_lds_fence();
                    if (FLAT) {
                        /* this stripe's postings of the term, 64 per coalesced load, straight into the bitmap pair */
                        uint32_t c = rl32(fl_cur, t);
                        const uint32_t e_ = rl32(fl_end, t), stripe_end = stripe_base + W, c0 = c;
                        /* the 64 postings at the cursor were requested when the cursor last moved */
                        int pf = -1;
                        uint32_t d0 = 0xFFFFFFFFu, w0 = 0u;
#pragma unroll
                        for (uint32_t u = 0; u < kFlPre; ++u) if (pre_t[u] == t) { pf = (int)u; d0 = pre_d[u]; w0 = pre_w[u]; }
                        bool first_round = pf >= 0;
                        while (true) {
                            const uint32_t g = c + lane;
                            const bool in_arr = g < e_;
                            uint32_t d, wf;
                            if (first_round) { d = d0; wf = w0; first_round = false; }
                            else {
                                d = in_arr ? seg.flat_did[g] : 0xFFFFFFFFu;
                                wf = in_arr ? (uint32_t)seg.flat_wdf[g] : 0u;
                            }
                            if (TALLY) { cn_blkw += 80u; }                            /* (64 docids + 64 wdf bytes, in 4-byte words) */
                            const bool in = in_arr && d < stripe_end;                 /* (the cursor sits at the stripe's first posting) */
                            if (in) {
                                const uint32_t s_ = d - stripe_base;
                                atomicOr(&bm_ess[s_ >> 5], 1u << (s_ & 31u));
                                if (wf >= 2u) atomicOr(&bm_ne[s_ >> 5], 1u << (s_ & 31u));
                            }
                            const uint32_t n_in = (uint32_t)__popcll(__ballot(in));
                            c += n_in;
                            if (n_in < 64u) break;
                        }
                        if (lane == t) { fl_cur = c; fl_c0 = c0; fl_n = c - c0; }
#pragma unroll
                        for (uint32_t u = 0; u < kFlPre; ++u) if (pf == (int)u) flat_prefetch(u);      /* the next stripe's */
                        ORW_PH(15);
                    }
                    const uint32_t rb0 = FLAT ? 0u : rs[t * SPG + sl], nb = FLAT ? 0u : re[t * SPG + sl] - rb0;
                    /* the pipelined terms' headers are already in registers (lane j = block j of the run) */
                    int pu = -1;
#pragma unroll
                    for (uint32_t u = 0; u < kOrwRegSparse; ++u) if (u < n_sp && sp_t[u] == t) pu = (int)u;
                    for (uint32_t j0 = 0; j0 < nb; j0 += 2u) {
                        uint32_t bmeta[2] = {0, 0}, bfirst[2] = {0, 0};
                        Words4 pv[2] = {Words4{0, 0, 0, 0}, Words4{0, 0, 0, 0}};
#pragma unroll
                        for (uint32_t v = 0; v < 2u; ++v) {
                            const uint32_t j = j0 + v;
                            if (j < nb) {
                                uint32_t bword;
                                if (pu >= 0 && j < 64u) {
                                    bmeta[v] = rl32(pu == 0 ? cm[0] : cm[kOrwRegSparse - 1u], j);
                                    bfirst[v] = rl32(pu == 0 ? cf[0] : cf[kOrwRegSparse - 1u], j);
                                    bword = rl32(pu == 0 ? cw[0] : cw[kOrwRegSparse - 1u], j);
                                } else {
                                    bmeta[v] = seg.blk_meta[rb0 + j]; bfirst[v] = seg.blk_first[rb0 + j]; bword = seg.blk_word[rb0 + j];
                                    if (TALLY) { cn_hdr += 1u; }
                                }
                                if (TALLY) { cn_blkw += XGM_SU(payload_words(bmeta[v])) - 2u; }
                                const uint64_t tbt = tbase(t);                                      /* read by every lane, not under the per-lane test */
                                if (lane * 4u < payload_words(bmeta[v])) pv[v] = *reinterpret_cast<const Words4*>(seg.words + tbt + bword + lane * 4u);
                            }
                        }
#pragma unroll
                        for (uint32_t v = 0; v < 2u; ++v)
                            if (j0 + v < nb) orw_block<TabT, false>(pv[v], bmeta[v], bfirst[v], stage, lane, stripe_base, bm_ess, bm_ne, nullptr, rankw, c_w, 0u, 0u);
                    }
                    wave_lds_fence();
                    uint32_t xb[4] = {0, 0, 0, 0}, xp[4] = {0, 0, 0, 0};
                    if (lane * 4u < NW) {
                        const orw_u4 vb = reinterpret_cast<const orw_u4*>(bm_ess)[lane], vp = reinterpret_cast<const orw_u4*>(bm_ne)[lane];
                        xb[0] = vb.x; xb[1] = vb.y; xb[2] = vb.z; xb[3] = vb.w;
                        xp[0] = vp.x; xp[1] = vp.y; xp[2] = vp.z; xp[3] = vp.w;
                    }
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i) a[i] |= xb[i];
                    if (use_sum) add_bound2(rl32(q1_reg, t), rl32(q2_reg, t), xb, xp);
                    else if ((ess_mask >> t) & 1ull) {
#pragma unroll
                        for (uint32_t i = 0; i < 4u; ++i) es[i] |= xb[i];
                    }
                    wave_lds_fence();
                }
            }
            /* ---- candidates ---- */
            if (use_sum) {
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) e[i] = ovf[i];
            } else if (prune && ess_mask != present_mask) {
                /* the bound sum is switched off (A/B): every document of an essential term */
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) e[i] = es[i];
                for (uint64_t dm = dense_mask & ess_mask; dm; dm &= dm - 1u) {
                    const uint32_t oo = __builtin_amdgcn_readlane(hc_cur, (uint32_t)__builtin_ctzll(dm));
                    if (TALLY) { if (oo) cn_bmpw += NW; }
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i) {
                        const uint32_t w = lane * 4u + i;
                        if (oo && w < NW) e[i] |= reinterpret_cast<const uint32_t*>(seg.dense_data + (size_t)oo * 16)[w];
                    }
                }
            } else {
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) e[i] = a[i];
            }
            /* the scatter of the block-decoded terms looks candidates up in bm_ess */
            if (sparse_mask && lane * 4u < NW) reinterpret_cast<orw_u4*>(bm_ess)[lane] = orw_u4{e[0], e[1], e[2], e[3]};
            ORW_PH(1);

            /* ---- exact match count (first pass); candidates in docid order ---- */
            if (!fix) matches += (unsigned long long)(__popc(a[0]) + __popc(a[1]) + __popc(a[2]) + __popc(a[3]));
            const uint32_t cnt = (uint32_t)(__popc(e[0]) + __popc(e[1]) + __popc(e[2]) + __popc(e[3]));
            const uint32_t incl = wave_incl_scan(cnt);
            const uint32_t n_total = __builtin_amdgcn_readlane(incl, 63);
            if (TALLY) {
                if (fix) cn_fixw += n_total; else if (sl == 0u) cn_first += n_total;
            }
            if (n_total == 0u) {
                if (sl_next < n_local) issue_headers(sl_next);
                sl = sl_next;
                continue;
            }
            /* chunks of consecutive lanes holding <= kOrwCand candidates (one lane owns <= 128) */
            for (uint32_t lane_lo = 0; lane_lo < 64u;) {
                const uint32_t ord_base = lane_lo ? __builtin_amdgcn_readlane(incl, lane_lo - 1u) : 0u;
                if (ord_base == n_total) break;
                const uint64_t fit = __ballot(incl - ord_base <= kOrwCand);          /* incl is monotone: bits up to the last lane that fits */
                const uint32_t lane_hi = 64u - (uint32_t)__builtin_clzll(fit);
                const uint32_t hi_incl = __builtin_amdgcn_readlane(incl, lane_hi - 1u);
                const uint32_t n_c = hi_incl - ord_base;
                const bool last_chunk = hi_incl == n_total;
                const bool in_chunk = lane >= lane_lo && lane < lane_hi;
                const uint32_t wlo = lane_lo * 4u, whi = lane_hi * 4u;
                lane_lo = lane_hi;
                if (qn + n_c > kOrwCand) flush_queue();
                const uint32_t qb = qn;                                /* the chunk's candidates go behind the queued ones */
                /* ---- 2a. enumerate the chunk's candidates in docid order ---- */
                if (in_chunk) {
                    uint32_t o = qb + incl - cnt - ord_base;
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i) {
                        const uint32_t w = lane * 4u + i;
                        if (sparse_mask && w < NW) rankw[w] = (uint16_t)o;
                        uint32_t m = e[i];
                        while (m) {
                            const uint32_t bit = (uint32_t)__ffs(m) - 1u;
                            c_did[o] = stripe_base + w * 32u + bit;
                            m &= m - 1u;
                            ++o;
                        }
                    }
                }
                const unsigned long long coarse = __ballot(in_chunk && cnt != 0u);   /* bit = 128-slot bucket with a candidate */
                wave_lds_fence();
                ORW_PH(2);

                /* ---- 2b. wdf of the dense terms: one byte per candidate and term, two rounds in flight ---- */
                for (uint32_t c0 = 0; !fast && c0 < n_c; c0 += 128u) {       /* (> 8 terms: through c_w) */
                    const uint32_t o0 = qb + c0 + lane, o1 = o0 + 64u;
                    const bool v0 = o0 < qb + n_c, v1 = o1 < qb + n_c;
                    const uint32_t slot0 = v0 ? c_did[o0] & (W - 1u) : 0u, slot1 = v1 ? c_did[o1] & (W - 1u) : 0u;
                    for (uint64_t dm = dense_mask; dm;) {
                        uint32_t tt[4], wv0[4], wv1[4];
#pragma unroll
                        for (uint32_t u = 0; u < 4u; ++u) {
                            tt[u] = 0; wv0[u] = 0; wv1[u] = 0;
                            if (dm) {
                                tt[u] = (uint32_t)__builtin_ctzll(dm); dm &= dm - 1u;
                                const uint32_t oo = __builtin_amdgcn_readlane(hc_cur, tt[u]);
                                if (oo) {
                                    if (TALLY) { cn_probe += tally_sectors(v0, slot0, 6u) + tally_sectors(v1, slot1, 6u); cn_probe_raw += n_c - c0 < 128u ? n_c - c0 : 128u; }
                                    const unsigned char* wb = seg.dense_data + (size_t)oo * 16 + (size_t)NW * 4;
                                    if (v0) wv0[u] = wb[slot0];
                                    if (v1) wv1[u] = wb[slot1];
                                }
                            }
                        }
#pragma unroll
                        for (uint32_t u = 0; u < 4u; ++u) {
                            if (wv0[u]) c_w[(size_t)tt[u] * kOrwCand + o0] = (TabT)wv0[u];
                            if (wv1[u]) c_w[(size_t)tt[u] * kOrwCand + o1] = (TabT)wv1[u];
                        }
                    }
                }
                ORW_PH(3);

                /* ---- 2c. wdf of the block-decoded terms: only blocks whose buckets hold a candidate ---- */
                auto bucket_need = [&](uint32_t first, uint32_t nfirst) {
                    const uint32_t lo = (first - stripe_base) >> 7;
                    const uint32_t hi = ((nfirst == 0xFFFFFFFFu ? W : nfirst - stripe_base) - 1u) >> 7;
                    const unsigned long long mm = (hi >= 63u ? ~0ull : ((1ull << (hi + 1u)) - 1ull)) & ~((1ull << lo) - 1ull);
                    return (coarse & mm) != 0ull;
                };
                if (FLAT) {
                    /* wdf of the flat terms for this chunk's candidates: the stripe's slice of the array once more (an L2 hit), scattered to
                     * the candidates' ordinals through the candidate bitmap and rankw — no decode */
                    /* (the first kFlPre terms' slices are requested together, before any is scattered) */
                    uint32_t sd[kFlPre], sw[kFlPre];
#pragma unroll
                    for (uint32_t u = 0; u < kFlPre; ++u) {
                        const uint32_t t = pre_t[u] & 63u;
                        const uint32_t c0 = rl32(fl_c0, t), nn = rl32(fl_n, t);
                        const bool v = u < n_pre && lane < nn;
                        sd[u] = v ? seg.flat_did[c0 + lane] : 0u;
                        sw[u] = v ? (uint32_t)seg.flat_wdf[c0 + lane] : 0u;
                    }
                    for (uint64_t sm = sparse_mask; sm; sm &= sm - 1u) {
                        const uint32_t t = (uint32_t)__builtin_ctzll(sm);
                        const uint32_t c0 = rl32(fl_c0, t), nn = rl32(fl_n, t);
                        TabT* row = c_w + (size_t)t * kOrwCand;
                        bool pre = false;
                        uint32_t d0 = 0u, w0 = 0u;
#pragma unroll
                        for (uint32_t u = 0; u < kFlPre; ++u) if (pre_t[u] == t) { pre = true; d0 = sd[u]; w0 = sw[u]; }
                        for (uint32_t done = 0; done < nn; done += 64u) {
                            const bool v = done + lane < nn;
                            uint32_t d, wf;
                            if (pre && done == 0u) { d = d0; wf = w0; }
                            else {
                                d = v ? seg.flat_did[c0 + done + lane] : 0u;
                                wf = v ? (uint32_t)seg.flat_wdf[c0 + done + lane] : 0u;
                            }
                            if (v) {
                                const uint32_t s_ = d - stripe_base, wd = s_ >> 5, bit = s_ & 31u;
                                if (wd >= wlo && wd < whi) {
                                    const uint32_t bm = bm_ess[wd];
                                    if ((bm >> bit) & 1u) row[(uint32_t)rankw[wd] + (uint32_t)__popc(bm & ((1u << bit) - 1u))] = (TabT)(wf + 1u);
                                }
                            }
                        }
                    }
                } else if (sparse_mask) {
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
                                const uint32_t bwd = __builtin_amdgcn_readlane(cw[u], jj[u]);      /* (read by every lane: not under the per-lane test below) */
                                const uint64_t tbu = tbase(sp_t[u]);
                                if (TALLY) { cn_blkw += payload_words(bm) - 2u; }
                                if (lane * 4u < payload_words(bm))
                                    pv[u] = *reinterpret_cast<const Words4*>(seg.words + tbu + bwd + lane * 4u);
                            }
                        }
#pragma unroll
                        for (uint32_t u = 0; u < kOrwRegSparse; ++u) {
                            if (have[u])
                                orw_block<TabT, true>(pv[u], __builtin_amdgcn_readlane(cm[u], jj[u]), __builtin_amdgcn_readlane(cf[u], jj[u]), stage, lane,
                                                      stripe_base, bm_ess, nullptr, nullptr, rankw, c_w + (size_t)sp_t[u] * kOrwCand, wlo, whi);
                        }
                    }
                    for (uint64_t sm = slow_sparse_mask; sm; sm &= sm - 1u) {
                        const uint32_t t = (uint32_t)__builtin_ctzll(sm);
                        const uint32_t rb0 = rs[t * SPG + sl], nb = re[t * SPG + sl] - rb0;
                        for (uint32_t j = 0; j < nb; ++j) {
                            const uint32_t meta = seg.blk_meta[rb0 + j], first = seg.blk_first[rb0 + j];
                            const uint32_t nfirst = j + 1u < nb ? seg.blk_first[rb0 + j + 1u] : 0xFFFFFFFFu;
                            if (TALLY) { cn_hdr += 1u; }
                            if (!bucket_need(first, nfirst)) continue;
                            if (TALLY) { cn_blkw += XGM_SU(payload_words(meta)) - 2u; }
                            Words4 pv = Words4{0, 0, 0, 0};
                            const uint64_t tbt = tbase(t);
                            if (lane * 4u < payload_words(meta)) pv = *reinterpret_cast<const Words4*>(seg.words + tbt + seg.blk_word[rb0 + j] + lane * 4u);
                            orw_block<TabT, true>(pv, meta, first, stage, lane, stripe_base, bm_ess, nullptr, nullptr, rankw, c_w + (size_t)t * kOrwCand, wlo, whi);
                        }
                    }
                }
                wave_lds_fence();
                ORW_PH(4);

                /* headers of the next active stripe: in flight while this chunk is scored */
                if (last_chunk && sl_next < n_local) issue_headers(sl_next);

                /* ---- 3. BM25, tree sum, top-k: once two rounds' worth wait (or a stripe came without a threshold) ---- */
                qn = qb + n_c;
                if (qn >= 128u || !use_sum) flush_queue();
                ORW_PH(5);
            }
            /* publish the histogram counts */
            if (lh_dirty && (fix || !have_th || (sl & 3u) == 2u || sl_next >= n_local)) {
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) {
                    const uint32_t v = lh[lane * 4u + i];
                    if (v) { atomicAdd(&hist_g[lane * 4u + i], v); lh[lane * 4u + i] = 0; }
                }
                lh_dirty = false;
                wave_lds_fence();
            }
            ORW_PH(6);
            sl = sl_next;
        }
        /* the pass is over: weigh what still waits, publish the counts (the units of the query learn from them) */
        flush_queue();
        if (lh_dirty) {
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) {
                const uint32_t v = lh[lane * 4u + i];
                if (v) { atomicAdd(&hist_g[lane * 4u + i], v); lh[lane * 4u + i] = 0; }
            }
            lh_dirty = false;
            wave_lds_fence();
        }
    }
    ORW_PH(7);
    if (XGM_ORW_TIMERS && phase_cycles && lane == 0) { for (int i = 0; i < 16; ++i) atomicAdd(&phase_cycles[i], pc[i]); }
#undef ORW_PH

    /* ---- unit epilogue ---- */
    orw_topk_sort(tk_w, tk_d, tk_m, cap, lane);
    for (int sh = 32; sh > 0; sh >>= 1) matches += (unsigned long long)__shfl_xor((long long)matches, sh);
    const uint32_t n_out = tkn < k ? tkn : k;
    xgm_cand* out = cand_out + (size_t)wk.slot * k_stride;
    const bool through = fuse != nullptr;                          /* the launch finishes its queries itself (xgm_unit_finish.h; round 6 — the conjunction kernel since round 3) */
    for (uint32_t i = lane; i < n_out; i += 64u) {
        xgm_cand c;
        c.wbits = tk_w[i]; c.did = tk_d[i]; c.subqs = tk_m[i];
        xgm_store_cand(through, &out[i], c);
    }
    if (lane == 0) {
        xgm_group_hdr h;
        h.matches = matches; h.n_cand = n_out; h.pad = n_scored;
        h.t_start = t_unit_start; h.t_end = __builtin_readcyclecounter();
        h.c_pos = 0; h.c_bmp_words = cn_bmpw; h.c_probes = cn_probe; h.c_blk_words = cn_blkw; h.c_hdrs = cn_hdr;
        h.c_doclen = cn_dl; h.c_aux_words = cn_aux; h.c_probes_raw = cn_probe_raw; h.c_doclen_raw = cn_dl_raw; h.c_pad[0] = (cn_fixw & 0x7FFFFFFFu) | (fix ? 0x80000000u : 0u); h.c_pad[1] = cn_first;
        xgm_store_hdr(through, &ghdr_out[wk.slot], h);
    }
    /* the query's last unit to arrive merges the units' lists into the final hits (the merge launch and the gaps around it cost 137 us per batch of C3) */
    if (through) xgm_unit_arrive<true>(*fuse, wk.qi, k, 0u, tk_w, tk_d, tk_m, cap, cand_out, ghdr_out, k_stride, lane, [&]() { orw_topk_sort(tk_w, tk_d, tk_m, cap, lane); });
#undef XGM_SU
}


/* ================================================================================================================================
 * xgm_orw2_kernel (round 5): the disjunction kernel written again for the shape that is nearly all of C3 — <= 8 terms, every one
 * with a probe container or (<= 2 of them) a flat posting array, one-byte wdf, k <= 192 — and for OCCUPANCY.  xgm_orw_kernel holds
 * ~250 VGPRs and 15 KB of LDS per wave: two waves per SIMD, and its stripe loop is a chain of dependent round trips (rounds 2-4:
 * 0.22 of HBM peak, issue slots half empty; forcing three waves onto it spills ~90 registers and loses).  What held the registers:
 * a lane owned FOUR words of every bitmap (5 terms x 2 planes x 4 words in flight, a 6-plane bit-sliced sum x 4 words kept across the
 * block decode of the terms without containers), the block headers of two such terms software-pipelined in registers, the decode's
 * own temporaries.  Here:
 *   - a stripe is taken in word-major PASSES: pass i = words i * 64 + lane.  A lane holds ONE word per term and plane, the bound sum's
 *     six planes are transient per pass (S0..S5: six registers, not twenty-four), candidates come out in docid order pass by pass;
 *   - terms without containers are read from their FLAT arrays (round 4: docids + one wdf byte, decoded once when the index was
 *     opened): a cursor per term, 64 postings per coalesced load, bits set in a per-term LDS bitmap pair that then joins the passes
 *     like a container's pair; the candidates' wdf of such a term is scattered from the same registers — no block headers, no payload,
 *     no unpack, no second decode;
 *   - LDS per wave: 6 KB (all terms with containers) or 12 KB: four resp. three waves per SIMD.
 * Semantics — exact match count, safe pruning by quantised bound sums against the planner's guess and the query-wide histogram, the
 * repair pass when the guess was too high, BM25 in the reference's operation order, top-k — are xgm_orw_kernel's, line for line where
 * the structure allows; both kernels are checked against the oracle by the same tests (XGM_NO_ORW2=1 selects the old one). */
#ifndef XGM_ORW2_WAVES_DENSE
#define XGM_ORW2_WAVES_DENSE 3      /* waves per SIMD the all-container instantiation is compiled for (A/B: tools/ab_build.sh) */
#endif
#ifndef XGM_ORW2_WAVES_SPARSE
#define XGM_ORW2_WAVES_SPARSE 3
#endif
constexpr uint32_t kO2Cand = 256u;          /* candidates the queue holds */
constexpr uint32_t kO2MaxT = 8u;
constexpr uint32_t kO2Sparse = 2u;          /* terms read from flat arrays (LDS bitmap pairs) */

__host__ __device__ inline size_t orw2_wave_bytes(uint32_t W, uint32_t T, uint32_t cap, uint32_t spg, bool sparse) {
    size_t off = 0;
    off += (size_t)cap * 8;                                    /* tk_w */
    off += (size_t)cap * 4;                                    /* tk_d */
    off += (size_t)XGM_OR_HIST * 4;                            /* lh */
    off += (size_t)T * spg * 4;                                /* dir_tab */
    off += (size_t)kO2Cand * 4;                                /* c_did */
    if (sparse) {
        off += (size_t)kO2Sparse * (W / 32u) * 4 * 2;          /* bmS: per flat term its documents / those with wdf >= 2 */
        off += (size_t)(W / 32u) * 4;                          /* bm_e: the stripe's candidates (the scatter looks them up) */
        off += (size_t)(W / 32u) * 2;                          /* rankw (u16) */
        off += (size_t)kO2Sparse * kO2Cand;                    /* c_w (u8): wdf + 1 of the queued candidates per flat term */
    }
    off += (size_t)cap;                                        /* tk_m (u8) */
    return (off + 15) & ~(size_t)15;
}

template <bool SPARSE, bool TALLY>
__global__ __launch_bounds__(XGM_WG, SPARSE ? XGM_ORW2_WAVES_SPARSE : XGM_ORW2_WAVES_DENSE) void xgm_orw2_kernel(xgm_seg_dev seg, const xgm_dev_query* __restrict__ queries,
                                                          const xgm_work* __restrict__ work, uint32_t n_work, uint32_t SPG,
                                                          uint32_t tab_terms, uint32_t cap, uint32_t k_stride,
                                                          uint32_t* __restrict__ hist_all,
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
    uint32_t cn_bmpw = 0, cn_probe_raw = 0, cn_dl_raw = 0, cn_aux = 0, cn_flat = 0;      /* traffic tallies (wave-uniform) */
    const uint32_t* ipa32 = reinterpret_cast<const uint32_t*>(q.ip_a);
    const uint32_t* ipb32 = reinterpret_cast<const uint32_t*>(q.ip_b);
    const uint64_t prog_a = ((uint64_t)rfl32(ipa32[1]) << 32) | rfl32(ipa32[0]);
    const uint64_t prog_b = ((uint64_t)rfl32(ipb32[1]) << 32) | rfl32(ipb32[0]);
    const uint32_t prog_root = __builtin_amdgcn_readfirstlane(q.ip_root);

    unsigned char* base = smem + (size_t)wave * orw2_wave_bytes(W, tab_terms, cap, SPG, SPARSE);
    size_t off = 0;
    uint64_t* tk_w = reinterpret_cast<uint64_t*>(base + off); off += (size_t)cap * 8;
    uint32_t* tk_d = reinterpret_cast<uint32_t*>(base + off); off += (size_t)cap * 4;
    uint32_t* lh = reinterpret_cast<uint32_t*>(base + off); off += (size_t)XGM_OR_HIST * 4;
    uint32_t* dir_tab = reinterpret_cast<uint32_t*>(base + off); off += (size_t)tab_terms * SPG * 4;
    uint32_t* c_did = reinterpret_cast<uint32_t*>(base + off); off += (size_t)kO2Cand * 4;
    uint32_t* bmS = nullptr; uint32_t* bm_e = nullptr; uint16_t* rankw = nullptr; uint8_t* c_w = nullptr;
    if (SPARSE) {
        bmS = reinterpret_cast<uint32_t*>(base + off); off += (size_t)kO2Sparse * NW * 4 * 2;
        bm_e = reinterpret_cast<uint32_t*>(base + off); off += (size_t)NW * 4;
        rankw = reinterpret_cast<uint16_t*>(base + off); off += (size_t)NW * 2;
        c_w = reinterpret_cast<uint8_t*>(base + off); off += (size_t)kO2Sparse * kO2Cand;
    }
    uint8_t* tk_m = reinterpret_cast<uint8_t*>(base + off);

    const uint32_t s_begin = wk.s_begin, s_end = wk.s_end;
    const bool empty = (q.flags & XGM_QF_EMPTY) || s_begin >= s_end || k == 0;
    for (uint32_t i = lane; i < cap; i += 64u) { tk_w[i] = 0; tk_d[i] = 0xFFFFFFFFu; tk_m[i] = 0; }
    for (uint32_t i = lane; i < XGM_OR_HIST; i += 64u) lh[i] = 0;
    if (SPARSE) for (uint32_t i = lane; i < kO2Sparse * kO2Cand; i += 64u) c_w[i] = 0;
    wave_lds_fence();

    /* lane t: term t's container index, or the slice of its flat array that lies in the unit's docid range */
    uint32_t dense_reg = kNoDense;
    bool present_reg = false;
    uint32_t fl_cur = 0, fl_end = 0;                                /* (indices into flat_did: plan_batch sends queries here only while the arrays hold < 2^32 postings) */
    if (!empty && lane < T) {
        const uint32_t id = q.term_id[lane];
        if (id != 0xFFFFFFFFu) {
            present_reg = true;
            if (seg.dense_id) dense_reg = seg.dense_id[id];
            if (dense_reg == kNoDense && seg.flat_off) { fl_cur = (uint32_t)seg.flat_off[id]; fl_end = (uint32_t)seg.flat_off[id + 1]; }
        }
    }
    const uint64_t present_mask = __ballot(present_reg);
    const uint64_t dense_mask = __ballot(present_reg && dense_reg != kNoDense);
    const uint64_t sparse_mask = present_mask & ~dense_mask;
    /* the flat terms (plan_batch sent the query here only if there are <= kO2Sparse of them): slot u = term sp_t[u] */
    uint32_t sp_t[kO2Sparse], n_sp = 0;
    {
        uint64_t sm = sparse_mask;
#pragma unroll
        for (uint32_t u = 0; u < kO2Sparse; ++u) { sp_t[u] = 0; if (sm) { sp_t[u] = (uint32_t)__builtin_ctzll(sm); sm &= sm - 1u; n_sp = u + 1u; } }
    }
    if (SPARSE) {
        /* the first posting at or after the unit's first docid: a 64-ary search of the term's slice, once per unit */
        for (uint32_t u = 0; u < n_sp; ++u) {
            const uint32_t lo = rl32(fl_cur, sp_t[u]), hi = rl32(fl_end, sp_t[u]);
            const uint32_t len = hi - lo;
            const uint32_t c = wave_lower_bound(seg.flat_did + lo, 0u, len, s_begin << SB, lane);
            if (lane == sp_t[u]) fl_cur = lo + c;
            if (TALLY) { cn_aux += 64u; }
        }
    }

    /* MaxScore order (ascending weight bound), prefix sums, quantisation: as xgm_orw_kernel */
    double prefix_reg = 0.0, ub_reg = 0.0, ub1_reg = 0.0;
    uint32_t rank_reg = 0;
    if (lane < T) {
        const double my = q.ub[lane];
        ub_reg = my;
        ub1_reg = (dense_reg != kNoDense && !seg.dense_plane) ? my : q.ub1[lane];
        for (uint32_t j = 0; j < T; ++j) {
            const double uj = q.ub[j];
            if (uj < my || (uj == my && j <= lane)) { prefix_reg += uj; ++rank_reg; }
        }
        prefix_reg *= 1.000000001;
    }
    /* every present term in ascending order of its bound, 4 bits each: the order the bound sum adds them in */
    uint64_t term_ord = 0;
    uint32_t n_ord = 0;
    for (uint32_t p = 1; p <= T; ++p) {
        const uint64_t m = __ballot(lane < T && rank_reg == p);
        if (m) {
            const uint32_t t = (uint32_t)__builtin_ctzll(m);
            if ((present_mask >> t) & 1ull) { term_ord |= (uint64_t)t << (4u * n_ord); ++n_ord; }
        }
    }
    const uint64_t top_mask = __ballot(present_reg && rank_reg == T);
    const uint32_t r_term = top_mask ? (uint32_t)__builtin_ctzll(top_mask) : 0u;
    const uint64_t mp_bits = top_mask ? rl64((uint64_t)__double_as_longlong(prefix_reg), r_term) : 0ull;
    const int hbase = (int)(mp_bits >> kHistShift) - (int)(XGM_OR_HIST - 1u);
    const bool prune = top_mask != 0ull && hbase > 0 && !empty;
    uint32_t* hist_g = hist_all + (size_t)wk.qi * XGM_OR_HIST;
    uint64_t seed_bits = 0;
    if (prune) {
        const uint64_t sb = (uint64_t)__double_as_longlong(q.theta_seed) >> kHistShift;
        const uint32_t sb_lo = __builtin_amdgcn_readfirstlane((uint32_t)sb);
        const int sbk = (int)sb_lo - hbase;
        if (sbk >= 1) seed_bits = (uint64_t)(sbk > (int)XGM_OR_HIST - 1 ? (uint32_t)hbase + XGM_OR_HIST - 1u : sb_lo) << kHistShift;
    }

    uint32_t tkn = 0;
    bool theta_valid = false;
    uint64_t theta_w = 0;
    uint32_t theta_d = 0;
    uint64_t theta_glob = 0;
    uint32_t matches32 = 0;                                         /* per lane: a unit's stripes x 1 word x 4 passes x 32 documents < 2^32 */
    uint32_t n_scored = 0;
    bool lh_dirty = false;
    const uint32_t n_local = empty ? 0u : s_end - s_begin;

    for (uint64_t dm = dense_mask; dm; dm &= dm - 1u) {
        const uint32_t t = (uint32_t)__builtin_ctzll(dm);
        const uint32_t dr = rl32(dense_reg, t);
        if (TALLY) { cn_aux += n_local; }
        for (uint32_t i = lane; i < n_local; i += 64u) dir_tab[t * SPG + i] = seg.dense_dir[(size_t)dr * seg.n_stripes + (s_begin + i)];
    }
    wave_lds_fence();

    bool fix = false;
    uint32_t qA2_reg = 0, qA1_reg = 0;

    /* one round of gathers a round ahead of its use: the document length and the container terms' wdf bytes */
    uint32_t pf_dl = 1u, pf_pb[kO2MaxT] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto prefetch_round = [&](uint32_t i0, uint32_t n_c) {
        const uint32_t o = i0 + lane;
        const bool valid = o < n_c;
        const uint32_t cd = valid ? c_did[o] : 0u;
        const uint32_t slot = cd & (W - 1u), csl = valid ? (cd >> SB) - s_begin : 0u;
        pf_dl = valid ? seg.doclen[cd] : 1u;
        if (TALLY) { const uint32_t nv = n_c - i0 < 64u ? n_c - i0 : 64u; cn_dl_raw += nv; cn_probe_raw += nv * (uint32_t)__popcll(dense_mask); }
#pragma unroll
        for (uint32_t t = 0; t < kO2MaxT; ++t) {
            pf_pb[t] = 0;
            if ((dense_mask >> t) & 1ull) {
                const uint32_t oo = valid ? dir_tab[t * SPG + csl] : 0u;
                if (oo) pf_pb[t] = seg.dense_data[(size_t)oo * 16 + (size_t)NW * 4 + slot];
            }
        }
    };

    auto score_candidates = [&](uint32_t n_c) {
        n_scored += n_c;
        for (uint32_t i0 = 0; i0 < n_c; i0 += 64u) {
            if (tkn + 64u > cap) {
                orw_topk_sort(tk_w, tk_d, tk_m, cap, lane);
                tkn = tkn < k ? tkn : k;
                if (tkn == k) { theta_valid = true; theta_w = tk_w[k - 1]; theta_d = tk_d[k - 1]; }
                for (uint32_t i = tkn + lane; i < cap; i += 64u) { tk_w[i] = 0; tk_d[i] = 0xFFFFFFFFu; tk_m[i] = 0; }
                wave_lds_fence();
            }
            const uint32_t o = i0 + lane;
            const bool valid = o < n_c;
            const uint32_t did = valid ? c_did[o] : 0u;
            const uint32_t dlen = pf_dl;
            uint32_t pb[kO2MaxT];
#pragma unroll
            for (uint32_t t = 0; t < kO2MaxT; ++t) pb[t] = pf_pb[t];
#ifndef XGM_ORW2_NO_PREFETCH
            if (i0 + 64u < n_c) prefetch_round(i0 + 64u, n_c);
#endif
            /* BM25Weight::get_sumpart, bm25weight.cc:170-181 — same operations, same order */
            const double len = (double)dlen;
            double normlen = len * q.len_factor;
            normlen = normlen > q.min_normlen ? normlen : q.min_normlen;
            const double denom_len = q.k1 * (normlen * q.b + (1.0 - q.b));
            uint32_t subqs = 0, sumA = 0;
            orw_d8 v;
#pragma unroll
            for (uint32_t g = 0; g < 2u; ++g) {
                uint32_t ev[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    const uint32_t t = g * 4u + u;
                    ev[u] = pb[t];
                    if (SPARSE) {
#pragma unroll
                        for (uint32_t x = 0; x < kO2Sparse; ++x)
                            if (x < n_sp && sp_t[x] == t && valid) ev[u] = (uint32_t)c_w[x * kO2Cand + o];
                    }
                }
                double wt[4] = {-0.0, -0.0, -0.0, -0.0};
                if (g * 4u < T && __ballot((ev[0] | ev[1] | ev[2] | ev[3]) != 0u)) {
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) {
                        const double wdf = (double)(ev[u] - 1u);
                        const double denom = denom_len + wdf;
                        const double x = q.termweight[g * 4u + u] * (wdf / denom);
                        wt[u] = ev[u] ? x : -0.0;
                        subqs += ev[u] ? 1u : 0u;
                        if (fix && g * 4u + u < T) {
                            const uint32_t t = g * 4u + u;
                            const uint32_t a2_ = rl32(qA2_reg, t), a1_ = rl32(qA1_reg, t);
                            sumA += ev[u] ? (ev[u] >= 3u ? a2_ : a1_) : 0u;
                        }
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) v[g * 4u + u] = wt[u];
            }
            if (SPARSE) {
#pragma unroll
                for (uint32_t x = 0; x < kO2Sparse; ++x) if (x < n_sp && valid) c_w[x * kO2Cand + o] = 0;       /* the queue slot is free again */
            }
            for (uint32_t j = 0; j + 1u < T; ++j) {
                const uint32_t a = (uint32_t)(prog_a >> (8u * j)) & 7u, b = (uint32_t)(prog_b >> (8u * j)) & 7u;
                const double x = v[a] + v[b];
                v[a] = x;
            }
            const double weight = v[prog_root & 7u];
            const uint64_t wb = (uint64_t)__double_as_longlong(weight);
            const bool in_first = fix && sumA >= kQ;
            const bool live = valid && subqs != 0u && wb >= theta_glob && !in_first;
            if (prune && live) {
                int b = (int)(wb >> kHistShift) - hbase;
                b = b < 0 ? 0 : (b > (int)XGM_OR_HIST - 1 ? (int)XGM_OR_HIST - 1 : b);
                atomicAdd(&lh[b], 1u);
            }
            if (prune && __ballot(live)) lh_dirty = true;
            const bool take = live && (!theta_valid || cand_before(wb, did, theta_w, theta_d));
            const uint64_t tm = __ballot(take);
            if (take) { const uint32_t p = tkn + mbcnt(tm); tk_w[p] = wb; tk_d[p] = did; tk_m[p] = (uint8_t)subqs; }
            tkn += (uint32_t)__popcll(tm);
#ifdef XGM_ORW2_NO_PREFETCH
            if (i0 + 64u < n_c) prefetch_round(i0 + 64u, n_c);
#endif
        }
    };
    uint32_t qn = 0;
    auto flush_queue = [&]() {
        if (qn == 0u) return;
        wave_lds_fence();
        prefetch_round(0u, qn);
        score_candidates(qn);
        wave_lds_fence();
        qn = 0;
    };

    uint64_t qz_bits = 0, qz_ess = 0;
    uint32_t qz_q2 = 0, qz_q1 = 0;
    auto quantise = [&](uint64_t th_bits, uint64_t& ess, uint32_t& q2, uint32_t& q1) {
        if (th_bits == qz_bits) { ess = qz_ess; q2 = qz_q2; q1 = qz_q1; return; }
        const double th = __longlong_as_double((long long)th_bits);
        ess = __ballot(present_reg && !(prefix_reg < th));
        const double r2 = ub_reg * (double)kQ / th, r1 = ub1_reg * (double)kQ / th;
        q2 = r2 >= (double)kQ ? kQ : (uint32_t)r2 + 1u;
        q1 = r1 >= (double)kQ ? kQ : (uint32_t)r1 + 1u;
        qz_bits = th_bits; qz_ess = ess; qz_q2 = q2; qz_q1 = q1;
    };
    auto hist_bound = [&](const uint32_t* hc) {
        const uint32_t s4 = hc[0] + hc[1] + hc[2] + hc[3];
        const uint32_t P = wave_incl_scan(s4);
        const uint32_t suf = __builtin_amdgcn_readlane(P, 63) - P + s4;
        const uint64_t okm = __ballot(suf >= k);
        if (okm) {
            const uint32_t Lh = 63u - (uint32_t)__builtin_clzll(okm);
            const uint32_t cum = __builtin_amdgcn_readlane(suf, Lh) - __builtin_amdgcn_readlane(s4, Lh);
            uint32_t bsel = 4u * Lh;
            const uint32_t c3 = __builtin_amdgcn_readlane(hc[3], Lh), c2 = __builtin_amdgcn_readlane(hc[2], Lh), c1 = __builtin_amdgcn_readlane(hc[1], Lh);
            if (cum + c3 >= k) bsel = 4u * Lh + 3u;
            else if (cum + c3 + c2 >= k) bsel = 4u * Lh + 2u;
            else if (cum + c3 + c2 + c1 >= k) bsel = 4u * Lh + 1u;
            if (bsel > 0u) {
                const uint64_t tb = (uint64_t)((uint32_t)hbase + bsel) << kHistShift;
                theta_glob = tb > theta_glob ? tb : theta_glob;
            }
        }
    };
    auto publish_hist = [&]() {
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i) {
            const uint32_t vv = lh[lane * 4u + i];
            if (vv) { atomicAdd(&hist_g[lane * 4u + i], vv); lh[lane * 4u + i] = 0; }
        }
        lh_dirty = false;
        wave_lds_fence();
    };

    const uint32_t fl_start = fl_cur;                               /* (the repair pass walks the flat slices again) */
    for (uint32_t pass = 0; pass < 2u; ++pass) {
        if (pass == 1u) {
            if (!seed_bits || empty) break;
            if (lh_dirty) publish_hist();
            uint32_t hc[4];
            if (TALLY) { cn_aux += XGM_OR_HIST; }
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) hc[i] = __hip_atomic_load(&hist_g[lane * 4u + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hist_bound(hc);
            const uint64_t th_now = theta_valid && theta_w > theta_glob ? theta_w : theta_glob;
            if (th_now >= seed_bits) break;
            fix = true;
            uint64_t essA;
            quantise(seed_bits, essA, qA2_reg, qA1_reg);
            fl_cur = fl_start;
        }
        for (uint32_t sl = 0; sl < n_local; ++sl) {
            const uint32_t stripe_base = (s_begin + sl) << SB;
            /* ---- threshold ---- */
            const bool have_th = seed_bits || theta_valid || theta_glob;
            const bool look = prune && (fix || !have_th || (sl & 3u) == 0u);
            /* ---- flat terms: this stripe's postings -> the term's LDS bitmap pair (kept in registers for the scatter) ---- */
            uint32_t sp_did[kO2Sparse], sp_wdf[kO2Sparse], sp_n[kO2Sparse];
            uint32_t sp_c0[kO2Sparse];
#pragma unroll
            for (uint32_t u = 0; u < kO2Sparse; ++u) { sp_did[u] = 0; sp_wdf[u] = 0; sp_n[u] = 0; sp_c0[u] = 0; }
            if (SPARSE) {
#pragma unroll
                for (uint32_t u = 0; u < kO2Sparse; ++u) {
                    if (u < n_sp) {
                        uint32_t* bA = bmS + (size_t)u * NW * 2u;
                        uint32_t* bB = bA + NW;
                        for (uint32_t w = lane; w < NW; w += 64u) { bA[w] = 0; bB[w] = 0; }
                        wave_lds_fence();
                        uint32_t c = rl32(fl_cur, sp_t[u]);
                        const uint32_t e = rl32(fl_end, sp_t[u]);
                        sp_c0[u] = c;
                        const uint32_t stripe_end = stripe_base + W;
                        while (true) {
                            const uint32_t g = c + lane;
                            const bool in_arr = g < e;
                            const uint32_t d = in_arr ? seg.flat_did[g] : 0xFFFFFFFFu;
                            const uint32_t wf = in_arr ? (uint32_t)seg.flat_wdf[g] : 0u;
                            if (TALLY) { cn_flat += 64u; }
                            const bool in = in_arr && d < stripe_end;             /* (the cursor sits at the stripe's first posting: d >= stripe_base) */
                            const uint32_t n_in = (uint32_t)__popcll(__ballot(in));
                            if (in) {
                                const uint32_t s = d - stripe_base;
                                atomicOr(&bA[s >> 5], 1u << (s & 31u));
                                if (wf >= 2u) atomicOr(&bB[s >> 5], 1u << (s & 31u));
                            }
                            if (sp_n[u] == 0u) { sp_did[u] = d; sp_wdf[u] = wf; }      /* (the stripe's first 64: what the scatter uses unless there are more) */
                            sp_n[u] += n_in;
                            c += n_in;
                            if (n_in < 64u) break;
                        }
                        if (lane == sp_t[u]) fl_cur = c;
                        wave_lds_fence();
                    }
                }
            }
            /* ---- this stripe's threshold, quantised bounds ---- */
            uint64_t ess_mask = present_mask;
            bool use_sum = false;
            uint32_t q2_reg = kQ, q1_reg = kQ;
            if (prune) {
                if (look) {
                    uint32_t hc[4];
                    if (TALLY) { cn_aux += XGM_OR_HIST; }
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i) hc[i] = __hip_atomic_load(&hist_g[lane * 4u + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hist_bound(hc);
                }
                uint64_t th_bits = theta_valid && theta_w > theta_glob ? theta_w : theta_glob;
                if (fix && th_bits >= seed_bits) break;                      /* whatever is left was weighed by the first pass or cannot reach the top k */
                if (!fix && seed_bits > th_bits) th_bits = seed_bits;
                if (th_bits) { quantise(th_bits, ess_mask, q2_reg, q1_reg); use_sum = true; }
            }
            const bool want_planes = use_sum && seg.dense_plane != 0u;
            const uint32_t hc_cur = (present_reg && dense_reg != kNoDense) ? dir_tab[lane * SPG + sl] : 0u;      /* lane t: container of term t in this stripe */

            /* ---- the passes: word w = i * 64 + lane of every term's bitmap pair ---- */
            for (uint32_t i = 0; i * 64u < NW; ++i) {
                const uint32_t w = i * 64u + lane;
                const bool wv = w < NW;
                uint32_t xb[kO2MaxT], xp[kO2MaxT];
#pragma unroll
                for (uint32_t u = 0; u < kO2MaxT; ++u) {
                    xb[u] = 0; xp[u] = 0;
                    if (u < n_ord) {
                        const uint32_t t = (uint32_t)(term_ord >> (4u * u)) & 15u;
                        const uint32_t oo = rl32(hc_cur, t);
                        if ((dense_mask >> t) & 1ull) {
                            if (TALLY) { if (oo) cn_bmpw += (want_planes ? 2u : 1u) * (NW < 64u ? NW : 64u); }
                            if (oo && wv) {
                                xb[u] = reinterpret_cast<const uint32_t*>(seg.dense_data + (size_t)oo * 16)[w];
                                if (want_planes) xp[u] = reinterpret_cast<const uint32_t*>(seg.dense_data + (size_t)oo * 16 + seg.dense_plane)[w];
                            }
                        } else if (SPARSE) {
#pragma unroll
                            for (uint32_t x = 0; x < kO2Sparse; ++x)
                                if (x < n_sp && sp_t[x] == t && wv) { xb[u] = bmS[(size_t)x * NW * 2u + w]; xp[u] = bmS[(size_t)x * NW * 2u + NW + w]; }
                        }
                    }
                }
                /* union (exact match count) and the bit-sliced sum of the quantised bounds: six planes, transient per pass */
                uint32_t a = 0, ovf = 0;
                uint32_t S0 = 0, S1 = 0, S2 = 0, S3 = 0, S4 = 0, S5 = 0;
                uint32_t max_sum = 0;
#pragma unroll
                for (uint32_t u = 0; u < kO2MaxT; ++u) {
                    if (u < n_ord) {
                        a |= xb[u];
                        if (use_sum) {
                            const uint32_t t = (uint32_t)(term_ord >> (4u * u)) & 15u;
                            uint32_t q1v = rl32(q1_reg, t), q2v = rl32(q2_reg, t);
                            const bool sparse_t = SPARSE && !((dense_mask >> t) & 1ull);
                            if (!want_planes && !sparse_t) q1v = q2v;                 /* no wdf >= 2 plane at hand: every document at the term's maximum */
                            const uint32_t B = xb[u], P = xp[u] & xb[u], lo = B & ~P;
                            /* documents of lo get q1v, those of P get q2v >= q1v */
                            if (q1v >= kQ) { ovf |= B; continue; }
                            uint32_t addlo = lo, addP = P, qlo = q1v, qP = q2v;
                            if (q2v >= kQ) { ovf |= P; addP = 0; qP = 0; }
                            if (q1v == q2v) { addlo = B; addP = 0; qP = 0; }
                            const uint32_t qmax = qP > qlo ? qP : qlo;
                            const uint32_t nm = max_sum + qmax;
                            max_sum = nm < kQ ? nm : kQ;
                            const uint32_t np = nm >= kQ ? 6u : 32u - (uint32_t)__builtin_clz(nm | 1u);
                            uint32_t carry = 0;
#define O2_PLANE(J, SJ)                                                                                                           \
                            if (J < np) {                                                                                         \
                                const uint32_t b1 = (qlo >> J) & 1u, b2 = (qP >> J) & 1u;                                         \
                                const uint32_t ad = (b1 ? addlo : 0u) | (b2 ? addP : 0u);                                         \
                                const uint32_t sj = SJ, xo = sj ^ ad;                                                             \
                                SJ = xo ^ carry;                                                                                  \
                                carry = orw_bfi(xo, carry, sj);                                                                   \
                            }
                            O2_PLANE(0u, S0) O2_PLANE(1u, S1) O2_PLANE(2u, S2) O2_PLANE(3u, S3) O2_PLANE(4u, S4) O2_PLANE(5u, S5)
#undef O2_PLANE
                            if (np == 6u) ovf |= carry;
                        }
                    }
                }
                (void)S5;
                const uint32_t e = use_sum ? ovf : a;
                if (!fix) matches32 += (uint32_t)__popc(a);
                /* ---- candidates of this pass, in docid order, to the queue (chunks of lanes holding <= kO2Cand of them) ---- */
                const uint32_t cnt = (uint32_t)__popc(e);
                const uint32_t incl = wave_incl_scan(cnt);
                const uint32_t n_total = __builtin_amdgcn_readlane(incl, 63);
                if (SPARSE && wv) bm_e[w] = e;
                for (uint32_t lane_lo = 0; lane_lo < 64u && n_total != 0u;) {
                    const uint32_t ord_base = lane_lo ? __builtin_amdgcn_readlane(incl, lane_lo - 1u) : 0u;
                    if (ord_base == n_total) break;
                    const uint64_t fit = __ballot(incl - ord_base <= kO2Cand);
                    const uint32_t lane_hi = 64u - (uint32_t)__builtin_clzll(fit);
                    const uint32_t hi_incl = __builtin_amdgcn_readlane(incl, lane_hi - 1u);
                    const uint32_t n_c = hi_incl - ord_base;
                    const bool in_chunk = lane >= lane_lo && lane < lane_hi;
                    const uint32_t wlo = i * 64u + lane_lo, whi = i * 64u + lane_hi;
                    lane_lo = lane_hi;
                    if (n_c == 0u) continue;
                    if (qn + n_c > kO2Cand) flush_queue();
                    const uint32_t qb = qn;
                    if (in_chunk) {
                        uint32_t o = qb + incl - cnt - ord_base;
                        if (SPARSE && wv) rankw[w] = (uint16_t)o;
                        uint32_t m = e;
                        while (m) {
                            const uint32_t bit = (uint32_t)__ffs(m) - 1u;
                            c_did[o] = stripe_base + w * 32u + bit;
                            m &= m - 1u;
                            ++o;
                        }
                    }
                    wave_lds_fence();
                    if (SPARSE) {
                        /* wdf of the flat terms for this chunk's candidates: from the registers the bitmaps were made of (a stripe with more
                         * than 64 postings of the term: its flat slice once more) */
#pragma unroll
                        for (uint32_t u = 0; u < kO2Sparse; ++u) {
                            if (u < n_sp && sp_n[u]) {
                                uint32_t done = 0;
                                const bool reload = sp_n[u] > 64u;
                                while (done < sp_n[u]) {
                                    uint32_t d = sp_did[u], wf = sp_wdf[u];
                                    const uint32_t nb = sp_n[u] - done < 64u ? sp_n[u] - done : 64u;
                                    if (reload) {
                                        const uint32_t g = sp_c0[u] + done + lane;
                                        d = lane < nb ? seg.flat_did[g] : 0xFFFFFFFFu;
                                        wf = lane < nb ? (uint32_t)seg.flat_wdf[g] : 0u;
                                    }
                                    if (lane < nb) {
                                        const uint32_t s = d - stripe_base, wd = s >> 5, bit = s & 31u;
                                        if (wd >= wlo && wd < whi) {
                                            const uint32_t bm = bm_e[wd];
                                            if ((bm >> bit) & 1u) c_w[u * kO2Cand + (uint32_t)rankw[wd] + (uint32_t)__popc(bm & ((1u << bit) - 1u))] = (uint8_t)(wf + 1u);
                                        }
                                    }
                                    done += nb;
                                }
                            }
                        }
                        wave_lds_fence();
                    }
                    qn = qb + n_c;
                    if (qn >= 128u || !use_sum) flush_queue();
                }
            }
            if (lh_dirty && (fix || !have_th || (sl & 3u) == 2u || sl + 1u >= n_local)) publish_hist();
        }
        flush_queue();
        if (lh_dirty) publish_hist();
    }

    /* ---- unit epilogue ---- */
    orw_topk_sort(tk_w, tk_d, tk_m, cap, lane);
    unsigned long long matches = matches32;
    for (int sh = 32; sh > 0; sh >>= 1) matches += (unsigned long long)__shfl_xor((long long)matches, sh);
    const uint32_t n_out = tkn < k ? tkn : k;
    xgm_cand* out = cand_out + (size_t)wk.slot * k_stride;
    for (uint32_t i = lane; i < n_out; i += 64u) {
        xgm_cand c;
        c.wbits = tk_w[i]; c.did = tk_d[i]; c.subqs = tk_m[i];
        out[i] = c;
    }
    if (lane == 0) {
        xgm_group_hdr h;
        h.matches = matches; h.n_cand = n_out; h.pad = n_scored;
        h.t_start = t_unit_start; h.t_end = __builtin_readcyclecounter();
        h.c_pos = 0; h.c_bmp_words = cn_bmpw; h.c_probes = cn_probe_raw; h.c_blk_words = cn_flat * 5u / 4u; h.c_hdrs = 0;
        h.c_doclen = cn_dl_raw; h.c_aux_words = cn_aux; h.c_probes_raw = cn_probe_raw; h.c_doclen_raw = cn_dl_raw; h.c_pad[0] = fix ? 0x80000000u : 0u; h.c_pad[1] = 0;
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

static unsigned long long* g_orw_cycles = nullptr;          /* device buffer, diagnostics only */

/* Diagnostics (XGM_PHASE_TIMING=1): cycle sums per section of xgm_orw_kernel since the last call:
 * [0] threshold + dense bitmaps + bound sum, [1] block decode + candidate set, [2] enumerate,
 * [3] dense probes, [4] block scatter, [5] scoring loop overhead, [6] histogram flush, [7] setup,
 * [8] top-k sort, [9] doclen wait + leaf weights, [10] tree sum, [11] histogram + top-k insert.  out: u64[16]. */
int xgm_orw_cycles_fetch(unsigned long long* out8) {
    if (!g_orw_cycles) return -1;
    (void)hipDeviceSynchronize();
    hipMemcpy(out8, g_orw_cycles, 128, hipMemcpyDeviceToHost);
    hipMemset(g_orw_cycles, 0, 128);
    return 0;
}

size_t xgm_orw_smem_bytes(uint32_t stripe_bits, uint32_t tab_terms, uint32_t cap, bool wide, uint32_t spg) {
    return XGM_WAVES * orw_wave_bytes(1u << stripe_bits, tab_terms, cap, wide ? 2 : 1, spg);
}

size_t xgm_orw2_smem_bytes(uint32_t stripe_bits, uint32_t tab_terms, uint32_t cap, uint32_t spg, bool sparse) {
    return XGM_WAVES * orw2_wave_bytes(1u << stripe_bits, tab_terms, cap, spg, sparse);
}

/* OPT-IN (XGM_ORW2=1): measured on the MI355X in round 5 the kernel is SLOWER than xgm_orw_kernel on C3 — 2.30 ms per launch of 250 queries
 * against 1.98 ms for 256 — although it runs three waves per SIMD without a spill in its loops where the old one runs two: taking a stripe
 * in four single-word passes repeats the wave-uniform control of the bound sum (plane / addend selection, lane reads of the quantised
 * bounds) four times, and that, not occupancy, is what the launch is made of (static count: 835 VALU + 536 SALU in one pass's sum).
 * Kept as an A/B kernel with its parity tests (tests/test_gpu_variants.py); it implements the default pruning configuration only. */
bool xgm_orw2_enabled() {
    static const bool on = getenv("XGM_ORW2") && atoi(getenv("XGM_ORW2")) && !getenv("XGM_NO_PRUNE") && !getenv("XGM_NO_PHASE_A") && !getenv("XGM_NO_BOUND_SUM") && !getenv("XGM_PHASE_TIMING");
    return on;
}

static int launch_orw2(const xgm_match_launch& L, uint32_t* hist, hipStream_t stream) {
    const bool sparse = L.orw2 == 2;
    const size_t smem = xgm_orw2_smem_bytes(L.seg.stripe_bits, L.tab_terms, L.cap, L.stripes_per_group, sparse);
    const dim3 grid((L.n_work + XGM_WAVES - 1u) / XGM_WAVES), block(XGM_WG);
    int rc = XGM_OK;
#define ORW2_LAUNCH(SP, TL)                                                                                                          \
    do {                                                                                                                             \
        auto kern = xgm_orw2_kernel<SP, TL>;                                                                                         \
        static std::atomic<size_t> seen{0};                                                                                          \
        if ((rc = orw_ensure_dyn_smem(kern, smem, seen))) return rc;                                                                 \
        hipLaunchKernelGGL(kern, grid, block, smem, stream, L.seg, L.queries, L.work, L.n_work, L.stripes_per_group, L.tab_terms,    \
                           L.cap, L.k_stride, hist, L.cand, L.ghdr);                                                                 \
    } while (0)
    if (sparse) { if (L.tally) ORW2_LAUNCH(true, true); else ORW2_LAUNCH(true, false); }
    else { if (L.tally) ORW2_LAUNCH(false, true); else ORW2_LAUNCH(false, false); }
#undef ORW2_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xgm_launch_error("xgm_orw2_kernel launch", (int)e, hipGetErrorString(e));
    return XGM_OK;
}

int xgm_launch_orw(const xgm_match_launch& L, uint32_t* hist, hipStream_t stream) {
    if (L.orw2) return launch_orw2(L, hist, stream);
    static const bool no_prune = getenv("XGM_NO_PRUNE") != nullptr;          /* A/B switches for measurements */
    static const bool no_phase_a = getenv("XGM_NO_PHASE_A") != nullptr;
    static const bool no_sum = getenv("XGM_NO_BOUND_SUM") != nullptr;
    static const bool timing = getenv("XGM_PHASE_TIMING") != nullptr;
    if (timing && !g_orw_cycles) { hipMalloc((void**)&g_orw_cycles, 128); hipMemset(g_orw_cycles, 0, 128); }
    /* bit 0: prune; bit 1: start from the planner's guess of the k-th weight (XGM_NO_PHASE_A keeps its old name: no seeded first pass);
     * bit 2: term-level MaxScore only, no bound sum (then no guess either) */
    const int flags = no_prune ? 0 : ((no_phase_a ? 1 : 3) | (no_sum ? 4 : 0));
    const bool p4 = L.orw_planes == 4;                             /* (plan_batch: every query of the batch has 4-8 terms; XGM_ORW_PLANES forces 4 or 6) */
    const size_t smem = xgm_orw_smem_bytes(L.seg.stripe_bits, L.tab_terms, L.cap, L.wide, L.stripes_per_group);
    const dim3 grid((L.n_work + XGM_WAVES - 1u) / XGM_WAVES), block(XGM_WG);
    int rc = XGM_OK;
#define ORW_LAUNCH(TABT, TL)                                                                                                         \
    do {                                                                                                                             \
        auto kern = p4 ? xgm_orw_kernel<TABT, TL, false, 4> : xgm_orw_kernel<TABT, TL, false, 6>;                                    \
        static std::atomic<size_t> seen[2];                          /* (one per instantiation: [p4]) */                            \
        if ((rc = orw_ensure_dyn_smem(kern, smem, seen[p4 ? 1 : 0]))) return rc;                                                     \
        XGM_LAUNCH_TIMED(L, kern, grid, block, smem, stream, L.seg, L.queries, L.work, L.n_work, L.stripes_per_group, L.tab_terms,    \
                           L.cap, L.k_stride, hist, flags, L.cand, L.ghdr, g_orw_cycles, L.fuse);                                    \
    } while (0)
#define ORW_LAUNCH_FLAT(TL)                                                                                                          \
    do {                                                                                                                             \
        auto kern = p4 ? xgm_orw_kernel<uint8_t, TL, true, 4> : xgm_orw_kernel<uint8_t, TL, true, 6>;                                \
        static std::atomic<size_t> seen[2];                          /* (one per instantiation: [p4]) */                            \
        if ((rc = orw_ensure_dyn_smem(kern, smem, seen[p4 ? 1 : 0]))) return rc;                                                     \
        XGM_LAUNCH_TIMED(L, kern, grid, block, smem, stream, L.seg, L.queries, L.work, L.n_work, L.stripes_per_group, L.tab_terms,    \
                           L.cap, L.k_stride, hist, flags, L.cand, L.ghdr, g_orw_cycles, L.fuse);                                    \
    } while (0)
    /* L.tally: the instantiation that also fills the traffic tallies of xgm_group_hdr (measurement only) */
    if (L.or_flat && !L.wide) { if (L.tally) ORW_LAUNCH_FLAT(true); else ORW_LAUNCH_FLAT(false); }
    else if (L.wide) { if (L.tally) ORW_LAUNCH(uint16_t, true); else ORW_LAUNCH(uint16_t, false); }
    else { if (L.tally) ORW_LAUNCH(uint8_t, true); else ORW_LAUNCH(uint8_t, false); }
#undef ORW_LAUNCH
#undef ORW_LAUNCH_FLAT
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xgm_launch_error("xgm_orw_kernel launch", (int)e, hipGetErrorString(e));
    return XGM_OK;
}
