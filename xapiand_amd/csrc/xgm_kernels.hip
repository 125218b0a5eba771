/* HIP kernels of the match/rank path for gfx950 (wave64).  No MFMA: this is integer/pointer work
 * bounded by HBM and LDS, plus fp64 BM25 on the survivors.
 *
 *   xgm_match_kernel  — one workgroup per (query, docid range).  Per stripe of W docids:
 *        K1 decode   : each wave stages one 128-posting block's bit-packed payload into its LDS
 *                      window (coalesced 16-B loads), unpacks two postings per lane, rebuilds
 *                      docids with a DPP wave prefix sum, and scatters wdf+1 into a per-term
 *                      direct-address LDS table (reference: GlassPostList::next/skip_to varint
 *                      decode, glass_postlist.cc:768-991);
 *        K2/K3 match : a byte-parallel AND (or OR) of the tables finds matching slots, compacted
 *                      with ballot/mbcnt into an LDS queue (reference: MultiAndPostList::
 *                      find_next_match multiandpostlist.cc:180-207, OrPostList::next
 *                      orpostlist.cc:114-204);
 *        K6 phrase   : optional positional filter per survivor (ExactPhrasePostList::test_doc
 *                      exactphrasepostlist.cc:75-133, PhrasePostList::test_doc
 *                      phrasepostlist.cc:60-90);
 *        K4 bm25     : fp64 BM25Weight::get_sumpart (bm25weight.cc:170-181) per (doc, term), summed
 *                      in the reference's association (multiandpostlist.cc:150-160,
 *                      orpostlist.cc:94-103);
 *        K5 top-k    : threshold-filtered LDS candidate buffer + bitonic selection under the total
 *                      order of msetcmp_by_relevance<true> (msetcmp.cc:55-62; ProtoMSet::add
 *                      protomset.h:340-400).
 *   xgm_merge_kernel  — per query, merges the groups' (or shards') candidates into the final hits
 *                      (reference: ProtoMSet::finalise sort protomset.h:657; Matcher::merge_mset
 *                      matcher.cc:653-781; MSet::unshard_docids mset.cc:367-373).
 *   xgm_decode_kernel — K1 alone, whole term → (did, wdf) arrays; used to verify segments and as a
 *                      decode micro-benchmark.
 *
 * Compile with -ffp-contract=off: BM25 must round exactly like the reference's x86-64 doubles.
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "xgm_device.h"
#include "xgm_launch.h"
#include "xgm_wave.h"
#include "xgm_posfilter.h"

namespace {

struct TopK {
    uint64_t* w;       /* [cap] */
    uint32_t* d;       /* [cap] */
    uint32_t* m;       /* [cap] subqs matched */
    uint32_t cap;      /* power of two */
};

/* Bitonic sort of all `cap` entries into "before" order (best first); unused entries must hold the
 * sentinel (w = 0, d = UINT32_MAX, m = UINT32_MAX) which sorts last.  All threads of the WG. */
__device__ void topk_sort(const TopK& tk, uint32_t tid) {
    for (uint32_t size = 2; size <= tk.cap; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (uint32_t i = tid; i < (tk.cap >> 1); i += XGM_WG) {
                uint32_t lo = 2u * i - (i & (stride - 1u));
                uint32_t hi = lo + stride;
                bool asc = ((lo & size) == 0);       /* this run sorts best-first */
                uint64_t aw = tk.w[lo], bw = tk.w[hi];
                uint32_t ad = tk.d[lo], bd = tk.d[hi];
                bool swap = asc ? cand_before(bw, bd, aw, ad) : cand_before(aw, ad, bw, bd);
                if (swap) {
                    tk.w[lo] = bw; tk.w[hi] = aw;
                    tk.d[lo] = bd; tk.d[hi] = ad;
                    uint32_t am = tk.m[lo], bm = tk.m[hi];
                    tk.m[lo] = bm; tk.m[hi] = am;
                }
            }
        }
    }
    __syncthreads();
}

/* ---------------------------------------------------------------- control block -------------- */

struct Ctrl {
    uint32_t cur[XGM_MAX_TERMS];      /* next unread block of each term (global block index)      */
    uint32_t end[XGM_MAX_TERMS];      /* end of the term's blocks inside this group's docid range  */
    uint32_t st[XGM_MAX_TERMS];       /* stripe of block cur[t], kInfStripe when exhausted         */
    uint32_t run[XGM_MAX_TERMS];      /* number of blocks of stripe st[t] starting at cur[t]       */
    uint32_t qn;                      /* queue fill                                                */
    uint32_t tkn;                     /* top-k buffer fill                                         */
    uint32_t theta_valid;
    uint32_t theta_d;
    uint64_t theta_w;
    unsigned long long matches;
};

struct KernelSmem {
    uint32_t* runs;       /* [2][T][SPG]: block range of every (term, stripe) of the group */
    unsigned char* tab;   /* [T][W] of TabT */
    uint32_t* ptab;       /* [T][W] (phrase) */
    uint16_t* queue;      /* [W] */
    TopK tk;
    uint32_t* stage;      /* [WAVES][kStageWords] */
    Ctrl* ctrl;
};

template <typename TabT>
__device__ __forceinline__ KernelSmem carve(unsigned char* smem, uint32_t W, uint32_t T, bool phrase, uint32_t cap, uint32_t spg) {
    KernelSmem s;
    size_t off = 0;
    s.tk.w = reinterpret_cast<uint64_t*>(smem + off); off += (size_t)cap * 8;
    s.ctrl = reinterpret_cast<Ctrl*>(smem + off); off += (sizeof(Ctrl) + 15) & ~(size_t)15;
    s.tab = smem + off; off += (size_t)T * W * sizeof(TabT);
    s.ptab = reinterpret_cast<uint32_t*>(smem + off); off += phrase ? (size_t)T * W * 4 : 0;
    s.tk.d = reinterpret_cast<uint32_t*>(smem + off); off += (size_t)cap * 4;
    s.tk.m = reinterpret_cast<uint32_t*>(smem + off); off += (size_t)cap * 4;
    s.stage = reinterpret_cast<uint32_t*>(smem + off); off += (size_t)XGM_WAVES * kStageWords * 4;
    s.queue = reinterpret_cast<uint16_t*>(smem + off); off += (size_t)W * 2;
    s.runs = reinterpret_cast<uint32_t*>(smem + off); off += (size_t)2 * T * spg * 4;
    s.tk.cap = cap;
    return s;
}

/* nonzero-byte mask of a packed word: bit 7 of every byte that is != 0 */
__device__ __forceinline__ uint32_t nz_bytes(uint32_t x) {
    return (x | ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu)) & 0x80808080u;
}
__device__ __forceinline__ uint32_t nz_halves(uint32_t x) {
    return (x | ((x & 0x7FFF7FFFu) + 0x7FFF7FFFu)) & 0x80008000u;
}

/* ---------------------------------------------------------------- K6: positional filter ------ */

/* (K6 fast path — positions staged in the wave's LDS, the predicates on LDS data — lives in xgm_posfilter.h: shared with the dense body) */

/* ---------------------------------------------------------------- the match kernel ----------- */

#define XGM_BODY_SORTED 0
#include "xgm_match_body.inc"
#undef XGM_BODY_SORTED

/* ---------------------------------------------------------------- the AND kernel ------------- */

/* Conjunctions get their own kernel (K2 specialised): the rarest term's postings of a stripe ARE the
 * candidate set (the reference drives MultiAndPostList from its shortest list the same way,
 * multiandpostlist.cc:180-207), so instead of per-term direct-address tables over the whole stripe
 * the workgroup keeps
 *     c_slot[ord], c_w[t][ord]   candidates in docid order and their wdf+1 per term (0 = absent)
 *     bitmap[W/32], rankw[W/32]  candidate membership + ordinal of the first candidate of each word
 * and only decodes those blocks of the other terms whose docid range contains a candidate.  LDS drops
 * from ~54 KB to ~20 KB per workgroup, which is what lets 7 workgroups (28 waves) share a CU and hide
 * the LDS round trips of the decode. */
constexpr uint32_t kAndCand = 1024;                 /* candidates per chunk = 8 blocks of term 0 */
constexpr uint32_t kAndChunkBlocks = kAndCand / XGM_BLOCK;
constexpr uint32_t kAndWindows = 1;                 /* blocks a wave unpacks at once (measured: 4-way interleave is slower - more LDS, fewer resident workgroups) */

struct AndCtrl {
    uint32_t cur[XGM_MAX_TERMS];
    uint32_t end[XGM_MAX_TERMS];
    uint64_t tbase[XGM_MAX_TERMS];
    uint32_t tkn, theta_valid, theta_d, pad;
    uint64_t theta_w;
    unsigned long long matches;
    unsigned long long coarse;        /* bit i: some candidate has slot in [128 i, 128 i + 127] */
};

struct AndSmem {
    TopK tk;
    AndCtrl* ctrl;
    uint32_t* runs;       /* [2][T][SPG] */
    uint32_t* stage;      /* [WAVES][kStageWords] */
    uint32_t* bitmap;     /* [W/32] */
    uint32_t* rankw;      /* [W/32] ordinal of the first candidate in the word (UINT32_MAX if none) */
    uint16_t* c_slot;     /* [kAndCand] */
    unsigned char* c_w;   /* [T][kAndCand] of TabT */
};

__host__ __device__ inline size_t and_smem_bytes(uint32_t W, uint32_t T, uint32_t cap, size_t tab_elem, uint32_t spg) {
    size_t off = 0;
    off += (size_t)cap * 8;                                    /* tk.w */
    off += (sizeof(AndCtrl) + 15) & ~(size_t)15;
    off += (size_t)cap * 4 * 2;                                /* tk.d, tk.m */
    off += (size_t)2 * T * spg * 4;                            /* runs */
    off += (size_t)XGM_WAVES * kAndWindows * kStageWords * 4;
    off += (size_t)(W / 32u) * 4 * 2;                          /* bitmap, rankw */
    off += (size_t)kAndCand * 2;                               /* c_slot */
    off += (size_t)T * kAndCand * tab_elem;                    /* c_w */
    return (off + 15) & ~(size_t)15;
}

template <typename TabT>
__device__ __forceinline__ AndSmem and_carve(unsigned char* smem, uint32_t W, uint32_t T, uint32_t cap, uint32_t spg) {
    AndSmem s;
    size_t off = 0;
    s.tk.w = reinterpret_cast<uint64_t*>(smem + off); off += (size_t)cap * 8;
    s.ctrl = reinterpret_cast<AndCtrl*>(smem + off); off += (sizeof(AndCtrl) + 15) & ~(size_t)15;
    s.tk.d = reinterpret_cast<uint32_t*>(smem + off); off += (size_t)cap * 4;
    s.tk.m = reinterpret_cast<uint32_t*>(smem + off); off += (size_t)cap * 4;
    s.runs = reinterpret_cast<uint32_t*>(smem + off); off += (size_t)2 * T * spg * 4;
    s.stage = reinterpret_cast<uint32_t*>(smem + off); off += (size_t)XGM_WAVES * kAndWindows * kStageWords * 4;
    s.bitmap = reinterpret_cast<uint32_t*>(smem + off); off += (size_t)(W / 32u) * 4;
    s.rankw = reinterpret_cast<uint32_t*>(smem + off); off += (size_t)(W / 32u) * 4;
    s.c_slot = reinterpret_cast<uint16_t*>(smem + off); off += (size_t)kAndCand * 2;
    s.c_w = smem + off; off += (size_t)T * kAndCand * sizeof(TabT);
    s.tk.cap = cap;
    return s;
}

/* number of candidates with slot < x (c_slot is ascending) */
__device__ __forceinline__ uint32_t cand_lower_bound(const uint16_t* c_slot, uint32_t n, uint32_t x) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (c_slot[mid] < x) lo = mid + 1u; else hi = mid;
    }
    return lo;
}

/* load the block's payload (16 B per lane) and unpack it through the wave's LDS window */
__device__ __forceinline__ DecodedPair fetch_unpack(const uint32_t* __restrict__ payload, uint32_t first, uint32_t meta, uint32_t* stage, uint32_t lane) {
    if (lane * 4u < payload_words(meta)) {
        const Words4 v = *reinterpret_cast<const Words4*>(payload + lane * 4u);
        stage[lane * 4u] = v.a; stage[lane * 4u + 1] = v.b; stage[lane * 4u + 2] = v.c; stage[lane * 4u + 3] = v.d;
    }
    wave_lds_fence();
    DecodedPair r = unpack_staged<false>(stage, first, meta, lane);
    wave_lds_fence();
    return r;
}

template <typename TabT>
__global__ __launch_bounds__(XGM_WG) void xgm_and_kernel(xgm_seg_dev seg, const xgm_dev_query* __restrict__ queries,
                                                          const xgm_work* __restrict__ work, uint32_t stripes_per_group, uint32_t tab_terms,
                                                          uint32_t cap, uint32_t k_stride, xgm_cand* __restrict__ cand_out,
                                                          xgm_group_hdr* __restrict__ ghdr_out, unsigned long long* __restrict__ phase_cycles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    /* optional phase timing (diagnostics): thread 0 accumulates s_memtime deltas per phase */
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_unit_start = __builtin_readcyclecounter();
    unsigned long long tmark = phase_cycles ? __builtin_readcyclecounter() : 0ull;
#define XGM_PHASE(i) do { if (phase_cycles) { unsigned long long n_ = __builtin_readcyclecounter(); pc[i] += n_ - tmark; tmark = n_; } } while (0)
    const xgm_work wk = work[blockIdx.x];
    const uint32_t qi = wk.qi;
    const xgm_dev_query& q = queries[qi];
    const uint32_t SB = seg.stripe_bits, W = 1u << SB, NW = W / 32u;
    const uint32_t T = q.n_terms, k = q.k, SPG = stripes_per_group;

    AndSmem sm = and_carve<TabT>(smem, W, tab_terms, cap, SPG);
    AndCtrl& ctl = *sm.ctrl;
    TabT* c_w = reinterpret_cast<TabT*>(sm.c_w);
    uint32_t* my_stage = sm.stage + wave * kAndWindows * kStageWords;
    uint32_t* rs = sm.runs;
    uint32_t* re = sm.runs + (size_t)tab_terms * SPG;

    const uint32_t n_stripes = (seg.lastdocid >> SB) + 1u;
    const uint32_t s_begin = wk.s_begin;
    const uint32_t s_end = wk.s_end;

    for (uint32_t i = tid; i < cap; i += XGM_WG) { sm.tk.w[i] = 0; sm.tk.d[i] = 0xFFFFFFFFu; sm.tk.m[i] = 0xFFFFFFFFu; }
    for (uint32_t i = tid; i < 2u * tab_terms * SPG; i += XGM_WG) sm.runs[i] = 0;
    for (uint32_t i = tid; i < NW; i += XGM_WG) { sm.bitmap[i] = 0; sm.rankw[i] = 0xFFFFFFFFu; }
    for (uint32_t i = tid; i < T * kAndCand; i += XGM_WG) c_w[i] = 0;
    if (tid == 0) { ctl.tkn = 0; ctl.theta_valid = 0; ctl.theta_w = 0; ctl.theta_d = 0; ctl.matches = 0; ctl.coarse = 0; }
    const bool empty = (q.flags & XGM_QF_EMPTY) || s_begin >= s_end || k == 0;
    if (!empty) {
        for (uint32_t t = wave; t < T; t += XGM_WAVES) {
            const uint32_t id = q.term_id[t];
            uint32_t c = 0, e = 0;
            uint64_t tb = 0;
            if (id != 0xFFFFFFFFu) {
                const uint32_t b0 = (uint32_t)seg.term_blk[id], b1 = (uint32_t)seg.term_blk[id + 1];
                c = wave_lower_bound(seg.blk_first, b0, b1, s_begin << SB, lane);
                e = (s_end >= n_stripes) ? b1 : wave_lower_bound(seg.blk_first, c, b1, s_end << SB, lane);
                tb = seg.term_word[id];
            }
            if (lane == 0) { ctl.cur[t] = c; ctl.end[t] = e; ctl.tbase[t] = tb; }
        }
    }
    __syncthreads();
    if (!empty) {
        for (uint32_t t = 0; t < T; ++t) {
            const uint32_t c = ctl.cur[t], e = ctl.end[t];
            for (uint32_t i = c + tid; i < e; i += XGM_WG) {
                const uint32_t s = (seg.blk_first[i] >> SB) - s_begin;
                const uint32_t sp = i > c ? (seg.blk_first[i - 1] >> SB) - s_begin : 0xFFFFFFFFu;
                const uint32_t sn = i + 1 < e ? (seg.blk_first[i + 1] >> SB) - s_begin : 0xFFFFFFFFu;
                if (s != sp) rs[t * SPG + s] = i;
                if (s != sn) re[t * SPG + s] = i + 1u;
            }
        }
    }
    __syncthreads();

    XGM_PHASE(0);                                                  /* init + run table */
    unsigned long long my_matches = 0;
    const uint32_t n_local = empty ? 0u : s_end - s_begin;

    /* Every wave serves every other term and takes each 4th needed block of it, so the four waves
     * finish P3 together whatever the terms' densities are. */
    auto next_active = [&](uint32_t from) {
        uint32_t x = from;
        for (; x < n_local; ++x) {
            bool all = true;
            for (uint32_t t = 0; t < T; ++t) all = all && (re[t * SPG + x] != rs[t * SPG + x]);
            if (all) break;
        }
        return x;
    };

    /* software-pipelined header registers: lane j holds block j of the run */
    uint32_t h0_meta = 0, h0_first = 0, h0_word = 0;                 /* term 0, first chunk (lanes 0..7) */
    uint32_t ha_meta = 0, ha_first = 0, ha_word = 0, ha_next = 0;    /* term 1 (lanes 0..63)             */
    uint32_t hb_meta = 0, hb_first = 0, hb_word = 0, hb_next = 0;    /* term 2, when T >= 3              */
    auto issue_headers = [&](uint32_t x) {
        const uint32_t r0 = rs[x], n0b = re[x] - r0;
        if (lane < n0b && lane < kAndChunkBlocks) {
            h0_meta = seg.blk_meta[r0 + lane]; h0_first = seg.blk_first[r0 + lane]; h0_word = seg.blk_word[r0 + lane];
        }
        {
            const uint32_t rb = rs[1u * SPG + x], nb = re[1u * SPG + x] - rb;
            if (lane < nb) {
                ha_meta = seg.blk_meta[rb + lane]; ha_first = seg.blk_first[rb + lane]; ha_word = seg.blk_word[rb + lane];
                ha_next = lane + 1u < nb ? seg.blk_first[rb + lane + 1u] : 0xFFFFFFFFu;
            }
        }
        if (T >= 3u) {
            const uint32_t rb = rs[2u * SPG + x], nb = re[2u * SPG + x] - rb;
            if (lane < nb) {
                hb_meta = seg.blk_meta[rb + lane]; hb_first = seg.blk_first[rb + lane]; hb_word = seg.blk_word[rb + lane];
                hb_next = lane + 1u < nb ? seg.blk_first[rb + lane + 1u] : 0xFFFFFFFFu;
            }
        }
    };

    uint32_t sl = next_active(0);
    if (sl < n_local) issue_headers(sl);
    while (sl < n_local) {
        if (phase_cycles) pc[7] += 1;
        const uint32_t s = s_begin + sl;
        const uint32_t stripe_base = s << SB;
        const uint32_t r0 = rs[sl], r0e = re[sl];                  /* term 0 (rarest) blocks */
        const uint32_t sl_next = next_active(sl + 1u);

        for (uint32_t cb = r0; cb < r0e; cb += kAndChunkBlocks) {
            const uint32_t nblk0 = r0e - cb < kAndChunkBlocks ? r0e - cb : kAndChunkBlocks;
            /* ---- P1: term-0 blocks of the chunk → candidates (ordinals follow docid order) ---- */
            uint32_t n_c;
            {
                uint32_t m0 = h0_meta, f0 = h0_first, w0 = h0_word;
                if (cb != r0) {                                     /* later chunks: not prefetched */
                    m0 = lane < nblk0 ? seg.blk_meta[cb + lane] : 0u;
                    f0 = lane < nblk0 ? seg.blk_first[cb + lane] : 0u;
                    w0 = lane < nblk0 ? seg.blk_word[cb + lane] : 0u;
                }
                uint32_t cnt0 = lane < nblk0 ? XGM_META_COUNT(m0) : 0u;
                uint32_t incl = wave_incl_scan(cnt0);
                n_c = __builtin_amdgcn_readlane(incl, 63);
                /* this wave's blocks: j = wave, wave + 4 (at most two): fetch both payloads first */
                const uint32_t ja = wave, jb = wave + XGM_WAVES;
                Words4 pa = {0, 0, 0, 0}, pb = {0, 0, 0, 0};
                uint32_t ma = 0, mb = 0;
                if (ja < nblk0) {
                    ma = __builtin_amdgcn_readlane(m0, ja);
                    { const uint32_t rl_ = __builtin_amdgcn_readlane(w0, ja);
                      if (lane * 4u < payload_words(ma)) pa = *reinterpret_cast<const Words4*>(seg.words + ctl.tbase[0] + rl_ + lane * 4u); }
                }
                if (jb < nblk0) {
                    mb = __builtin_amdgcn_readlane(m0, jb);
                    { const uint32_t rl_ = __builtin_amdgcn_readlane(w0, jb);
                      if (lane * 4u < payload_words(mb)) pb = *reinterpret_cast<const Words4*>(seg.words + ctl.tbase[0] + rl_ + lane * 4u); }
                }
                for (uint32_t pass = 0; pass < 2u; ++pass) {
                    const uint32_t j = pass ? jb : ja;
                    if (j >= nblk0) break;
                    const uint32_t meta = pass ? mb : ma;
                    const Words4 pv = pass ? pb : pa;
                    const uint32_t obase = __builtin_amdgcn_readlane(incl, j) - XGM_META_COUNT(meta);
                    const uint32_t first = __builtin_amdgcn_readlane(f0, j);
                    if (lane * 4u < payload_words(meta)) {
                        my_stage[lane * 4u] = pv.a; my_stage[lane * 4u + 1] = pv.b; my_stage[lane * 4u + 2] = pv.c; my_stage[lane * 4u + 3] = pv.d;
                    }
                    wave_lds_fence();
                    DecodedPair r = unpack_staged<false>(my_stage, first, meta, lane);
                    wave_lds_fence();
                    const uint32_t s0 = r.d0 - stripe_base, s1 = r.d1 - stripe_base;
                    uint32_t prev1 = (uint32_t)__shfl_up((int)s1, 1);
                    const uint32_t pw0 = lane == 0 ? 0xFFFFFFFFu : (prev1 >> 5);
                    unsigned long long cbits = 0;
                    if (r.v0) {
                        const uint32_t o = obase + 2u * lane;
                        sm.c_slot[o] = (uint16_t)s0;
                        c_w[o] = (TabT)(r.w0 + 1u);
                        atomicOr(&sm.bitmap[s0 >> 5], 1u << (s0 & 31u));
                        if ((s0 >> 5) != pw0) atomicMin(&sm.rankw[s0 >> 5], o);
                        cbits |= 1ull << (s0 >> 7);
                    }
                    if (r.v1) {
                        const uint32_t o = obase + 2u * lane + 1u;
                        sm.c_slot[o] = (uint16_t)s1;
                        c_w[o] = (TabT)(r.w1 + 1u);
                        atomicOr(&sm.bitmap[s1 >> 5], 1u << (s1 & 31u));
                        if ((s1 >> 5) != (s0 >> 5)) atomicMin(&sm.rankw[s1 >> 5], o);
                        cbits |= 1ull << (s1 >> 7);
                    }
                    /* wave-wide OR of the buckets touched by this block, one LDS atomic */
                    uint32_t clo = (uint32_t)cbits, chi = (uint32_t)(cbits >> 32);
                    for (int sh = 32; sh > 0; sh >>= 1) { clo |= (uint32_t)__shfl_xor((int)clo, sh); chi |= (uint32_t)__shfl_xor((int)chi, sh); }
                    if (lane == 0) atomicOr(&ctl.coarse, ((unsigned long long)chi << 32) | clo);
                }
            }
            XGM_PHASE(1);                                          /* P1 work */
            __syncthreads();
            XGM_PHASE(2);

            /* doclen of every candidate: requested now, consumed in P4 after the decodes */
            uint32_t dl[kAndCand / XGM_WG];
#pragma unroll
            for (uint32_t c = 0; c < kAndCand / XGM_WG; ++c) {
                const uint32_t o = tid + c * XGM_WG;
                dl[c] = o < n_c ? seg.doclen[stripe_base + sm.c_slot[o]] : 0u;
            }
            const unsigned long long coarse = ctl.coarse;

            /* ---- P3: the other terms: decode only blocks whose docid range may contain a candidate.
             * Terms are taken two at a time so that the payload loads of both are in flight together;
             * every wave takes each 4th needed block of each term. ---- */
            auto bucket_need = [&](uint32_t first, uint32_t nfirst) {
                const uint32_t lo = (first - stripe_base) >> 7;
                const uint32_t hi = ((nfirst == 0xFFFFFFFFu ? W : nfirst - stripe_base) - 1u) >> 7;
                const unsigned long long m = (hi >= 63u ? ~0ull : ((1ull << (hi + 1u)) - 1ull)) & ~((1ull << lo) - 1ull);
                return (coarse & m) != 0ull;
            };
            for (uint32_t ta = 1; ta < T; ta += 2u) {
                const uint32_t tb = ta + 1u;
                const bool have_b = tb < T;
                uint32_t a_meta, a_first, a_word, a_next, b_meta = 0, b_first = 0, b_word = 0, b_next = 0xFFFFFFFFu;
                const uint32_t nba = re[ta * SPG + sl] - rs[ta * SPG + sl];
                const uint32_t nbb = have_b ? re[tb * SPG + sl] - rs[tb * SPG + sl] : 0u;
                if (ta == 1u) {
                    a_meta = ha_meta; a_first = ha_first; a_word = ha_word; a_next = ha_next;
                    b_meta = hb_meta; b_first = hb_first; b_word = hb_word; b_next = hb_next;
                } else {                                            /* 4th term onwards: not prefetched */
                    a_meta = a_first = a_word = 0; a_next = 0xFFFFFFFFu;
                    const uint32_t rba = rs[ta * SPG + sl];
                    if (lane < nba) {
                        a_meta = seg.blk_meta[rba + lane]; a_first = seg.blk_first[rba + lane]; a_word = seg.blk_word[rba + lane];
                        a_next = lane + 1u < nba ? seg.blk_first[rba + lane + 1u] : 0xFFFFFFFFu;
                    }
                    if (have_b) {
                        const uint32_t rbb = rs[tb * SPG + sl];
                        if (lane < nbb) {
                            b_meta = seg.blk_meta[rbb + lane]; b_first = seg.blk_first[rbb + lane]; b_word = seg.blk_word[rbb + lane];
                            b_next = lane + 1u < nbb ? seg.blk_first[rbb + lane + 1u] : 0xFFFFFFFFu;
                        }
                    }
                }
                uint64_t mask_a = __ballot(lane < nba && bucket_need(a_first, a_next));
                uint64_t mask_b = __ballot(have_b && lane < nbb && bucket_need(b_first, b_next));
                for (uint32_t x = 0; x < wave; ++x) { mask_a &= mask_a - 1u; mask_b &= mask_b - 1u; }      /* this wave's share */
                TabT* row_a = c_w + (size_t)ta * kAndCand;
                TabT* row_b = c_w + (size_t)tb * kAndCand;
                while (mask_a | mask_b) {
                    /* up to 4 + 4 of my blocks: all payload loads first, then the decodes */
                    uint32_t jj[8], n_a = 0, n_b = 0;
                    Words4 pv[8];
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) {
                        jj[u] = 0; pv[u] = Words4{0, 0, 0, 0};
                        if (mask_a) {
                            jj[u] = (uint32_t)__builtin_ctzll(mask_a);
                            for (uint32_t x = 0; x < XGM_WAVES && mask_a; ++x) mask_a &= mask_a - 1u;
                            const uint32_t bm = __builtin_amdgcn_readlane(a_meta, jj[u]);
                            { const uint32_t rl_ = __builtin_amdgcn_readlane(a_word, jj[u]);
                              if (lane * 4u < payload_words(bm)) pv[u] = *reinterpret_cast<const Words4*>(seg.words + ctl.tbase[ta] + rl_ + lane * 4u); }
                            n_a = u + 1u;
                        }
                    }
#pragma unroll
                    for (uint32_t u = 4; u < 8u; ++u) {
                        jj[u] = 0; pv[u] = Words4{0, 0, 0, 0};
                        if (mask_b) {
                            jj[u] = (uint32_t)__builtin_ctzll(mask_b);
                            for (uint32_t x = 0; x < XGM_WAVES && mask_b; ++x) mask_b &= mask_b - 1u;
                            const uint32_t bm = __builtin_amdgcn_readlane(b_meta, jj[u]);
                            { const uint32_t rl_ = __builtin_amdgcn_readlane(b_word, jj[u]);
                              if (lane * 4u < payload_words(bm)) pv[u] = *reinterpret_cast<const Words4*>(seg.words + ctl.tbase[tb] + rl_ + lane * 4u); }
                            n_b = u - 3u;
                        }
                    }
#pragma unroll
                    for (uint32_t u = 0; u < 8u; ++u) {
                        const bool is_b = u >= 4u;
                        if (is_b ? (u - 4u < n_b) : (u < n_a)) {
                            const uint32_t bmeta = __builtin_amdgcn_readlane(is_b ? b_meta : a_meta, jj[u]);
                            const uint32_t bfirst = __builtin_amdgcn_readlane(is_b ? b_first : a_first, jj[u]);
                            TabT* row = is_b ? row_b : row_a;
                            if (lane * 4u < payload_words(bmeta)) {
                                my_stage[lane * 4u] = pv[u].a; my_stage[lane * 4u + 1] = pv[u].b; my_stage[lane * 4u + 2] = pv[u].c; my_stage[lane * 4u + 3] = pv[u].d;
                            }
                            wave_lds_fence();
                            DecodedPair r = unpack_staged<false>(my_stage, bfirst, bmeta, lane);
                            wave_lds_fence();
                            if (r.v0) {
                                const uint32_t sl0 = r.d0 - stripe_base, wd = sl0 >> 5, bit = sl0 & 31u;
                                const uint32_t bm = sm.bitmap[wd];
                                if ((bm >> bit) & 1u) row[sm.rankw[wd] + (uint32_t)__popc(bm & ((1u << bit) - 1u))] = (TabT)(r.w0 + 1u);
                            }
                            if (r.v1) {
                                const uint32_t sl1 = r.d1 - stripe_base, wd = sl1 >> 5, bit = sl1 & 31u;
                                const uint32_t bm = sm.bitmap[wd];
                                if ((bm >> bit) & 1u) row[sm.rankw[wd] + (uint32_t)__popc(bm & ((1u << bit) - 1u))] = (TabT)(r.w1 + 1u);
                            }
                        }
                    }
                }
            }
            XGM_PHASE(3);
            __syncthreads();
            XGM_PHASE(4);

            /* headers of the next active stripe: in flight while this one is scored */
            if (cb + kAndChunkBlocks >= r0e && sl_next < n_local) issue_headers(sl_next);

            /* ---- P4: candidates present in every term are matches: BM25 + top-k ---- */
            for (uint32_t i0 = 0; i0 < n_c; i0 += XGM_WG) {
                const uint32_t fill_now = ctl.tkn;
                __syncthreads();
                if (fill_now + XGM_WG > cap) {
                    topk_sort(sm.tk, tid);
                    if (tid == 0) {
                        uint32_t keep = ctl.tkn < k ? ctl.tkn : k;
                        ctl.tkn = keep;
                        if (keep == k) { ctl.theta_valid = 1; ctl.theta_w = sm.tk.w[k - 1]; ctl.theta_d = sm.tk.d[k - 1]; }
                    }
                    __syncthreads();
                    for (uint32_t i = ctl.tkn + tid; i < cap; i += XGM_WG) { sm.tk.w[i] = 0; sm.tk.d[i] = 0xFFFFFFFFu; sm.tk.m[i] = 0xFFFFFFFFu; }
                    __syncthreads();
                }
                const uint32_t o = i0 + tid;
                if (o < n_c) {
                    bool pass = true;
                    for (uint32_t t = 1; t < T; ++t) pass = pass && c_w[(size_t)t * kAndCand + o] != 0;
                    if (pass) {
                        ++my_matches;
                        const uint32_t did = stripe_base + sm.c_slot[o];
                        /* BM25Weight::get_sumpart, bm25weight.cc:170-181 — same operations, same order */
                        uint32_t dlen = dl[0];
#pragma unroll
                        for (uint32_t c = 1; c < kAndCand / XGM_WG; ++c) dlen = (i0 == c * XGM_WG) ? dl[c] : dlen;
                        const double len = (double)dlen;
                        double normlen = len * q.len_factor;
                        normlen = normlen > q.min_normlen ? normlen : q.min_normlen;
                        const double denom_len = q.k1 * (normlen * q.b + (1.0 - q.b));
                        /* MultiAndPostList::get_weight: ((0 + w0) + w1) + ... in plan order */
                        double weight = 0.0;
                        for (uint32_t t = 0; t < T; ++t) {
                            const double wdf = (double)((uint32_t)c_w[(size_t)t * kAndCand + o] - 1u);
                            const double denom = denom_len + wdf;
                            weight = weight + q.termweight[t] * (wdf / denom);
                        }
                        const uint64_t wb = (uint64_t)__double_as_longlong(weight);
                        const bool take = !ctl.theta_valid || cand_before(wb, did, ctl.theta_w, ctl.theta_d);
                        if (take) {
                            const uint32_t p = atomicAdd(&ctl.tkn, 1u);
                            sm.tk.w[p] = wb; sm.tk.d[p] = did; sm.tk.m[p] = (uint32_t)__popc(q.score_mask);
                        }
                    }
                    for (uint32_t t = 1; t < T; ++t) c_w[(size_t)t * kAndCand + o] = 0;
                }
                __syncthreads();
            }
            for (uint32_t i = tid; i < NW; i += XGM_WG) { sm.bitmap[i] = 0; sm.rankw[i] = 0xFFFFFFFFu; }
            if (tid == 0) ctl.coarse = 0;
            __syncthreads();
            XGM_PHASE(5);
        }
        sl = sl_next;
    }
    if (phase_cycles && tid == 0) { for (int i = 0; i < 8; ++i) atomicAdd(&phase_cycles[i], pc[i]); }
#undef XGM_PHASE

    /* ---- group epilogue ---- */
    __syncthreads();
    topk_sort(sm.tk, tid);
    if (my_matches) atomicAdd(&ctl.matches, my_matches);
    __syncthreads();
    const uint32_t n_out = ctl.tkn < k ? ctl.tkn : k;
    xgm_cand* out = cand_out + (size_t)wk.slot * k_stride;
    for (uint32_t i = tid; i < n_out; i += XGM_WG) {
        xgm_cand c;
        c.wbits = sm.tk.w[i]; c.did = sm.tk.d[i]; c.subqs = sm.tk.m[i];
        out[i] = c;
    }
    if (tid == 0) {
        xgm_group_hdr h = {};
        h.matches = ctl.matches; h.n_cand = n_out; h.pad = 0;
        h.t_start = t_unit_start; h.t_end = __builtin_readcyclecounter();
        ghdr_out[wk.slot] = h;
    }
}

/* ---------------------------------------------------------------- wave-autonomous AND kernel ---- */

/* Same algorithm as xgm_and_kernel, different decomposition: ONE WAVE owns a work unit (a query and a
 * short stripe range) and runs it start to finish out of a private LDS slice — no workgroup barrier
 * anywhere, so a CU keeps ~20 independent instruction streams in flight instead of 4-6 barrier-coupled
 * groups, and the per-stripe latency chain of one unit is covered by the others.  Wave-uniform state
 * (top-k fill, threshold, coarse mask) lives in registers.  Used when first+maxitems <= kAndwMaxK. */
constexpr uint32_t kAndwCandPlain = 512;             /* candidates per chunk = 4 blocks of term 0 */
constexpr uint32_t kAndwCandPhrase = 128;            /* with the position offsets and the K6 staging area: 1 block (3 workgroups per CU at 3 terms) */
#ifndef XGM_ANDW_WAVES
#define XGM_ANDW_WAVES 4           /* min waves per SIMD the register allocator must allow */
#endif
#ifndef XGM_PHRASE_WAVES
#define XGM_PHRASE_WAVES 3         /* ... of the positional instantiation (A/B: tools/ab_build.sh) */
#endif                  /* top-k buffer cap 256 */

__host__ __device__ inline size_t andw_wave_bytes(uint32_t W, uint32_t T, uint32_t cap, size_t tab_elem, uint32_t spg, bool phrase, bool sided) {
    const uint32_t kAndwCand = phrase ? kAndwCandPhrase : kAndwCandPlain;
    size_t off = 0;
    off += (size_t)cap * 8;                                    /* tk_w */
    off += (size_t)cap * 4;                                    /* tk_d */
    off += (size_t)kStageWords * 4;                            /* stage */
    off += (size_t)(W / 32u > 128u ? W / 32u : 128u) * 4;      /* bitmap (queue path: the match stack's 128 docids) */
    off += (size_t)2 * T * spg * 4;                            /* runs */
    off += (size_t)(W / 32u > 256u ? W / 32u : 256u) * 2;      /* rankw (u16; queue path: 256 wdf entries) */
    off += (size_t)kAndwCand * 2;                              /* c_slot */
    off += (size_t)T * kAndwCand * tab_elem;                   /* c_w */
    off += phrase ? (size_t)T * kAndwCand * 4 : 0;             /* c_pos: position-list offset per candidate and term */
    off += phrase ? (size_t)T * kPosFast * 64 * 2 : 0;         /* lpos: the round's positions, u16 [T][kPosFast][64 lanes] */
    off += sided ? (size_t)cap : 0;                            /* tk_m: weighted subqueries matched, per top-k entry */
    /* the bodies for all-container / long-tail-led queries (xgm_dense_unit, xgm_flat_unit) run out of the first bytes of the same slice:
     * top-k 1.5 KiB + offsets 0.5 KiB + the dense body's ring 1 KiB (positional: + survivor queue <= 4 KiB + T x 2 KiB of staged positions) —
     * static_asserts below */
    const size_t body = phrase ? 6144 + (size_t)T * 2048 : 3072;
    if (off < body) off = body;
    return (off + 15) & ~(size_t)15;
}

/* the same with a byte of payload per entry (weighted subqueries matched) */
__device__ void wave_topk_sort_m(uint64_t* w_, uint32_t* d_, uint8_t* m_, uint32_t cap, uint32_t lane) {
    XGM_AS_LDS uint64_t* w = (XGM_AS_LDS uint64_t*)w_;
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

typedef double andw_d8 __attribute__((ext_vector_type(8)));

/* bitonic sort of cap (power of two, >= 128) candidates by one wave; best first */
__device__ void wave_topk_sort(uint64_t* w_, uint32_t* d_, uint32_t cap, uint32_t lane) {
    XGM_AS_LDS uint64_t* w = (XGM_AS_LDS uint64_t*)w_;                               /* (the wave's top-k buffer lives in its LDS slice) */
    XGM_AS_LDS uint32_t* d = (XGM_AS_LDS uint32_t*)d_;
    for (uint32_t size = 2; size <= cap; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            wave_lds_fence();
            for (uint32_t i = lane; i < (cap >> 1); i += 64u) {
                const uint32_t lo = 2u * i - (i & (stride - 1u)), hi = lo + stride;
                const bool asc = ((lo & size) == 0);
                const uint64_t aw = w[lo], bw = w[hi];
                const uint32_t ad = d[lo], bd = d[hi];
                const bool swap = asc ? cand_before(bw, bd, aw, ad) : cand_before(aw, ad, bw, bd);
                if (swap) { w[lo] = bw; w[hi] = aw; d[lo] = bd; d[hi] = ad; }
            }
        }
    }
    wave_lds_fence();
}

#include "xgm_unit_finish.h"
#include "xgm_dense_body.inc"
#include "xgm_flat_body.inc"
static_assert(dense_wave_bytes(false) <= 3072 && flat_wave_bytes(false, 0) <= 3072, "andw_wave_bytes reserves 3 KiB for the plain bodies");
static_assert(dense_wave_bytes(true, 4) <= 6144 + 4 * 2048 && flat_wave_bytes(true, 4) <= 6144 + 4 * 2048 &&
              dense_wave_bytes(true, 2) <= 6144 + 2 * 2048 && flat_wave_bytes(true, 2) <= 6144 + 2 * 2048,
              "andw_wave_bytes reserves 6 KiB + T x 2 KiB for the positional bodies");

/* The reference-identical batch mode of positional queries (include/xgm.h, XGM_REPLAY_BATCH_FROZEN): the same work units, the same two bodies,
 * compiled in their LIST form — a unit reports its FIRST 2 (k + 1) matches in docid order with their successors in the conjunction
 * (xgm_prefix_entry) instead of its best k, tests every document of the conjunction (no threshold) and stops as soon as the list is full.
 * xgm_frozen_finish_kernel (xgm_frozen.hip) then walks each query's lists as ProtoMSet + SelectPostList would (protomset.h:340-400,
 * selectpostlist.cc:28-55).  Queries with neither body (terms without containers and flat arrays, more than 4 terms, k > 64) are declined
 * per unit: the host answers them with the per-query replay. */
template <bool TALLY>
__global__ __launch_bounds__(XGM_WG, XGM_PHRASE_WAVES) void xgm_andw_list_kernel(xgm_seg_dev seg, const xgm_dev_query* __restrict__ queries,
                                                                                const xgm_work* __restrict__ work, uint32_t n_work, uint32_t spg_max,
                                                                                uint32_t tab_terms, uint32_t cap, uint32_t k_stride,
                                                                                xgm_cand* __restrict__ cand_out, xgm_group_hdr* __restrict__ ghdr_out,
                                                                                uint32_t* __restrict__ unit_matches, const xgm_fuse* __restrict__ fuse) {
    /* unit_matches [n_work] zeroed: matches found so far per unit slot — a unit stops once the units of lower stripes hold 2 (k + 1) (PrefixList::look_back);
     * fuse: only goff (the first unit slot of every query) */
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = rfl32(threadIdx.x >> 6);
    const uint32_t unit = blockIdx.x * XGM_WAVES + wave;
    if (unit >= n_work) return;                                    /* no barriers below */
    const xgm_work wk = work[unit];
    const uint32_t flags = rfl32(queries[wk.qi].flags);
    const uint32_t W = 1u << seg.stripe_bits;
    unsigned char* base = smem + (size_t)wave * andw_wave_bytes(W, tab_terms, cap, 1u, spg_max, true, false);
    if (flags & XGM_QF_DENSE) { xgm_dense_unit<true, TALLY, true>(seg, queries, wk, base, lane, k_stride, cand_out, ghdr_out, unit_matches, fuse); return; }
    if (flags & XGM_QF_FLAT) { xgm_flat_unit<true, TALLY, true>(seg, queries, wk, base, lane, k_stride, cand_out, ghdr_out, unit_matches, fuse); return; }
    if (lane == 0u) {
        xgm_group_hdr h = {};
        h.pad = XGM_PFX_DECLINED;
        ghdr_out[wk.slot] = h;
    }
}

/* XGM_REPLAY_BATCH_COUNT on plain conjunctions (include/xgm.h): the same units and bodies, which also list EVERY match — docid, weight — in docid order
 * (xgm_device.h, xgm_all_out); xgm_count.hip then replays ProtoMSet's collation over the lists: an exclusive scan of the units' top-k lists gives every
 * unit the state the reference's walk has on arrival, one wave per unit counts what ProtoMSet::add would be shown (protomset.h:340-400). */
__global__ __launch_bounds__(XGM_WG, XGM_ANDW_WAVES) void xgm_andw_all_kernel(xgm_seg_dev seg, const xgm_dev_query* __restrict__ queries,
                                                                             const xgm_work* __restrict__ work, uint32_t n_work, uint32_t spg_max,
                                                                             uint32_t tab_terms, uint32_t cap, uint32_t k_stride,
                                                                             xgm_cand* __restrict__ cand_out, xgm_group_hdr* __restrict__ ghdr_out, xgm_all_out all_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = rfl32(threadIdx.x >> 6);
    const uint32_t unit = blockIdx.x * XGM_WAVES + wave;
    if (unit >= n_work) return;                                    /* no barriers below */
    const xgm_work wk = work[unit];
    const uint32_t flags = rfl32(queries[wk.qi].flags);
    const uint32_t W = 1u << seg.stripe_bits;
    unsigned char* base = smem + (size_t)wave * andw_wave_bytes(W, tab_terms, cap, 1u, spg_max, false, false);
    if (flags & XGM_QF_DENSE) { xgm_dense_unit<false, false, false, true>(seg, queries, wk, base, lane, k_stride, cand_out, ghdr_out, nullptr, nullptr, &all_out); return; }
    if (flags & XGM_QF_FLAT) { xgm_flat_unit<false, false, false, true>(seg, queries, wk, base, lane, k_stride, cand_out, ghdr_out, nullptr, nullptr, &all_out); return; }
    if (lane == 0u) {
        xgm_group_hdr h = {};
        h.c_pad[1] = XGM_ALL_DECLINED;
        ghdr_out[wk.slot] = h;
    }
}

/* SIDED: 1 = the batch holds AND_NOT queries (excluded terms after the required ones), 2 = also AND_MAYBE
 * (optional terms: weight by the query's summation program, per-document subquery counts).  Separate
 * instantiations, so that the plain conjunction pays nothing for them. */
template <typename TabT, bool PHRASE, int SIDED, bool TALLY>
__global__ __launch_bounds__(XGM_WG, PHRASE ? XGM_PHRASE_WAVES : XGM_ANDW_WAVES) void xgm_andw_kernel(xgm_seg_dev seg, const xgm_dev_query* __restrict__ queries,
                                                              const xgm_work* __restrict__ work, uint32_t n_work, uint32_t spg_max,
                                                              uint32_t tab_terms, uint32_t cap, uint32_t k_stride,
                                                              xgm_cand* __restrict__ cand_out, xgm_group_hdr* __restrict__ ghdr_out, uint32_t* __restrict__ hist_all,
                                                              const xgm_fuse* __restrict__ fuse) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t CAND = PHRASE ? kAndwCandPhrase : kAndwCandPlain;     /* candidates per chunk */
    constexpr uint32_t CHUNKB = CAND / XGM_BLOCK;                             /* = blocks of term 0 per chunk */
    const uint32_t lane = threadIdx.x & 63u, wave = rfl32(threadIdx.x >> 6);  /* (a scalar: the unit, the query and the loops over its terms with it — xgm_or.hip, round 6) */
    const uint32_t unit = blockIdx.x * XGM_WAVES + wave;
    if (unit >= n_work) return;                                    /* no barriers below: early exit is safe */
    const xgm_work wk = work[unit];
    const xgm_dev_query& q = queries[wk.qi];
    const uint32_t SB = seg.stripe_bits, W = 1u << SB, NW = W / 32u;
    const uint32_t T = q.n_terms, k = q.k, SPG = spg_max;
    if (SIDED == 0 && (rfl32(q.flags) & XGM_QF_DENSE)) {
        /* every term has containers: the body written for that case alone (same launch, same outputs) — plain conjunctions and,
         * since round 4, positional queries that prune by weight (C5's frequent-term phrases) */
        xgm_dense_unit<PHRASE, TALLY>(seg, queries, wk, smem + (size_t)wave * andw_wave_bytes(W, tab_terms, cap, sizeof(TabT), SPG, PHRASE, SIDED == 2), lane, k_stride,
                                      cand_out, ghdr_out, PHRASE ? hist_all : nullptr, fuse);
        return;
    }
    if (SIDED == 0 && (rfl32(q.flags) & XGM_QF_FLAT)) {
        /* led by a long-tail term: its flat posting array is streamed 64 postings per round (same launch, same outputs) — plain conjunctions
         * and positional queries that prune by weight */
        xgm_flat_unit<PHRASE, TALLY>(seg, queries, wk, smem + (size_t)wave * andw_wave_bytes(W, tab_terms, cap, sizeof(TabT), SPG, PHRASE, SIDED == 2), lane, k_stride,
                                     cand_out, ghdr_out, PHRASE ? hist_all : nullptr, fuse);
        return;
    }
    /* plan positions [0, TR) must index a document; [TR, T) are the right-hand side of an AND_NOT (must not
     * index it: AndNotPostList) or of an AND_MAYBE (add their weight where they do: AndMaybePostList).  A plain
     * conjunction / FILTER has TR == T. */
    const uint32_t TR = SIDED ? q.n_req : T;
    const unsigned long long t_unit_start = __builtin_readcyclecounter();
    /* traffic tallies (xgm_group_hdr): wave-uniform, kept in scalar registers */
    uint32_t cn_bmpw = 0, cn_probe = 0, cn_blkw = 0, cn_hdr = 0, cn_dl = 0, cn_aux = 0, cn_probe_raw = 0, cn_dl_raw = 0;
    unsigned long long cn_pos = 0;                                 /* PHRASE only, per lane */
    uint32_t q_cands = 0;                                          /* TALLY: candidates the queue path produced */
    unsigned long long ph_a = 0, ph_b = 0;                         /* TALLY: cycles spent in the queue path's probe / weigh rounds */
    /* lanes hold ascending keys: how many distinct (key >> sh) values = memory sectors does one gather round touch? */
    auto tally_sectors = [&](bool valid, uint32_t key, uint32_t sh) {
        const uint32_t prev = (uint32_t)__shfl_up((int)key, 1);
        return (uint32_t)__popcll(__ballot(valid && (lane == 0u || (prev >> sh) != (key >> sh))));
    };
#define XGM_SU(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))

    /* private LDS slice */
    unsigned char* base = smem + (size_t)wave * andw_wave_bytes(W, tab_terms, cap, sizeof(TabT), SPG, PHRASE, SIDED == 2);
    size_t off = 0;
    uint64_t* tk_w = reinterpret_cast<uint64_t*>(base + off); off += (size_t)cap * 8;
    uint32_t* tk_d = reinterpret_cast<uint32_t*>(base + off); off += (size_t)cap * 4;
    uint32_t* stage = reinterpret_cast<uint32_t*>(base + off); off += (size_t)kStageWords * 4;
    uint32_t* bitmap = reinterpret_cast<uint32_t*>(base + off); off += (size_t)(NW > 128u ? NW : 128u) * 4;
    uint32_t* rs = reinterpret_cast<uint32_t*>(base + off); off += (size_t)tab_terms * SPG * 4;
    uint32_t* re = reinterpret_cast<uint32_t*>(base + off); off += (size_t)tab_terms * SPG * 4;
    uint16_t* rankw = reinterpret_cast<uint16_t*>(base + off); off += (size_t)(NW > 256u ? NW : 256u) * 2;
    uint16_t* c_slot = reinterpret_cast<uint16_t*>(base + off); off += (size_t)CAND * 2;
    TabT* c_w = reinterpret_cast<TabT*>(base + off); off += (size_t)tab_terms * CAND * sizeof(TabT);
    uint32_t* c_pos = reinterpret_cast<uint32_t*>(base + off);     /* PHRASE only */
    uint16_t* lpos = reinterpret_cast<uint16_t*>(base + off + (PHRASE ? (size_t)tab_terms * CAND * 4 : 0));   /* PHRASE only: K6 staging */
    constexpr bool MAYBE = SIDED == 2;
    uint8_t* tk_m = reinterpret_cast<uint8_t*>(base + off);        /* MAYBE only (never with PHRASE) */
    /* SIDED: the in-place summation program (<= 8 terms) in scalar registers */
    uint64_t prog_a = 0, prog_b = 0;
    uint32_t prog_root = 0;
    if (MAYBE) {
        const uint32_t* ipa32 = reinterpret_cast<const uint32_t*>(q.ip_a);
        const uint32_t* ipb32 = reinterpret_cast<const uint32_t*>(q.ip_b);
        prog_a = ((uint64_t)rfl32(ipa32[1]) << 32) | rfl32(ipa32[0]);
        prog_b = ((uint64_t)rfl32(ipb32[1]) << 32) | rfl32(ipb32[0]);
        prog_root = __builtin_amdgcn_readfirstlane(q.ip_root);
    }
    const bool phrase = PHRASE && (q.flags & XGM_QF_PHRASE);
    const bool pos_prune = PHRASE && (q.flags & XGM_QF_POSPRUNE);   /* weigh before testing positions: the match count may be a lower bound */
    bool pos_pruned = false;                                        /* per lane: a candidate was dropped that way */
    /* ... and the units of a query share what they learn: every positional match that is taken is counted in a 256-bucket
     * histogram of weight bit patterns (32 buckets per octave below the largest possible weight, global atomics — matches that
     * pass the pruning are rare); the highest bucket with >= k matches at or above it bounds the final k-th weight from below
     * although no single unit may hold k matches (phrases whose matches are rare). */
    uint32_t* hist_g = nullptr;
    int hbase = 0;
    uint64_t theta_glob = 0;
    if (pos_prune && hist_all && !(q.flags & XGM_QF_EMPTY)) {
        double mp = 0.0;
        for (uint32_t t = 0; t < q.n_terms; ++t) mp += q.ub[t];
        mp *= 1.000000001;
        hbase = (int)rfl32((uint32_t)((uint64_t)__double_as_longlong(mp) >> 47)) - (int)(XGM_OR_HIST - 1u);
        if (hbase > 0) hist_g = hist_all + (size_t)wk.qi * XGM_OR_HIST;
    }
    auto look_at_histogram = [&]() {
        uint32_t hc[4];
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i) hc[i] = __hip_atomic_load(&hist_g[lane * 4u + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t s4 = hc[0] + hc[1] + hc[2] + hc[3];
        const uint32_t P = wave_incl_scan(s4);
        const uint32_t suf = rl32(P, 63) - P + s4;                   /* matches in buckets >= 4 * lane */
        const uint64_t okm = __ballot(suf >= q.k);
        if (okm) {
            const uint32_t Lh = 63u - (uint32_t)__builtin_clzll(okm);
            const uint32_t cum = rl32(suf, Lh) - rl32(s4, Lh);
            const uint32_t c3 = rl32(hc[3], Lh), c2 = rl32(hc[2], Lh), c1 = rl32(hc[1], Lh);
            uint32_t bsel = 4u * Lh;
            if (cum + c3 >= q.k) bsel = 4u * Lh + 3u;
            else if (cum + c3 + c2 >= q.k) bsel = 4u * Lh + 2u;
            else if (cum + c3 + c2 + c1 >= q.k) bsel = 4u * Lh + 1u;
            if (bsel > 0u) {
                const uint64_t tb = (uint64_t)((uint32_t)hbase + bsel) << 47;
                theta_glob = tb > theta_glob ? tb : theta_glob;
            }
        }
    };

    const uint32_t n_stripes = (seg.lastdocid >> SB) + 1u;
    const uint32_t s_begin = wk.s_begin, s_end = wk.s_end;
    const bool empty = (q.flags & XGM_QF_EMPTY) || s_begin >= s_end || k == 0;

    for (uint32_t i = lane; i < cap; i += 64u) { tk_w[i] = 0; tk_d[i] = 0xFFFFFFFFu; if (MAYBE) tk_m[i] = 0; }
    for (uint32_t i = lane; i < 2u * tab_terms * SPG; i += 64u) rs[i] = 0;        /* rs and re are adjacent */
    for (uint32_t i = lane; i < NW; i += 64u) { bitmap[i] = 0; rankw[i] = 0xFFFFu; }
    for (uint32_t i = lane; i < T * CAND; i += 64u) c_w[i] = 0;
    wave_lds_fence();

    /* block ranges of every term inside the unit's docid range → run table; lane t keeps term t's
     * payload base and dense-container index */
    uint64_t tbase_reg = 0, tpos_reg = 0;
    uint32_t tflag_reg = 0;                                         /* lane t: term t's flags (PHRASE: position width) */
    uint32_t dense_reg = 0xFFFFFFFFu;
    bool have_reg = false;                                         /* lane t: term t exists in this shard */
    if (!empty) {
        for (uint32_t t = 0; t < T; ++t) {
            const uint32_t id = q.term_id[t];
            if (SIDED && id == 0xFFFFFFFFu) continue;              /* an excluded term the shard does not have */
            if (lane == t) {
                have_reg = true;
                tbase_reg = seg.term_word[id];
                if (PHRASE) { tpos_reg = seg.term_pos[id]; tflag_reg = seg.term_flags[id]; }
                /* the positional filter needs every term's position offsets: the block decode yields them, and so do
                 * containers that carry the buckets' position bases (seg.dense_pos) */
                if ((!PHRASE || seg.dense_pos) && sizeof(TabT) == 1 && seg.dense_id) dense_reg = seg.dense_id[id];
            }
        }
    }
    /* Plan order is ascending termfreq and "dense" is a termfreq threshold, so the dense terms are a
     * suffix [td, T).  td == 0: every term is dense → candidates come from the AND of the bitmaps.
     * (With more than 4 terms term 0 is always decoded, to bound the registers of that path.) */
    uint32_t td = (uint32_t)__popcll(__ballot(lane < TR && dense_reg == 0xFFFFFFFFu));
    /* right-hand terms without containers are block-decoded against the candidates' bitmap + rank table (P3c): the
     * decode path builds them while it enumerates term 0, the all-dense path writes them from the AND of the bitmaps */
    const uint64_t sparse_neg = SIDED ? __ballot(lane >= TR && lane < T && have_reg && dense_reg == 0xFFFFFFFFu) : 0ull;   /* right-hand terms without containers */
    if (td == 0 && TR > 4u) td = 1;
    /* the queue path (below): every term that takes part has containers and there is no positional filter — the unit
     * never looks at a block, so the table holds the containers' offsets instead of block ranges */
    const bool qpath = !PHRASE && !empty && td <= 1u && sparse_neg == 0ull && T <= 8u;
    uint32_t q_b0 = 0u, q_b1 = 0u;                                 /* td == 1: term 0's blocks inside the unit's docid range */
    if (qpath) {
        for (uint32_t t = td; t < T; ++t) {
            const uint32_t d = __builtin_amdgcn_readlane(dense_reg, t);
            if (d == 0xFFFFFFFFu) continue;                        /* a right-hand term the shard does not have */
            if (TALLY) { cn_aux += s_end - s_begin; }
            for (uint32_t x = lane; x < s_end - s_begin; x += 64u) rs[t * SPG + x] = seg.dense_dir[(size_t)d * seg.n_stripes + (s_begin + x)];
        }
        if (td == 1u) {
            const uint32_t id = q.term_id[0];
            const uint32_t b0 = (uint32_t)seg.term_blk[id], b1 = (uint32_t)seg.term_blk[id + 1];
            q_b0 = wave_lower_bound(seg.blk_first, b0, b1, s_begin << SB, lane);
            q_b1 = (s_end >= n_stripes) ? b1 : wave_lower_bound(seg.blk_first, q_b0, b1, s_end << SB, lane);
        }
    } else if (!empty) {
        for (uint32_t t = 0; t < T; ++t) {
            const uint32_t id = q.term_id[t];
            if (SIDED && id == 0xFFFFFFFFu) continue;
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
    }
    wave_lds_fence();
    auto tbase = [&](uint32_t t) {
        return rl64(tbase_reg, t);
    };
    uint32_t tkn = 0;                                              /* wave-uniform top-k state */
    bool theta_valid = false;
    uint64_t theta_w = 0;
    uint32_t theta_d = 0;
    unsigned long long matches = 0;                                /* per lane, reduced at the end */
    const uint32_t n_local = empty ? 0u : s_end - s_begin;

    auto next_active = [&](uint32_t from) {
        uint32_t x = from;
        for (; x < n_local; ++x) {
            bool all = true;
            for (uint32_t t = 0; t < TR; ++t) all = all && (re[t * SPG + x] != rs[t * SPG + x]);
            if (all) break;
        }
        return x;
    };

    /* software-pipelined per-stripe registers: block headers of the sparse terms (lane j = block j of
     * the run) and, in lane t, the container offset of dense term t */
    uint32_t h0_meta = 0, h0_first = 0, h0_word = 0, h0_pos = 0;
    uint32_t ha_meta = 0, ha_first = 0, ha_word = 0, ha_next = 0, ha_pos = 0;
    uint32_t hb_meta = 0, hb_first = 0, hb_word = 0, hb_next = 0, hb_pos = 0;
    uint32_t hc_off = 0;
    auto issue_headers = [&](uint32_t x) {
        if (td > 0u) {
            const uint32_t r0 = rs[x], n0b = re[x] - r0;
            if (TALLY) { cn_hdr += XGM_SU(n0b < CHUNKB ? n0b : CHUNKB); }
            if (lane < n0b && lane < CHUNKB) {
                h0_meta = seg.blk_meta[r0 + lane]; h0_first = seg.blk_first[r0 + lane]; h0_word = seg.blk_word[r0 + lane];
                if (PHRASE) h0_pos = seg.blk_pos[r0 + lane];
            }
        }
        if (td > 1u) {
            const uint32_t rb = rs[1u * SPG + x], nb = re[1u * SPG + x] - rb;
            if (TALLY) { cn_hdr += XGM_SU(nb); }
            if (lane < nb) {
                ha_meta = seg.blk_meta[rb + lane]; ha_first = seg.blk_first[rb + lane]; ha_word = seg.blk_word[rb + lane];
                ha_next = lane + 1u < nb ? seg.blk_first[rb + lane + 1u] : 0xFFFFFFFFu;
                if (PHRASE) ha_pos = seg.blk_pos[rb + lane];
            }
        }
        if (td > 2u) {
            const uint32_t rb = rs[2u * SPG + x], nb = re[2u * SPG + x] - rb;
            if (TALLY) { cn_hdr += XGM_SU(nb); }
            if (lane < nb) {
                hb_meta = seg.blk_meta[rb + lane]; hb_first = seg.blk_first[rb + lane]; hb_word = seg.blk_word[rb + lane];
                hb_next = lane + 1u < nb ? seg.blk_first[rb + lane + 1u] : 0xFFFFFFFFu;
                if (PHRASE) hb_pos = seg.blk_pos[rb + lane];
            }
        }
        if (SIDED) hc_off = 0;                                     /* 0 = no container (for this term / in this stripe) */
        if (TALLY) { cn_aux += T - td; }
        if (lane >= td && lane < T && (!SIDED || dense_reg != 0xFFFFFFFFu)) hc_off = seg.dense_dir[(size_t)dense_reg * seg.n_stripes + (s_begin + x)];
    };

    uint32_t dl[4] = {0, 0, 0, 0};
    uint32_t stripe_base = 0;
    uint32_t hc_cur = 0;                                           /* container offsets of the stripe being processed */

    /* wdf of the dense terms [t_lo, T) for the n_c candidates in c_slot: ONE byte load per candidate
     * and term from the container's direct wdf+1 array (0 = the term does not index that docid) */
    /* with positions: the candidate's whole 64-byte sector of wdf+1 bytes (the memory transaction the one-byte probe costs
     * anyway) gives its wdf AND, summed over the bucket's earlier slots, where its positions start behind the bucket's base:
     * entry = pos_base[slot / 64] + Σ wdf of the slots before it in the bucket */
    auto sector_probe = [&](uint32_t t, uint32_t slot, bool active, uint32_t& wdf1, uint32_t& pos) {
        const uint32_t oo = __builtin_amdgcn_readlane(hc_cur, t & 63u);
        wdf1 = 0; pos = 0;
        if (active) {
            const unsigned char* wbytes = seg.dense_data + (size_t)oo * 16 + (size_t)NW * 4;
            const uint4* sp = reinterpret_cast<const uint4*>(wbytes + (slot & ~63u));
            const uint4 q0 = sp[0], q1 = sp[1], q2 = sp[2], q3 = sp[3];
            pos = reinterpret_cast<const uint32_t*>(wbytes + W)[slot >> 6];
            const uint32_t wv[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
            const uint32_t kk = slot & 63u;
#pragma unroll
            for (uint32_t j = 0; j < 16u; ++j) {
                const uint32_t x = wv[j];
                const uint32_t nz1 = ((x | ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu)) >> 7) & 0x01010101u;     /* 1 in every present slot's byte */
                const int rel = (int)kk - (int)(4u * j);                                             /* bytes of this word before the slot */
                const uint32_t m = rel >= 4 ? 0xFFFFFFFFu : (rel <= 0 ? 0u : ((1u << (8u * (uint32_t)rel)) - 1u));
                pos = __builtin_amdgcn_sad_u8((x - nz1) & m, 0u, pos);                               /* += Σ (wdf+1-1) of those bytes */
                if ((kk >> 2) == j) wdf1 = (x >> (8u * (kk & 3u))) & 0xFFu;
            }
        }
    };
    /* Once a positional query prunes by weight, nearly every candidate is dropped before its positions matter: the probe then
     * fetches the wdf byte only, and where the positions start is worked out for the few survivors (score_candidates). */
    bool lazy_pos = false;
    uint32_t lazy_tlo = 0;
    auto probe_dense = [&](uint32_t t_lo, uint32_t n_c) {
        lazy_pos = PHRASE && pos_prune && (theta_valid || theta_glob);
        lazy_tlo = t_lo;
        for (uint32_t c0 = 0; c0 < n_c; c0 += 64u) {
            const uint32_t o = c0 + lane;
            const bool valid = o < n_c;
            const uint32_t slot = valid ? c_slot[o] : 0u;
            const uint32_t sec = TALLY ? tally_sectors(valid, slot, 6u) : 0u;
            if (PHRASE && !lazy_pos) {
                for (uint32_t t = t_lo; t < T; ++t) {
                    if (TALLY) { cn_probe += sec; cn_probe_raw += n_c - c0 < 64u ? n_c - c0 : 64u; cn_aux += sec; }
                    uint32_t wdf1, pos;
                    sector_probe(t, slot, valid, wdf1, pos);
                    if (valid) {
                        c_w[(size_t)t * CAND + o] = (TabT)wdf1;
                        c_pos[(size_t)t * CAND + o] = pos;
                    }
                }
                continue;
            }
            for (uint32_t t0 = t_lo; t0 < T; t0 += 4u) {
                uint32_t wv[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    wv[u] = 0;
                    const uint32_t oo = __builtin_amdgcn_readlane(hc_cur, (t0 + u) & 63u);
                    if (TALLY) { if (t0 + u < T && (!SIDED || oo)) { cn_probe += sec; cn_probe_raw += n_c - c0 < 64u ? n_c - c0 : 64u; } }
                    if (t0 + u < T && valid && (!SIDED || oo))
                        wv[u] = seg.dense_data[(size_t)oo * 16 + (size_t)NW * 4 + slot];
                }
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u)
                    if (valid && t0 + u < T) c_w[(size_t)(t0 + u) * CAND + o] = (TabT)wv[u];
            }
        }
        wave_lds_fence();
    };

    bool q_mode = false;                                           /* queue path: the round's docids are in q_did (one per lane) */
    uint32_t q_did = 0;
    uint32_t q_off = 0;                                            /* ... and their rows start at c_w[..][q_off] */
    /* candidates present in every term are matches: BM25 + top-k, 64 per round; also clears c_w */
    auto score_candidates = [&](uint32_t n_c, bool dl_ready) {
        if (PHRASE && hist_g) look_at_histogram();
        for (uint32_t i0 = 0; i0 < n_c; i0 += 64u) {
            if (__builtin_expect(tkn + 64u > cap || (PHRASE && pos_prune && !theta_valid && tkn >= k), 0)) {      /* (positional pruning wants its threshold as soon as k matches are held; rare: spills stay out of the loops) */
                if (MAYBE) wave_topk_sort_m(tk_w, tk_d, tk_m, cap, lane); else wave_topk_sort(tk_w, tk_d, cap, lane);
                tkn = tkn < k ? tkn : k;
                if (tkn == k) { theta_valid = true; theta_w = tk_w[k - 1]; theta_d = tk_d[k - 1]; }
                for (uint32_t i = tkn + lane; i < cap; i += 64u) { tk_w[i] = 0; tk_d[i] = 0xFFFFFFFFu; if (MAYBE) tk_m[i] = 0; }
                wave_lds_fence();
            }
            const uint32_t oi = i0 + lane, o = q_off + oi;
            bool take = false;
            uint64_t wb = 0;
            uint32_t did = 0, subqs = 0;
            bool pass = false;
            if (oi < n_c) {
                pass = true;
                for (uint32_t t = 0; t < TR; ++t) pass = pass && c_w[(size_t)t * CAND + o] != 0;
                for (uint32_t t = TR; t < T; ++t) pass = pass && (!((q.neg_mask >> t) & 1u) || c_w[(size_t)t * CAND + o] == 0);   /* AND_NOT */
            }
            /* Positional queries whose match count need not be exact (check_at_least within the page, Xapiand's default): the
             * document's weight does not depend on its positions, so weigh FIRST — a candidate that cannot enter the unit's top k
             * (k positional matches are already held) is dropped without looking at its positions.  For phrases of frequent
             * terms this removes nearly every positional test; the hits are the same, the match count becomes a lower bound. */
            bool pre_weighed = false;
            if (PHRASE && phrase && pos_prune && (theta_valid || theta_glob)) {
                pre_weighed = true;
                if (pass) {
                    did = q_mode ? q_did : stripe_base + c_slot[o];
                    uint32_t dlen;
                    if (dl_ready && i0 < 256u) {
                        dlen = dl[0];
#pragma unroll
                        for (uint32_t c = 1; c < 4u; ++c) dlen = (i0 == c * 64u) ? dl[c] : dlen;
                    } else {
                        dlen = seg.doclen[did];
                    }
                    const double len = (double)dlen;
                    double normlen = len * q.len_factor;
                    normlen = normlen > q.min_normlen ? normlen : q.min_normlen;
                    const double denom_len = q.k1 * (normlen * q.b + (1.0 - q.b));
                    double weight = 0.0;
                    for (uint32_t t = 0; t < TR; ++t) {
                        const double wdf = (double)((uint32_t)c_w[(size_t)t * CAND + o] - 1u);
                        const double denom = denom_len + wdf;
                        weight = weight + q.termweight[t] * (wdf / denom);
                    }
                    wb = (uint64_t)__double_as_longlong(weight);
                    if ((theta_valid && !cand_before(wb, did, theta_w, theta_d)) || wb < theta_glob) { pass = false; pos_pruned = true; }
                }
                if (TALLY) { const uint32_t n_ = (uint32_t)__popcll(__ballot(oi < n_c && !(dl_ready && i0 < 256u))); cn_dl += n_; cn_dl_raw += n_; }
            }
            if (PHRASE && phrase && lazy_pos && __ballot(pass)) {
                const uint32_t slot = pass ? (uint32_t)c_slot[o < CAND ? o : 0u] : 0u;
                for (uint32_t t = lazy_tlo; t < T; ++t) {
                    uint32_t wdf1, pos;
                    sector_probe(t, slot, pass, wdf1, pos);
                    if (pass) c_pos[(size_t)t * CAND + o] = pos;
                }
                wave_lds_fence();
            }
            if (PHRASE && phrase && __ballot(pass)) {
                /* K6: ExactPhrasePostList / PhrasePostList / NearPostList::test_doc for the round's 64 documents.
                 * Stage: every lane fetches up to kPosFast positions of each term of ITS document with two 16-byte
                 * loads (independent of any control flow: all of them are in flight together, two terms at a time) and
                 * the wave parks them in LDS; then each lane runs the predicate on LDS data.  Documents with more
                 * positions of a term, or a term with 4-byte positions, take the serial path straight from HBM. */
                auto cnt = [&](uint32_t t) { return (uint32_t)c_w[(size_t)t * CAND + (o < CAND ? o : 0u)] - 1u; };
                bool slow = (q.flags & XGM_QF_NEAR_COLOC) != 0u;      /* (NEAR where terms may share a position: the reference's own steps, serially) */
                if (TALLY) { if (pass) { for (uint32_t t = 0; t < T; ++t) cn_pos += cnt(t); } }
                for (uint32_t t0 = 0; t0 < T; t0 += 2u) {
                    Pos8 ra[2][2];
                    uint32_t jmax = 0;
#pragma unroll
                    for (uint32_t u = 0; u < 2u; ++u) {
                        const uint32_t t = t0 + u;
                        ra[u][0] = Pos8{0, 0, 0, 0}; ra[u][1] = Pos8{0, 0, 0, 0};
                        if (t < T) {
                            const uint32_t w16 = __builtin_amdgcn_readlane(tflag_reg, t) & XGM_TF_POS16;
                            const uint32_t nt = pass ? cnt(t) : 0u;
                            const bool fast = w16 != 0u && nt <= kPosFast;
                            slow = slow || (pass && !fast);
                            const uint64_t tp = rl64(tpos_reg, t);                /* (read by every lane, not under the per-lane test) */
                            if (pass && fast) {
                                const unsigned char* src = seg.positions + tp + (size_t)c_pos[(size_t)t * CAND + o] * 2u;
                                if (nt > 0u) ra[u][0] = *reinterpret_cast<const Pos8*>(src);
                                if (nt > 8u) ra[u][1] = *reinterpret_cast<const Pos8*>(src + 16);
                            }
                            const uint32_t tier = __ballot(fast && nt > 8u) ? 16u : (__ballot(fast && nt > 4u) ? 8u : 4u);
                            jmax = tier > jmax ? tier : jmax;
                        }
                    }
#pragma unroll
                    for (uint32_t u = 0; u < 2u; ++u) {
                        const uint32_t t = t0 + u;
                        if (t < T) {
                            const uint32_t r8[8] = {ra[u][0].a, ra[u][0].b, ra[u][0].c, ra[u][0].d, ra[u][1].a, ra[u][1].b, ra[u][1].c, ra[u][1].d};
#pragma unroll
                            for (uint32_t j = 0; j < kPosFast; ++j)
                                if (j < jmax) lpos[(t * kPosFast + j) * 64u + lane] = (uint16_t)((j & 1u) ? (r8[j >> 1] >> 16) : (r8[j >> 1] & 0xFFFFu));
                        }
                    }
                }
                wave_lds_fence();
                if (pass && !slow) {
                    const LdsPos L{lpos, lane};
                    pass = (q.flags & XGM_QF_NEAR) ? lds_near_window(L, cnt, T, q.window)
                         : (q.flags & XGM_QF_EXACT) ? lds_phrase_exact(L, cnt, q.phrase_index, T)
                                                    : lds_phrase_window(L, cnt, q.phrase_index, T, q.window);
                }
                if (__builtin_expect(__ballot(pass && slow) != 0ull, 0)) {
                    /* (the loop over the terms is wave-uniform, only the stores and the test are per lane: a lane whose document
                     * takes the serial path reads the terms' registers together with the lanes that do not) */
                    PosList pl[XGM_PHRASE_MAX_TERMS];
                    for (uint32_t t = 0; t < T && t < XGM_PHRASE_MAX_TERMS; ++t) {
                        const uint64_t tp = rl64(tpos_reg, t);
                        const uint32_t w16 = __builtin_amdgcn_readlane(tflag_reg, t) & XGM_TF_POS16;
                        if (pass && slow) {
                            pl[t].p = seg.positions + tp + (size_t)c_pos[(size_t)t * CAND + o] * (w16 ? 2u : 4u);
                            pl[t].n = cnt(t);
                            pl[t].w16 = w16;
                        }
                    }
                    if (pass && slow) pass = posfilter_slow(pl, q, T);
                }
                wave_lds_fence();
            }
            if (oi < n_c) {
                if (pass && pre_weighed) {
                    ++matches;
                    subqs = (uint32_t)__popc(q.score_mask);
                    take = true;                                   /* it beat the k-th best known before its positions were tested */
                } else if (pass) {
                    ++matches;
                    did = q_mode ? q_did : stripe_base + c_slot[o];
                    uint32_t dlen;
                    if (dl_ready && i0 < 256u) {
                        dlen = dl[0];
#pragma unroll
                        for (uint32_t c = 1; c < 4u; ++c) dlen = (i0 == c * 64u) ? dl[c] : dlen;
                    } else {
                        dlen = seg.doclen[did];
                    }
                    /* BM25Weight::get_sumpart, bm25weight.cc:170-181 — same operations, same order */
                    const double len = (double)dlen;
                    double normlen = len * q.len_factor;
                    normlen = normlen > q.min_normlen ? normlen : q.min_normlen;
                    const double denom_len = q.k1 * (normlen * q.b + (1.0 - q.b));
                    double weight = 0.0;                       /* MultiAndPostList::get_weight: ((0 + w0) + w1) + ... */
                    if (MAYBE && q.op == XGM_OP_AND_MAYBE) {
                        /* leaves in registers, summed by the query's in-place program: the chain of the required
                         * terms, plus — AND_MAYBE — the OR tree of the optional ones (absent leaf = -0.0) */
                        andw_d8 v;
#pragma unroll
                        for (uint32_t t = 0; t < 8u; ++t) {
                            double wt = -0.0;
                            const uint32_t e = t < T ? (uint32_t)c_w[(size_t)t * CAND + o] : 0u;
                            if (e) {
                                const double wdf = (double)(e - 1u);
                                wt = q.termweight[t] * (wdf / (denom_len + wdf));
                                subqs += (q.score_mask >> t) & 1u;
                            }
                            v[t] = wt;
                        }
                        for (uint32_t j = 0; j < q.n_nodes; ++j) {
                            const uint32_t a = (uint32_t)(prog_a >> (8u * j)) & 7u, b = (uint32_t)(prog_b >> (8u * j)) & 7u;
                            const double x = v[a] + v[b];
                            v[a] = x;
                        }
                        weight = v[prog_root & 7u];
                    } else {
                        subqs = (uint32_t)__popc(q.score_mask);
                        for (uint32_t t = 0; t < TR; ++t) {
                            const double wdf = (double)((uint32_t)c_w[(size_t)t * CAND + o] - 1u);
                            const double denom = denom_len + wdf;
                            weight = weight + q.termweight[t] * (wdf / denom);
                        }
                    }
                    wb = (uint64_t)__double_as_longlong(weight);
                    take = !theta_valid || cand_before(wb, did, theta_w, theta_d);
                }
                for (uint32_t t = 0; t < T; ++t) c_w[(size_t)t * CAND + o] = 0;
            }
            if (TALLY) { if (!pre_weighed && !(dl_ready && i0 < 256u)) { const uint32_t n_ = (uint32_t)__popcll(__ballot(pass)); cn_dl += n_; cn_dl_raw += n_; } }
            if (PHRASE && hist_g && take) {
                int b = (int)(wb >> 47) - hbase;
                b = b < 0 ? 0 : (b > (int)XGM_OR_HIST - 1 ? (int)XGM_OR_HIST - 1 : b);
                atomicAdd(&hist_g[b], 1u);
            }
            const uint64_t tm = __ballot(take);
            if (take) { const uint32_t p = tkn + mbcnt(tm); tk_w[p] = wb; tk_d[p] = did; if (MAYBE) tk_m[p] = (uint8_t)subqs; }
            tkn += (uint32_t)__popcll(tm);
        }
    };

    /* P3c: wdf of the right-hand terms WITHOUT containers (AND_NOT: excluded, AND_MAYBE: optional) for the candidates
     * whose ordinals are [lo, lo + CAND): per term the run's block headers in ONE load per field (lane j = block j),
     * a ballot of the blocks whose 128-slot buckets hold a candidate, then their payloads four at a time in flight;
     * every decoded posting that is a candidate (bitmap) is scattered to its ordinal (rank table) — as P3a does for
     * the required terms.  `coarse`: bit b = bucket b of the stripe holds a candidate. */
    auto scatter_sparse_rhs = [&](uint32_t sl_, unsigned long long coarse_, uint32_t lo_) {
        for (uint64_t xm = sparse_neg; xm; xm &= xm - 1u) {
            const uint32_t t = (uint32_t)__builtin_ctzll(xm);
            const uint32_t rbx = rs[t * SPG + sl_], nbx = re[t * SPG + sl_] - rbx;
            TabT* row = c_w + (size_t)t * CAND;
            const uint64_t tbx = tbase(t);
            for (uint32_t j0 = 0; j0 < nbx; j0 += 64u) {
                const uint32_t nb = nbx - j0 < 64u ? nbx - j0 : 64u;
                uint32_t x_meta = 0, x_first = 0, x_word = 0, x_next = 0xFFFFFFFFu;
                if (lane < nb) {
                    x_meta = seg.blk_meta[rbx + j0 + lane]; x_first = seg.blk_first[rbx + j0 + lane]; x_word = seg.blk_word[rbx + j0 + lane];
                    x_next = j0 + lane + 1u < nbx ? seg.blk_first[rbx + j0 + lane + 1u] : 0xFFFFFFFFu;
                }
                if (TALLY) { cn_hdr += XGM_SU(nb); }
                bool need = false;
                if (lane < nb) {
                    const uint32_t blo = (x_first - stripe_base) >> 7;
                    const uint32_t bhi = ((x_next == 0xFFFFFFFFu ? W : x_next - stripe_base) - 1u) >> 7;
                    const unsigned long long mm = (bhi >= 63u ? ~0ull : ((1ull << (bhi + 1u)) - 1ull)) & ~((1ull << blo) - 1ull);
                    need = (coarse_ & mm) != 0ull;
                }
                uint64_t mask = __ballot(need);
                while (mask) {
                    uint32_t jj[4], nn = 0;
                    Words4 pv[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) {
                        jj[u] = 0; pv[u] = Words4{0, 0, 0, 0};
                        if (mask) {
                            jj[u] = (uint32_t)__builtin_ctzll(mask);
                            mask &= mask - 1u;
                            const uint32_t bm = __builtin_amdgcn_readlane(x_meta, jj[u]);
                            if (TALLY) { cn_blkw += payload_words(bm) - 2u; }
                            const uint32_t bwd = __builtin_amdgcn_readlane(x_word, jj[u]);      /* read by every lane, not under the per-lane test */
                            if (lane * 4u < payload_words(bm)) pv[u] = *reinterpret_cast<const Words4*>(seg.words + tbx + bwd + lane * 4u);
                            nn = u + 1u;
                        }
                    }
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) {
                        if (u < nn) {
                            const uint32_t bmeta = __builtin_amdgcn_readlane(x_meta, jj[u]);
                            const uint32_t bfirst = __builtin_amdgcn_readlane(x_first, jj[u]);
                            if (lane * 4u < payload_words(bmeta)) {
                                stage[lane * 4u] = pv[u].a; stage[lane * 4u + 1] = pv[u].b; stage[lane * 4u + 2] = pv[u].c; stage[lane * 4u + 3] = pv[u].d;
                            }
                            wave_lds_fence();
                            const DecodedPair r = unpack_staged<false>(stage, bfirst, bmeta, lane);
                            wave_lds_fence();
#pragma unroll
                            for (uint32_t h = 0; h < 2u; ++h) {
                                if (h ? r.v1 : r.v0) {
                                    const uint32_t sx = (h ? r.d1 : r.d0) - stripe_base, wd = sx >> 5, bit = sx & 31u;
                                    const uint32_t bm = bitmap[wd];
                                    if ((bm >> bit) & 1u) {
                                        const uint32_t ord = (uint32_t)rankw[wd] + (uint32_t)__popc(bm & ((1u << bit) - 1u)) - lo_;
                                        if (ord < CAND) row[ord] = (TabT)((h ? r.w1 : r.w0) + 1u);
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
        wave_lds_fence();
    };

    if (qpath) {
        /* ---- queue path: units that never need a per-stripe table.  Either every required term has containers
         * (td == 0: the candidates of a stripe are the bits of the AND of the bitmaps) or only the rarest one is
         * block-coded (td == 1: its postings are the candidates, unpacked 64 at a time ACROSS block and stripe
         * boundaries — a tail term has a handful of postings per block).  Candidates — (stripe, slot) pairs — queue up
         * in LDS across stripes.  Round A takes 64 of them, probes the containers (each lane its own stripe's) and
         * pushes the survivors with their wdf bytes on a match stack; round B pops 64 matches, gathers their document
         * lengths and weighs them with every lane busy.  The per-stripe path spent most of its issue slots weighing a
         * few documents on 64 lanes and paid a memory latency chain per stripe. */
        typedef uint32_t andw_u4 __attribute__((ext_vector_type(4)));
        uint32_t* ring = reinterpret_cast<uint32_t*>(c_slot);      /* CAND u16 = 256 u32 entries: (local stripe << 16) | slot */
        uint16_t* ring_w = rankw;                                  /* td == 1: wdf + 1 of term 0, same index */
        uint32_t* st_did = bitmap;                                 /* match stack: docids (rows of c_w hold the wdf bytes) */
        uint32_t* st_dl = stage;                                   /* ... and, td == 0, their document lengths */
        const bool dl_in_a = td == 0u;                             /* nearly every candidate of an all-container unit is a match: its document
                                                                      length travels with the probes instead of waiting for round B */
        constexpr uint32_t QCAP = CAND / 2u;
        uint32_t pend = 0u, head = 0u, nB = 0u;
        auto next_q = [&](uint32_t from) {
            uint32_t x = from;
            for (; x < n_local; ++x) {
                bool all = true;
                for (uint32_t t = 0; t < TR; ++t) all = all && rs[t * SPG + x] != 0u;
                if (all) break;
            }
            return x;
        };
        auto load_bitmaps = [&](uint32_t x, andw_u4 (&raw)[4]) {
            if (TALLY) { cn_bmpw += (TR < 4u ? TR : 4u) * NW; }
#pragma unroll
            for (uint32_t t = 0; t < 4u; ++t) {
                raw[t] = andw_u4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                if (t < TR) {
                    const uint32_t off = XGM_SU(rs[t * SPG + x]);
                    const andw_u4* bmp = reinterpret_cast<const andw_u4*>(seg.dense_data + (size_t)off * 16);
                    raw[t] = lane * 4u < NW ? bmp[lane] : andw_u4{0u, 0u, 0u, 0u};
                }
            }
        };
        /* round A: probe the first n (<= 64) queued candidates, push the matches */
        auto round_a = [&](uint32_t n) {
            const bool valid = lane < n;
            const uint32_t ri = (head + lane) & (QCAP - 1u);
            const uint32_t e = valid ? ring[ri] : 0u;
            const uint32_t x = e >> 16, slot = e & 0xFFFFu;
            uint32_t wv[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
            const uint32_t sec = TALLY ? tally_sectors(valid, e, 6u) : 0u;
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u) {
                const uint32_t oo = (u < T && u >= td && valid) ? rs[u * SPG + x] : 0u;
                if (TALLY) { if (u < T && __ballot(oo != 0u)) { cn_probe += sec; cn_probe_raw += n; } }
                if (oo) wv[u] = seg.dense_data[(size_t)oo * 16 + (size_t)NW * 4 + slot];
            }
            uint32_t dlv = 0u;
            if (dl_in_a) {
                dlv = valid ? seg.doclen[((s_begin + x) << SB) + slot] : 0u;
                if (TALLY) { cn_dl += tally_sectors(valid, ((s_begin + x) << SB) + slot, 4u); cn_dl_raw += n; }
            }
            if (td == 1u && valid) wv[0] = ring_w[ri];
            bool pass = valid;
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u) {
                if (u < TR) pass = pass && wv[u] != 0u;
                else if (SIDED && u < T) pass = pass && (!((q.neg_mask >> u) & 1u) || wv[u] == 0u);       /* AND_NOT */
            }
            const uint64_t pm = __ballot(pass);
            if (pass) {
                const uint32_t pos = nB + mbcnt(pm);
                st_did[pos] = ((s_begin + x) << SB) + slot;
                if (dl_in_a) st_dl[pos] = dlv;
#pragma unroll
                for (uint32_t u = 0; u < 8u; ++u) if (u < T) c_w[(size_t)u * CAND + pos] = (TabT)wv[u];
            }
            nB += (uint32_t)__popcll(pm);
            head = (head + n) & (QCAP - 1u);
            pend -= n;
            wave_lds_fence();
        };
        /* round B: weigh the top n (<= 64) matches of the stack */
        auto round_b = [&](uint32_t n) {
            q_off = nB - n;
            const bool valid = lane < n;
            const uint32_t did = valid ? st_did[q_off + lane] : 0u;
            if (dl_in_a) {
                dl[0] = valid ? st_dl[q_off + lane] : 0u;
            } else {
                dl[0] = valid ? seg.doclen[did] : 0u;
                if (TALLY) { cn_dl += tally_sectors(valid, did, 4u); cn_dl_raw += n; }
            }
            q_mode = true; q_did = did;
            score_candidates(n, true);
            q_mode = false; q_off = 0u;
            nB -= n;
            wave_lds_fence();
        };
        /* producer state, td == 0: the stripe whose bits are being enumerated */
        uint32_t sl = td == 0u ? next_q(0) : n_local;
        uint32_t m[4] = {0u, 0u, 0u, 0u};
        uint32_t o = 0u, n_total = 0u, done = 0u;
        bool fresh = true;                                         /* m holds a stripe whose bits have not been counted yet */
        if (sl < n_local) {
            andw_u4 raw[4];
            load_bitmaps(sl, raw);
            const andw_u4 a = raw[0] & raw[1] & raw[2] & raw[3];
            m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w;
        }
        /* producer state, td == 1: lane j holds the header of block bb + j of the batch; p0 = postings of the batch consumed */
        uint32_t b_next = q_b0, nbat = 0u, p_tot = 0u, p0 = 0u, carry = 0u;
        uint32_t bj_first = 0u, bj_meta = 0u, bj_word = 0u, bj_p = 0u;
        const uint64_t tb0 = tbase(0);
        bool more = td == 0u ? sl < n_local : b_next < q_b1;
        while (more || pend || nB) {
            if (more && td == 0u) {
                /* one stripe (or what fits of it): its bits go to the queue; the next stripe's bitmaps are in flight meanwhile */
                andw_u4 raw[4];
                uint32_t sl1 = n_local;
                bool pre = false;
                if (fresh) {
                    const uint32_t cnt = (uint32_t)(__popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]));
                    const uint32_t incl = wave_incl_scan(cnt);
                    n_total = __builtin_amdgcn_readlane(incl, 63);
                    o = incl - cnt;                                /* this lane's next ordinal in the stripe */
                    done = 0u;
                    fresh = false;
                }
                const uint32_t room = QCAP - pend;
                const uint32_t take_n = n_total - done < room ? n_total - done : room;
                const uint32_t lim = done + take_n;
                const bool last_piece = lim == n_total;
                if (last_piece) {
                    sl1 = next_q(sl + 1u);
                    pre = sl1 < n_local;
                    if (pre) load_bitmaps(sl1, raw);
                }
                const uint32_t wbase = head + pend - done;         /* ring position of ordinal 0 (mod QCAP) */
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) {
                    while (m[i] && o < lim) {
                        const uint32_t bit = (uint32_t)__ffs(m[i]) - 1u;
                        ring[(wbase + o) & (QCAP - 1u)] = (sl << 16) | ((lane * 4u + i) * 32u + bit);
                        m[i] &= m[i] - 1u;
                        ++o;
                    }
                }
                pend += take_n;
                if (TALLY) { q_cands += take_n; }
                done = lim;
                if (last_piece) {
                    if (pre) {
                        const andw_u4 a = raw[0] & raw[1] & raw[2] & raw[3];
                        m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w;
                    }
                    sl = sl1;
                    fresh = true;
                    more = pre;
                }
                wave_lds_fence();
            } else if (more) {
                /* 64 postings of term 0, one per lane, across block boundaries */
                if (p0 >= p_tot) {
                    nbat = q_b1 - b_next < 64u ? q_b1 - b_next : 64u;
                    bj_meta = bj_first = bj_word = 0u;
                    if (lane < nbat) { bj_meta = seg.blk_meta[b_next + lane]; bj_first = seg.blk_first[b_next + lane]; bj_word = seg.blk_word[b_next + lane]; }
                    const uint32_t cj = lane < nbat ? XGM_META_COUNT(bj_meta) : 0u;
                    const uint32_t incl = wave_incl_scan(cj);
                    bj_p = incl - cj;
                    p_tot = __builtin_amdgcn_readlane(incl, 63);
                    p0 = 0u;
                    b_next += nbat;
                    if (TALLY) {
                        cn_hdr += nbat;
                        const uint32_t wj = lane < nbat ? payload_words(bj_meta) - 2u : 0u;
                        cn_blkw += __builtin_amdgcn_readlane(wave_incl_scan(wj), 63);
                    }
                }
                const uint32_t p = p0 + lane;
                const bool valid = p < p_tot;
                uint32_t j = 0u;                                   /* the block that holds posting p: the last j with bj_p[j] <= p */
#pragma unroll
                for (uint32_t sft = 32u; sft; sft >>= 1) {
                    const uint32_t cj = j + sft;
                    const uint32_t pc = (uint32_t)__shfl((int)bj_p, (int)(cj & 63u));
                    if (cj < nbat && pc <= p) j = cj;
                }
                const uint32_t pj = (uint32_t)__shfl((int)bj_p, (int)j);
                const uint32_t first = (uint32_t)__shfl((int)bj_first, (int)j);
                const uint32_t meta = (uint32_t)__shfl((int)bj_meta, (int)j);
                const uint32_t word = (uint32_t)__shfl((int)bj_word, (int)j);
                const uint32_t i = p - pj;
                const uint32_t nj = XGM_META_COUNT(meta), bwg = XGM_META_BWG(meta), bww = XGM_META_BWW(meta);
                const uint32_t* pw = seg.words + tb0 + word;
                uint32_t gap = 0u, wdf = 0u;
                if (valid) {
                    if (i > 0u) gap = extract_bits(pw, i, bwg) + 1u;
                    wdf = extract_bits(pw + ((nj * bwg + 31u) >> 5), i, bww);
                }
                const uint32_t sc = wave_incl_scan(gap);
                /* docid = first docid of the block + the gaps since; a block that began in an earlier step continues from
                 * the last docid of that step */
                const bool began_here = pj >= p0;
                const uint32_t s_at = (uint32_t)__shfl((int)sc, (int)((pj - p0) & 63u));
                const uint32_t did = began_here ? first + (sc - s_at) : carry + sc;
                carry = __builtin_amdgcn_readlane(did, 63);
                if (valid) {
                    const uint32_t ri = (head + pend + lane) & (QCAP - 1u);
                    const uint32_t x = (did >> SB) - s_begin;
                    ring[ri] = (x << 16) | (did & (W - 1u));
                    ring_w[ri] = (uint16_t)(wdf + 1u);
                }
                pend += p_tot - p0 < 64u ? p_tot - p0 : 64u;
                if (TALLY) { q_cands += p_tot - p0 < 64u ? p_tot - p0 : 64u; }
                p0 += 64u;
                more = p0 < p_tot || b_next < q_b1;
                wave_lds_fence();
            }
            /* consumers: full rounds while the producer has more, the remainders after it */
            unsigned long long ph_t = TALLY ? __builtin_readcyclecounter() : 0ull;
            if (pend >= 64u || (!more && pend)) round_a(pend < 64u ? pend : 64u);
            if (TALLY) { const unsigned long long t1 = __builtin_readcyclecounter(); ph_a += t1 - ph_t; ph_t = t1; }
            if (nB >= 64u || (!more && !pend && nB)) round_b(nB < 64u ? nB : 64u);
            if (TALLY) { ph_b += __builtin_readcyclecounter() - ph_t; }
        }
    }
    uint32_t sl = qpath ? n_local : next_active(0);
    if ((PHRASE || SIDED) && td == 0u && !qpath) {             /* (a plain conjunction with td == 0 always takes the queue path) */
        /* ---- every required term dense: candidates = AND of the containers' bitmaps (one 16-byte load per lane and
         * term).  Software pipeline, ONE exposed memory latency per stripe instead of three: the container offsets run two
         * stripes ahead, the NEXT stripe's bitmaps are requested before this stripe's probes and ANDed after them, and the
         * candidates' document lengths travel with their wdf probes (without a positional filter nearly every candidate
         * of this path is a match). */
        typedef uint32_t andw_u4 __attribute__((ext_vector_type(4)));
        auto load_dir = [&](uint32_t x) {
            uint32_t v = 0u;
            if (TALLY) { cn_aux += T; }
            if (lane < T && (!SIDED || dense_reg != 0xFFFFFFFFu)) v = seg.dense_dir[(size_t)dense_reg * seg.n_stripes + (s_begin + x)];
            return v;
        };
        auto load_bitmaps = [&](uint32_t hc, andw_u4 (&raw)[4]) {
            if (TALLY) { cn_bmpw += (TR < 4u ? TR : 4u) * NW; }
#pragma unroll
            for (uint32_t t = 0; t < 4u; ++t) {
                raw[t] = andw_u4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                if (t < TR) {
                    const andw_u4* bmp = reinterpret_cast<const andw_u4*>(seg.dense_data + (size_t)rl32(hc, t) * 16);
                    raw[t] = lane * 4u < NW ? bmp[lane] : andw_u4{0u, 0u, 0u, 0u};
                }
            }
        };
        uint32_t sl1 = sl < n_local ? next_active(sl + 1u) : n_local;
        uint32_t hcA = sl < n_local ? load_dir(sl) : 0u;
        uint32_t hcB = sl1 < n_local ? load_dir(sl1) : 0u;
        uint32_t m[4] = {0u, 0u, 0u, 0u};
        if (sl < n_local) {
            andw_u4 raw[4];
            load_bitmaps(hcA, raw);
            const andw_u4 a = raw[0] & raw[1] & raw[2] & raw[3];
            m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w;
        }
        while (sl < n_local) {
            stripe_base = (s_begin + sl) << SB;
            hc_cur = hcA;
            const uint32_t sl2 = sl1 < n_local ? next_active(sl1 + 1u) : n_local;
            uint32_t hcC = 0u;                                         /* container offsets two stripes ahead: requested below */
            const uint32_t cnt = (uint32_t)(__popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]));
            const uint32_t incl = wave_incl_scan(cnt);
            const uint32_t n_total = __builtin_amdgcn_readlane(incl, 63);
            uint32_t o = incl - cnt;                                   /* this lane's next ordinal */
            const bool rhs_sparse = SIDED && sparse_neg != 0ull && n_total != 0u;
            unsigned long long coarse0 = 0ull;
            if (rhs_sparse) {
                /* the AND of the bitmaps IS the candidates' bitmap; ordinals follow from the same prefix sums */
                uint32_t run = o;
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) {
                    const uint32_t w = lane * 4u + i;
                    if (w < NW) { bitmap[w] = m[i]; rankw[w] = (uint16_t)run; }
                    run += (uint32_t)__popc(m[i]);
                }
                coarse0 = __ballot((m[0] | m[1] | m[2] | m[3]) != 0u);      /* lane l owns words 4l..4l+3 = the 128 slots of bucket l */
                wave_lds_fence();
            }
            const bool pre = sl1 < n_local;
            uint32_t mn[4] = {0u, 0u, 0u, 0u};
            /* the common stripe holds at most one round of candidates: their document lengths, their wdf bytes AND the next
             * stripe's bitmaps are all requested before anything is waited for */
            const bool one_round = !PHRASE && n_total <= 64u;
            if (one_round) {
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) {
                    while (m[i]) {
                        const uint32_t bit = (uint32_t)__ffs(m[i]) - 1u;
                        c_slot[o] = (uint16_t)((lane * 4u + i) * 32u + bit);
                        m[i] &= m[i] - 1u;
                        ++o;
                    }
                }
                wave_lds_fence();
                const bool valid = lane < n_total;
                const uint32_t slot = valid ? c_slot[lane] : 0u;
                uint32_t wv[4] = {0u, 0u, 0u, 0u};
                uint32_t wx[4] = {0u, 0u, 0u, 0u};
                const uint32_t sec = TALLY ? tally_sectors(valid, slot, 6u) : 0u;
                if (TALLY) { if (n_total) { cn_dl += tally_sectors(valid, slot, 4u); cn_dl_raw += n_total; } }
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    const uint32_t oo = __builtin_amdgcn_readlane(hcA, u);
                    if (TALLY) { if (u < T && n_total && (!SIDED || oo)) { cn_probe += sec; cn_probe_raw += n_total; } }
                    if (u < T && valid && (!SIDED || oo)) wv[u] = seg.dense_data[(size_t)oo * 16 + (size_t)NW * 4 + slot];
                }
                if (SIDED) {
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) {
                        const uint32_t oo = __builtin_amdgcn_readlane(hcA, 4u + u);
                        if (TALLY) { if (4u + u < T && n_total && oo) { cn_probe += sec; cn_probe_raw += n_total; } }
                        if (4u + u < T && valid && oo) wx[u] = seg.dense_data[(size_t)oo * 16 + (size_t)NW * 4 + slot];
                    }
                }
                {
                    andw_u4 raw[4];
                    if (pre) load_bitmaps(hcB, raw);
                    if (sl2 < n_local) hcC = load_dir(sl2);
                    dl[0] = valid ? seg.doclen[stripe_base + slot] : 0u;       /* last: everything above is in flight by now */
                    if (pre) {
                        const andw_u4 a = raw[0] & raw[1] & raw[2] & raw[3];
                        mn[0] = a.x; mn[1] = a.y; mn[2] = a.z; mn[3] = a.w;
                    }
                }
                if (valid) {
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) if (u < T) c_w[(size_t)u * CAND + lane] = (TabT)wv[u];
                    if (SIDED) {
#pragma unroll
                        for (uint32_t u = 0; u < 4u; ++u) if (4u + u < T) c_w[(size_t)(4u + u) * CAND + lane] = (TabT)wx[u];
                    }
                }
                wave_lds_fence();
            } else {
                if (sl2 < n_local) hcC = load_dir(sl2);
            }
            if (!one_round && pre) {
                andw_u4 raw[4];
                load_bitmaps(hcB, raw);
                const andw_u4 a = raw[0] & raw[1] & raw[2] & raw[3];
                mn[0] = a.x; mn[1] = a.y; mn[2] = a.z; mn[3] = a.w;
            }
            for (uint32_t lo = 0; lo < n_total; lo += CAND) {
                const uint32_t n_c = n_total - lo < CAND ? n_total - lo : CAND;
                if (!one_round) {
#pragma unroll
                    for (uint32_t i = 0; i < 4u; ++i) {
                        while (m[i] && o < lo + CAND) {
                            const uint32_t bit = (uint32_t)__ffs(m[i]) - 1u;
                            c_slot[o - lo] = (uint16_t)((lane * 4u + i) * 32u + bit);
                            m[i] &= m[i] - 1u;
                            ++o;
                        }
                    }
                    wave_lds_fence();
                    if (!PHRASE) {
                        /* document lengths of the first 256 candidates: requested with their probes */
#pragma unroll
                        for (uint32_t c = 0; c < 4u; ++c) {
                            const uint32_t oc = lane + c * 64u;
                            dl[c] = oc < n_c ? seg.doclen[stripe_base + c_slot[oc]] : 0u;
                            if (TALLY) { cn_dl += tally_sectors(oc < n_c, oc < n_c ? (uint32_t)c_slot[oc] : 0u, 4u); }
                        }
                        if (TALLY) { cn_dl_raw += n_c < 256u ? n_c : 256u; }
                    }
                    probe_dense(0u, n_c);
                }
                if (rhs_sparse) scatter_sparse_rhs(sl, coarse0, lo);
                score_candidates(n_c, !PHRASE);
                wave_lds_fence();
            }
            if (rhs_sparse) {
                for (uint32_t i = lane; i < NW; i += 64u) { bitmap[i] = 0; rankw[i] = 0xFFFFu; }
                wave_lds_fence();
            }
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) m[i] = mn[i];
            sl = sl1; sl1 = sl2; hcA = hcB; hcB = hcC;
        }
    }
    if (td != 0u && sl < n_local) issue_headers(sl);
    while (sl < n_local) {
        stripe_base = (s_begin + sl) << SB;
        const uint32_t sl_next = next_active(sl + 1u);
        hc_cur = hc_off;

        const uint32_t r0 = rs[sl], r0e = re[sl];
        for (uint32_t cb = r0; cb < r0e; cb += CHUNKB) {
            const uint32_t nblk0 = r0e - cb < CHUNKB ? r0e - cb : CHUNKB;
            /* ---- P1: term-0 blocks of the chunk → candidates (ordinals follow docid order) ---- */
            uint32_t m0 = h0_meta, f0 = h0_first, w0 = h0_word, ps0 = h0_pos;
            if (cb != r0) {
                if (TALLY) { cn_hdr += XGM_SU(nblk0); }
                m0 = lane < nblk0 ? seg.blk_meta[cb + lane] : 0u;
                f0 = lane < nblk0 ? seg.blk_first[cb + lane] : 0u;
                w0 = lane < nblk0 ? seg.blk_word[cb + lane] : 0u;
                if (PHRASE) ps0 = lane < nblk0 ? seg.blk_pos[cb + lane] : 0u;
            }
            const uint32_t cnt0 = lane < nblk0 ? XGM_META_COUNT(m0) : 0u;
            const uint32_t incl = wave_incl_scan(cnt0);
            const uint32_t n_c = __builtin_amdgcn_readlane(incl, 63);
            Words4 p0[CHUNKB];
#pragma unroll
            for (uint32_t j = 0; j < CHUNKB; ++j) {
                p0[j] = Words4{0, 0, 0, 0};
                if (j < nblk0) {
                    const uint32_t mj = __builtin_amdgcn_readlane(m0, j);
                    if (TALLY) { cn_blkw += payload_words(mj) - 2u; }
                    const uint32_t wj = __builtin_amdgcn_readlane(w0, j);                       /* read by every lane, not under the per-lane test */
                    const uint64_t tb0j = tbase(0);
                    if (lane * 4u < payload_words(mj)) p0[j] = *reinterpret_cast<const Words4*>(seg.words + tb0j + wj + lane * 4u);
                }
            }
            unsigned long long coarse = 0;
#pragma unroll
            for (uint32_t j = 0; j < CHUNKB; ++j) {
                if (j < nblk0) {
                    const uint32_t meta = __builtin_amdgcn_readlane(m0, j);
                    const uint32_t obase = __builtin_amdgcn_readlane(incl, j) - XGM_META_COUNT(meta);
                    const uint32_t first = __builtin_amdgcn_readlane(f0, j);
                    if (lane * 4u < payload_words(meta)) {
                        stage[lane * 4u] = p0[j].a; stage[lane * 4u + 1] = p0[j].b; stage[lane * 4u + 2] = p0[j].c; stage[lane * 4u + 3] = p0[j].d;
                    }
                    wave_lds_fence();
                    DecodedPair r = unpack_staged<PHRASE>(stage, first, meta, lane);
                    wave_lds_fence();
                    const uint32_t pbase0 = PHRASE ? __builtin_amdgcn_readlane(ps0, j) : 0u;
                    const uint32_t s0 = r.d0 - stripe_base, s1 = r.d1 - stripe_base;
                    const uint32_t prev1 = (uint32_t)__shfl_up((int)s1, 1);
                    const uint32_t pw0 = lane == 0 ? 0xFFFFFFFFu : (prev1 >> 5);
                    unsigned long long cbits = 0;
                    if (r.v0) {
                        const uint32_t o = obase + 2u * lane;
                        c_slot[o] = (uint16_t)s0;
                        c_w[o] = (TabT)(r.w0 + 1u);
                        if (PHRASE) c_pos[o] = pbase0 + r.p0;
                        atomicOr(&bitmap[s0 >> 5], 1u << (s0 & 31u));
                        /* first candidate of its word inside this block; a later block can only add larger ordinals */
                        if ((s0 >> 5) != pw0 && rankw[s0 >> 5] == 0xFFFFu) rankw[s0 >> 5] = (uint16_t)o;
                        cbits |= 1ull << (s0 >> 7);
                    }
                    if (r.v1) {
                        const uint32_t o = obase + 2u * lane + 1u;
                        c_slot[o] = (uint16_t)s1;
                        c_w[o] = (TabT)(r.w1 + 1u);
                        if (PHRASE) c_pos[o] = pbase0 + r.p1;
                        atomicOr(&bitmap[s1 >> 5], 1u << (s1 & 31u));
                        if ((s1 >> 5) != (s0 >> 5) && rankw[s1 >> 5] == 0xFFFFu) rankw[s1 >> 5] = (uint16_t)o;
                        cbits |= 1ull << (s1 >> 7);
                    }
                    uint32_t clo = (uint32_t)cbits, chi = (uint32_t)(cbits >> 32);
                    for (int sh = 32; sh > 0; sh >>= 1) { clo |= (uint32_t)__shfl_xor((int)clo, sh); chi |= (uint32_t)__shfl_xor((int)chi, sh); }
                    coarse |= ((unsigned long long)chi << 32) | clo;
                    wave_lds_fence();                                   /* rankw of this block visible to the next */
                }
            }

            /* doclen of the first 256 candidates: requested now, consumed when scoring */
#pragma unroll
            for (uint32_t c = 0; c < 4u; ++c) {
                const uint32_t o = lane + c * 64u;
                dl[c] = o < n_c ? seg.doclen[stripe_base + c_slot[o]] : 0u;
                if (TALLY) { cn_dl += tally_sectors(o < n_c, o < n_c ? (uint32_t)c_slot[o] : 0u, 4u); }
            }
            if (TALLY) { cn_dl_raw += n_c < 256u ? n_c : 256u; }

            /* ---- P3a: sparse other terms [1, td), two at a time; only blocks whose 128-slot buckets hold a candidate ---- */
            auto bucket_need = [&](uint32_t first, uint32_t nfirst) {
                const uint32_t lo = (first - stripe_base) >> 7;
                const uint32_t hi = ((nfirst == 0xFFFFFFFFu ? W : nfirst - stripe_base) - 1u) >> 7;
                const unsigned long long mm = (hi >= 63u ? ~0ull : ((1ull << (hi + 1u)) - 1ull)) & ~((1ull << lo) - 1ull);
                return (coarse & mm) != 0ull;
            };
            for (uint32_t ta = 1; ta < td; ta += 2u) {
                const uint32_t tb = ta + 1u;
                const bool have_b = tb < td;
                uint32_t a_meta, a_first, a_word, a_next, b_meta = 0, b_first = 0, b_word = 0, b_next = 0xFFFFFFFFu;
                uint32_t a_pos = 0, b_pos = 0;
                const uint32_t nba = re[ta * SPG + sl] - rs[ta * SPG + sl];
                const uint32_t nbb = have_b ? re[tb * SPG + sl] - rs[tb * SPG + sl] : 0u;
                if (ta == 1u) {
                    a_meta = ha_meta; a_first = ha_first; a_word = ha_word; a_next = ha_next; a_pos = ha_pos;
                    b_meta = hb_meta; b_first = hb_first; b_word = hb_word; b_next = hb_next; b_pos = hb_pos;
                } else {
                    a_meta = a_first = a_word = 0; a_next = 0xFFFFFFFFu;
                    if (TALLY) { cn_hdr += XGM_SU(nba + nbb); }
                    const uint32_t rba = rs[ta * SPG + sl];
                    if (lane < nba) {
                        a_meta = seg.blk_meta[rba + lane]; a_first = seg.blk_first[rba + lane]; a_word = seg.blk_word[rba + lane];
                        a_next = lane + 1u < nba ? seg.blk_first[rba + lane + 1u] : 0xFFFFFFFFu;
                        if (PHRASE) a_pos = seg.blk_pos[rba + lane];
                    }
                    if (have_b) {
                        const uint32_t rbb = rs[tb * SPG + sl];
                        if (lane < nbb) {
                            b_meta = seg.blk_meta[rbb + lane]; b_first = seg.blk_first[rbb + lane]; b_word = seg.blk_word[rbb + lane];
                            b_next = lane + 1u < nbb ? seg.blk_first[rbb + lane + 1u] : 0xFFFFFFFFu;
                            if (PHRASE) b_pos = seg.blk_pos[rbb + lane];
                        }
                    }
                }
                uint64_t mask_a = __ballot(lane < nba && bucket_need(a_first, a_next));
                uint64_t mask_b = __ballot(have_b && lane < nbb && bucket_need(b_first, b_next));
                const uint64_t tba = tbase(ta), tbb = have_b ? tbase(tb) : 0ull;
                TabT* row_a = c_w + (size_t)ta * CAND;
                TabT* row_b = c_w + (size_t)tb * CAND;
                uint32_t* prow_a = c_pos + (size_t)ta * CAND;
                uint32_t* prow_b = c_pos + (size_t)tb * CAND;
                while (mask_a | mask_b) {
                    uint32_t jj[8], n_a = 0, n_b = 0;
                    Words4 pv[8];
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) {
                        jj[u] = 0; pv[u] = Words4{0, 0, 0, 0};
                        if (mask_a) {
                            jj[u] = (uint32_t)__builtin_ctzll(mask_a);
                            mask_a &= mask_a - 1u;
                            const uint32_t bm = __builtin_amdgcn_readlane(a_meta, jj[u]);
                            if (TALLY) { cn_blkw += payload_words(bm) - 2u; }
                            const uint32_t bwd = __builtin_amdgcn_readlane(a_word, jj[u]);      /* read by every lane, not under the per-lane test */
                            if (lane * 4u < payload_words(bm)) pv[u] = *reinterpret_cast<const Words4*>(seg.words + tba + bwd + lane * 4u);
                            n_a = u + 1u;
                        }
                    }
#pragma unroll
                    for (uint32_t u = 4; u < 8u; ++u) {
                        jj[u] = 0; pv[u] = Words4{0, 0, 0, 0};
                        if (mask_b) {
                            jj[u] = (uint32_t)__builtin_ctzll(mask_b);
                            mask_b &= mask_b - 1u;
                            const uint32_t bm = __builtin_amdgcn_readlane(b_meta, jj[u]);
                            if (TALLY) { cn_blkw += payload_words(bm) - 2u; }
                            const uint32_t bwd = __builtin_amdgcn_readlane(b_word, jj[u]);
                            if (lane * 4u < payload_words(bm)) pv[u] = *reinterpret_cast<const Words4*>(seg.words + tbb + bwd + lane * 4u);
                            n_b = u - 3u;
                        }
                    }
#pragma unroll
                    for (uint32_t u = 0; u < 8u; ++u) {
                        const bool is_b = u >= 4u;
                        if (is_b ? (u - 4u < n_b) : (u < n_a)) {
                            const uint32_t bmeta = __builtin_amdgcn_readlane(is_b ? b_meta : a_meta, jj[u]);
                            const uint32_t bfirst = __builtin_amdgcn_readlane(is_b ? b_first : a_first, jj[u]);
                            TabT* row = is_b ? row_b : row_a;
                            uint32_t* prow = is_b ? prow_b : prow_a;
                            const uint32_t bpos = PHRASE ? __builtin_amdgcn_readlane(is_b ? b_pos : a_pos, jj[u]) : 0u;
                            if (lane * 4u < payload_words(bmeta)) {
                                stage[lane * 4u] = pv[u].a; stage[lane * 4u + 1] = pv[u].b; stage[lane * 4u + 2] = pv[u].c; stage[lane * 4u + 3] = pv[u].d;
                            }
                            wave_lds_fence();
                            DecodedPair r = unpack_staged<PHRASE>(stage, bfirst, bmeta, lane);
                            wave_lds_fence();
                            if (r.v0) {
                                const uint32_t sl0 = r.d0 - stripe_base, wd = sl0 >> 5, bit = sl0 & 31u;
                                const uint32_t bm = bitmap[wd];
                                if ((bm >> bit) & 1u) {
                                    const uint32_t ord = (uint32_t)rankw[wd] + (uint32_t)__popc(bm & ((1u << bit) - 1u));
                                    row[ord] = (TabT)(r.w0 + 1u);
                                    if (PHRASE) prow[ord] = bpos + r.p0;
                                }
                            }
                            if (r.v1) {
                                const uint32_t sl1 = r.d1 - stripe_base, wd = sl1 >> 5, bit = sl1 & 31u;
                                const uint32_t bm = bitmap[wd];
                                if ((bm >> bit) & 1u) {
                                    const uint32_t ord = (uint32_t)rankw[wd] + (uint32_t)__popc(bm & ((1u << bit) - 1u));
                                    row[ord] = (TabT)(r.w1 + 1u);
                                    if (PHRASE) prow[ord] = bpos + r.p1;
                                }
                            }
                        }
                    }
                }
            }
            wave_lds_fence();

            /* ---- P3b: dense other terms [td, T): O(1) probes of their containers ---- */
            if (td < T) probe_dense(td, n_c);

            /* ---- P3c: right-hand terms without containers: the blocks whose buckets hold a candidate ---- */
            if (SIDED && sparse_neg) scatter_sparse_rhs(sl, coarse, 0u);
            wave_lds_fence();

            /* headers of the next active stripe: in flight while this one is scored */
            if (cb + CHUNKB >= r0e && sl_next < n_local) issue_headers(sl_next);

            /* ---- P4 ---- */
            score_candidates(n_c, true);
            for (uint32_t i = lane; i < NW; i += 64u) { bitmap[i] = 0; rankw[i] = 0xFFFFu; }
            wave_lds_fence();
        }
        sl = sl_next;
    }

    /* ---- unit epilogue ---- */
    if (MAYBE) wave_topk_sort_m(tk_w, tk_d, tk_m, cap, lane); else wave_topk_sort(tk_w, tk_d, cap, lane);
    for (int sh = 32; sh > 0; sh >>= 1) matches += (unsigned long long)__shfl_xor((long long)matches, sh);
    if (PHRASE && TALLY) { for (int sh = 32; sh > 0; sh >>= 1) cn_pos += (unsigned long long)__shfl_xor((long long)cn_pos, sh); }
    const bool any_pruned = PHRASE && __ballot(pos_pruned) != 0ull;
    const uint32_t n_out = tkn < k ? tkn : k;
    xgm_cand* out = cand_out + (size_t)wk.slot * k_stride;
    const bool through = fuse != nullptr;                          /* the launch finishes its queries itself (xgm_unit_finish.h) */
    for (uint32_t i = lane; i < n_out; i += 64u) {
        xgm_cand c;
        c.wbits = tk_w[i]; c.did = tk_d[i]; c.subqs = MAYBE ? (uint32_t)tk_m[i] : (uint32_t)__popc(q.score_mask);   /* plain: the weighted leaves all match */
        xgm_store_cand(through, &out[i], c);
    }
    if (lane == 0) {
        xgm_group_hdr h;
        h.matches = matches | (PHRASE && any_pruned ? XGM_MATCHES_LOWER_BOUND : 0ull); h.n_cand = n_out; h.pad = TALLY ? q_cands : 0u;
        h.t_start = t_unit_start; h.t_end = __builtin_readcyclecounter();
        h.c_pos = cn_pos; h.c_bmp_words = cn_bmpw; h.c_probes = cn_probe; h.c_blk_words = cn_blkw; h.c_hdrs = cn_hdr;
        h.c_doclen = cn_dl; h.c_aux_words = cn_aux; h.c_probes_raw = cn_probe_raw; h.c_doclen_raw = cn_dl_raw; h.c_pad[0] = (uint32_t)(ph_a >> 6); h.c_pad[1] = (uint32_t)(ph_b >> 6);
        xgm_store_hdr(through, &ghdr_out[wk.slot], h);
    }
    if (through) xgm_unit_arrive<MAYBE>(*fuse, wk.qi, k, (uint32_t)__popc(q.score_mask), tk_w, tk_d, tk_m, cap, cand_out, ghdr_out, k_stride, lane,
                           [&]() { if (MAYBE) wave_topk_sort_m(tk_w, tk_d, tk_m, cap, lane); else wave_topk_sort(tk_w, tk_d, cap, lane); });
#undef XGM_SU
}

/* ---------------------------------------------------------------- merge kernel --------------- */

constexpr uint32_t kMergeSel = 1024;   /* survivors the selection path of the merge ranks by counting */
#ifndef XGM_MERGE_TIMERS
#define XGM_MERGE_TIMERS 0
#endif
__device__ unsigned long long g_merge_cycles[8];   /* diagnostics (-DXGM_MERGE_TIMERS=1): summed section cycles of thread 0 of every workgroup */
__device__ unsigned long long g_merge_max[8];      /* ... and the largest single section of any workgroup of a full batch (the kernel's duration is its slowest workgroup) */
#define MG_PH(i) do { if (XGM_MERGE_TIMERS && tid == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); atomicAdd(&g_merge_cycles[i], n_ - mg_t); \
        if (gridDim.x >= 64u) atomicMax(&g_merge_max[i], ((n_ - mg_t) << 20) | ((unsigned long long)(goff[qi + 1] - goff[qi]) << 4) | (i)); mg_t = n_; } } while (0)

/* One workgroup per query.  Sources: n_src candidate lists of up to k_stride entries
 * (groups of one shard, or shards after the all-gather).  did_mul/did_add remap shard-local
 * docids: global = (local - 1) * n_shards + shard + 1 (multi.h:69-73) when unshard != 0. */
__global__ __launch_bounds__(XGM_WG) void xgm_merge_kernel(const xgm_cand* __restrict__ cand, const xgm_group_hdr* __restrict__ ghdr,
                                                            const uint32_t* __restrict__ goff, uint32_t k_stride_in, const uint32_t* __restrict__ kq,
                                                            uint32_t cap, uint32_t k_stride_out, xgm_hit* __restrict__ hits,
                                                            xgm_result_hdr* __restrict__ hdrs, const double* __restrict__ max_possible,
                                                            const uint32_t* __restrict__ row_of) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, qi = blockIdx.x;
    /* a heterogeneous batch runs one launch per kernel class: query qi of this launch is row row_of[qi] of the caller's batch */
    const uint32_t orow = row_of ? row_of[qi] : qi;
    unsigned long long mg_t = XGM_MERGE_TIMERS ? __builtin_readcyclecounter() : 0ull;
    const unsigned long long mg_w0 = XGM_MERGE_TIMERS ? wall_clock64() : 0ull;
    if (XGM_MERGE_TIMERS && tid == 0 && gridDim.x >= 64u && qi == 0) __hip_atomic_store(&g_merge_max[5], mg_w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    TopK tk;
    tk.w = reinterpret_cast<uint64_t*>(smem);
    tk.d = reinterpret_cast<uint32_t*>(smem + (size_t)cap * 8);
    tk.m = reinterpret_cast<uint32_t*>(smem + (size_t)cap * 12);
    tk.cap = cap;
    unsigned long long& matches = *reinterpret_cast<unsigned long long*>(smem + (size_t)cap * 16);
    uint32_t& fill = *reinterpret_cast<uint32_t*>(smem + (size_t)cap * 16 + 8);
    uint32_t& lower_only = *reinterpret_cast<uint32_t*>(smem + (size_t)cap * 16 + 12);      /* some unit dropped candidates unweighed against positions */
    if (tid == 0) { fill = 0; matches = 0; lower_only = 0; }
    __syncthreads();
    /* gather: unit u of the query owns the fixed window [u * k_in, (u+1) * k_in) of the sort buffer
     * (k_in = its candidate stride); empty places get the sentinel.  Fully parallel: no prefix sums.
     * Two latency-bound steps kept short: the units' headers once (their candidate counts parked in LDS — the selection
     * arrays are not in use yet), then the candidates four loads at a time (a query cut into 512 units has 5 120 of them:
     * one dependent load pair per entry used to make the busiest query's merge 60 us long). */
    const uint32_t g0 = goff[qi], n_src = goff[qi + 1] - g0;
    uint32_t* n_cand_s = reinterpret_cast<uint32_t*>(smem + (size_t)cap * 16 + 64);        /* [n_src] <= cap / k_in entries; the selection arrays start here later */
    const uint32_t n_src_fit = n_src < 2u * kMergeSel ? n_src : 0u;                          /* (kMergeSel * 16 bytes = room for 4 * kMergeSel counts; be conservative) */
    for (uint32_t u = tid; u < n_src_fit; u += XGM_WG) {
        const xgm_group_hdr h = ghdr[g0 + u];
        n_cand_s[u] = h.n_cand;
        atomicAdd(&matches, (unsigned long long)(h.matches & ~XGM_MATCHES_LOWER_BOUND)); atomicAdd(&fill, h.n_cand);
        if (h.matches & XGM_MATCHES_LOWER_BOUND) atomicOr(&lower_only, 1u);
    }
    __syncthreads();
    MG_PH(0);
    if (n_src_fit) {
        for (uint32_t x0 = tid; x0 < cap; x0 += 4u * XGM_WG) {
            xgm_cand c[4];
            bool have[4];
#pragma unroll
            for (uint32_t v = 0; v < 4u; ++v) {
                const uint32_t x = x0 + v * XGM_WG;
                const uint32_t u = x / k_stride_in, j = x - u * k_stride_in;
                have[v] = x < cap && u < n_src && j < n_cand_s[u];
                if (have[v]) c[v] = cand[(size_t)(g0 + u) * k_stride_in + j];
            }
#pragma unroll
            for (uint32_t v = 0; v < 4u; ++v) {
                const uint32_t x = x0 + v * XGM_WG;
                if (x < cap) {
                    tk.w[x] = have[v] ? c[v].wbits : 0ull; tk.d[x] = have[v] ? c[v].did : 0xFFFFFFFFu; tk.m[x] = have[v] ? c[v].subqs : 0xFFFFFFFFu;
                }
            }
        }
    } else {
    for (uint32_t x = tid; x < cap; x += XGM_WG) {
        const uint32_t u = x / k_stride_in, j = x - u * k_stride_in;
        uint64_t w = 0; uint32_t d = 0xFFFFFFFFu, m = 0xFFFFFFFFu;
        if (u < n_src) {
            const xgm_group_hdr h = ghdr[g0 + u];
            if (j < h.n_cand) {
                const xgm_cand c = cand[(size_t)(g0 + u) * k_stride_in + j];
                w = c.wbits; d = c.did; m = c.subqs;
            }
            if (j == 0) {
                atomicAdd(&matches, (unsigned long long)(h.matches & ~XGM_MATCHES_LOWER_BOUND)); atomicAdd(&fill, h.n_cand);
                if (h.matches & XGM_MATCHES_LOWER_BOUND) atomicOr(&lower_only, 1u);
            }
        }
        tk.w[x] = w; tk.d[x] = d; tk.m[x] = m;
    }
    }
    __syncthreads();
    MG_PH(1);
    const uint32_t k = kq[qi];
    const uint32_t n = fill < k ? fill : k;
    /* Selection instead of a full sort: take the best r = ceil(k / units) candidates of every unit (each
     * unit's list is sorted): those are >= k candidates, so their k-th best bounds the final k-th from
     * below and everything worse is dropped; the few survivors are ranked by counting.  (weight, docid)
     * keys are distinct, so ranks are too. */
    if (k > 0 && fill > 0 && n_src > 0) {
        uint64_t* sel_w = reinterpret_cast<uint64_t*>(smem + (size_t)cap * 16 + 64);
        uint32_t* sel_d = reinterpret_cast<uint32_t*>(smem + (size_t)cap * 16 + 64 + (size_t)kMergeSel * 8);
        uint32_t* sel_m = sel_d + kMergeSel;
        uint64_t& thr_w = *reinterpret_cast<uint64_t*>(smem + (size_t)cap * 16 + 16);
        uint32_t& thr_d = *reinterpret_cast<uint32_t*>(smem + (size_t)cap * 16 + 24);
        uint32_t& n_sel = *reinterpret_cast<uint32_t*>(smem + (size_t)cap * 16 + 28);
        uint64_t& best_w = *reinterpret_cast<uint64_t*>(smem + (size_t)cap * 16 + 32);
        uint32_t& best_m = *reinterpret_cast<uint32_t*>(smem + (size_t)cap * 16 + 40);
        if (tid == 0) { thr_w = 0; thr_d = 0xFFFFFFFFu; n_sel = 0; best_w = 0; best_m = 0; }
        __syncthreads();
        const uint32_t rows = (k + n_src - 1u) / n_src;               /* <= k <= k_stride_in */
        const uint32_t n_h = n_src * rows;
        if (rows == 1u && n_h > 128u && n_h <= kMergeSel) {
            /* a query cut into hundreds of units: ranking every head against every other is quadratic (262 k comparisons at 512
             * units: the busiest query's workgroup then IS the kernel's duration) — sort the heads instead (bitonic, best first,
             * in the selection arrays, which are free until the threshold is known) and read the k-th off */
            uint32_t P = 256u;
            while (P < n_h) P <<= 1;
            for (uint32_t i = tid; i < P; i += XGM_WG) {
                sel_w[i] = i < n_h ? tk.w[i * k_stride_in] : 0ull;
                sel_d[i] = i < n_h ? tk.d[i * k_stride_in] : 0xFFFFFFFFu;
            }
            for (uint32_t size = 2; size <= P; size <<= 1) {
                for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                    __syncthreads();
                    for (uint32_t i = tid; i < (P >> 1); i += XGM_WG) {
                        const uint32_t lo = 2u * i - (i & (stride - 1u)), hi = lo + stride;
                        const bool asc = ((lo & size) == 0);
                        const uint64_t aw = sel_w[lo], bw = sel_w[hi];
                        const uint32_t ad = sel_d[lo], bd = sel_d[hi];
                        const bool swap = asc ? cand_before(bw, bd, aw, ad) : cand_before(aw, ad, bw, bd);
                        if (swap) { sel_w[lo] = bw; sel_w[hi] = aw; sel_d[lo] = bd; sel_d[hi] = ad; }
                    }
                }
            }
            __syncthreads();
            if (tid == 0) { thr_w = sel_w[k - 1u]; thr_d = sel_d[k - 1u]; }
        } else {
        for (uint32_t t = tid; t < n_h; t += XGM_WG) {
            const uint32_t xt = (t / rows) * k_stride_in + (t % rows);
            const uint64_t hw = tk.w[xt];
            const uint32_t hd = tk.d[xt];
            uint32_t rank = 0;
            for (uint32_t u = 0; u < n_src; ++u)
                for (uint32_t j = 0; j < rows; ++j)
                    rank += cand_before(tk.w[u * k_stride_in + j], tk.d[u * k_stride_in + j], hw, hd) ? 1u : 0u;
            if (rank == k - 1u) { thr_w = hw; thr_d = hd; }        /* empty places tie on the sentinel: same value from all writers */
        }
        }
        __syncthreads();
        MG_PH(2);
        const uint64_t tw = thr_w;
        const uint32_t td = thr_d;
        for (uint32_t x = tid; x < cap; x += XGM_WG) {
            const uint64_t w = tk.w[x];
            const uint32_t d = tk.d[x];
            if (d != 0xFFFFFFFFu && !cand_before(tw, td, w, d)) {
                const uint32_t p = atomicAdd(&n_sel, 1u);
                if (p < kMergeSel) { sel_w[p] = w; sel_d[p] = d; sel_m[p] = tk.m[x]; }
            }
        }
        __syncthreads();
        MG_PH(3);
        const uint32_t ns = n_sel;
        if (ns <= kMergeSel) {
            for (uint32_t i = tid; i < ns; i += XGM_WG) {
                const uint64_t w = sel_w[i];
                const uint32_t d = sel_d[i];
                uint32_t rank = 0;
                for (uint32_t j = 0; j < ns; ++j) rank += cand_before(sel_w[j], sel_d[j], w, d) ? 1u : 0u;
                if (rank < n) {
                    xgm_hit hit;
                    hit.docid = d; hit.subqs_matched = sel_m[i]; hit.weight = __longlong_as_double((long long)w);
                    hits[(size_t)orow * k_stride_out + rank] = hit;
                }
                if (rank == 0u) { best_w = w; best_m = sel_m[i]; }
            }
            __syncthreads();
            MG_PH(4);
            if (XGM_MERGE_TIMERS && tid == 0 && gridDim.x >= 64u) { const unsigned long long w1_ = wall_clock64(); atomicMax(&g_merge_max[6], w1_ - __hip_atomic_load(&g_merge_max[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); atomicMax(&g_merge_max[7], w1_ - mg_w0); }
            if (tid == 0) {
                xgm_result_hdr r;
                r.n_hits = n;
                r.max_weight_subqs_matched = best_m;
                r.matches_exact = matches | (lower_only ? XGM_MATCHES_LOWER_BOUND : 0ull);
                r.max_attained = __longlong_as_double((long long)best_w);
                r.max_possible = max_possible ? max_possible[qi] : 0.0;
                hdrs[orow] = r;
            }
            return;
        }
    }
    if (fill == 0) {
        /* a query without a single candidate (a conjunction that matches nothing): nothing to rank — sorting the sentinels of
         * the whole buffer made exactly these workgroups the slowest of the launch (round 3: 119 us of a 530 us step) */
        if (tid == 0) {
            xgm_result_hdr r;
            r.n_hits = 0;
            r.max_weight_subqs_matched = 0u;
            r.matches_exact = matches | (lower_only ? XGM_MATCHES_LOWER_BOUND : 0ull);
            r.max_attained = 0.0;
            r.max_possible = max_possible ? max_possible[qi] : 0.0;
            hdrs[orow] = r;
        }
        return;
    }
    topk_sort(tk, tid);
    for (uint32_t i = tid; i < n; i += XGM_WG) {
        xgm_hit hit;
        hit.docid = tk.d[i]; hit.subqs_matched = tk.m[i]; hit.weight = __longlong_as_double((long long)tk.w[i]);
        hits[(size_t)orow * k_stride_out + i] = hit;
    }
    if (tid == 0) {
        xgm_result_hdr r;
        r.n_hits = n;
        r.max_weight_subqs_matched = fill ? tk.m[0] : 0u;
        r.matches_exact = matches | (lower_only ? XGM_MATCHES_LOWER_BOUND : 0ull);
        r.max_attained = fill ? __longlong_as_double((long long)tk.w[0]) : 0.0;
        r.max_possible = max_possible ? max_possible[qi] : 0.0;
        hdrs[orow] = r;
    }
}

/* Shard merge after the all-gather: sources are whole result lists (xgm_hit) of each shard. */
__global__ __launch_bounds__(XGM_WG) void xgm_merge_shards_kernel(const xgm_hit* __restrict__ all_hits, const xgm_result_hdr* __restrict__ all_hdrs,
                                                                   uint32_t n_shards, uint32_t nq, uint32_t k_stride, const uint32_t* __restrict__ kq,
                                                                   uint32_t cap, xgm_hit* __restrict__ hits, xgm_result_hdr* __restrict__ hdrs,
                                                                   uint32_t unshard, const uint32_t* __restrict__ row_of,
                                                                   unsigned long long hit_shard_bytes, unsigned long long hdr_shard_bytes) {
    /* unshard == 0: the sources are the PARTS of one shard's query (same docid space; xgm_launch_merge_parts).
     * hit_shard_bytes / hdr_shard_bytes: distance between two shards' arrays — nq * k_stride hits / nq headers when the shards' arrays are
     * gathered one after the other; one packed record ([nq][k_stride] hits then [nq] headers) when ONE all-gather brought both */
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, qi = blockIdx.x;
    const uint32_t orow = row_of ? row_of[qi] : qi;
    TopK tk;
    tk.w = reinterpret_cast<uint64_t*>(smem);
    tk.d = reinterpret_cast<uint32_t*>(smem + (size_t)cap * 8);
    tk.m = reinterpret_cast<uint32_t*>(smem + (size_t)cap * 12);
    tk.cap = cap;
    unsigned long long& matches = *reinterpret_cast<unsigned long long*>(smem + (size_t)cap * 16);
    double& max_possible = *reinterpret_cast<double*>(smem + (size_t)cap * 16 + 8);
    double& max_attained = *reinterpret_cast<double*>(smem + (size_t)cap * 16 + 16);
    uint32_t& fill = *reinterpret_cast<uint32_t*>(smem + (size_t)cap * 16 + 24);
    uint32_t& base = *reinterpret_cast<uint32_t*>(smem + (size_t)cap * 16 + 28);
    uint32_t& max_subqs = *reinterpret_cast<uint32_t*>(smem + (size_t)cap * 16 + 32);
    if (tid == 0) { fill = 0; matches = 0; max_possible = 0.0; max_attained = 0.0; max_subqs = 0; }
    for (uint32_t i = tid; i < cap; i += XGM_WG) { tk.w[i] = 0; tk.d[i] = 0xFFFFFFFFu; tk.m[i] = 0xFFFFFFFFu; }
    __syncthreads();
    for (uint32_t sh = 0; sh < n_shards; ++sh) {
        const xgm_result_hdr h = reinterpret_cast<const xgm_result_hdr*>(reinterpret_cast<const unsigned char*>(all_hdrs) + (size_t)sh * hdr_shard_bytes)[qi];
        const xgm_hit* src = reinterpret_cast<const xgm_hit*>(reinterpret_cast<const unsigned char*>(all_hits) + (size_t)sh * hit_shard_bytes) + (size_t)qi * k_stride;
        if (tid == 0) {
            base = fill; fill += h.n_hits;
            matches = ((matches & ~XGM_MATCHES_LOWER_BOUND) + (h.matches_exact & ~XGM_MATCHES_LOWER_BOUND)) | ((matches | h.matches_exact) & XGM_MATCHES_LOWER_BOUND);
            /* MSet::Internal::merge_stats, mset.cc:376-395: max of max_possible / max_attained */
            if (h.max_possible > max_possible) max_possible = h.max_possible;
            if (h.max_attained > max_attained) { max_attained = h.max_attained; max_subqs = h.max_weight_subqs_matched; }
        }
        __syncthreads();
        for (uint32_t i = tid; i < h.n_hits; i += XGM_WG) {
            xgm_hit c = src[i];
            tk.w[base + i] = (uint64_t)__double_as_longlong(c.weight);
            tk.d[base + i] = unshard ? (c.docid - 1u) * n_shards + sh + 1u : c.docid;
            tk.m[base + i] = c.subqs_matched;
        }
        __syncthreads();
    }
    topk_sort(tk, tid);
    const uint32_t k = kq[qi];
    const uint32_t n = fill < k ? fill : k;
    for (uint32_t i = tid; i < n; i += XGM_WG) {
        xgm_hit hit;
        hit.docid = tk.d[i]; hit.subqs_matched = tk.m[i]; hit.weight = __longlong_as_double((long long)tk.w[i]);
        hits[(size_t)orow * k_stride + i] = hit;
    }
    if (tid == 0) {
        xgm_result_hdr r;
        r.n_hits = n; r.max_weight_subqs_matched = max_subqs; r.matches_exact = matches;
        r.max_attained = max_attained; r.max_possible = max_possible;
        hdrs[orow] = r;
    }
}

/* ---------------------------------------------------------------- decode-only kernel ---------- */

/* One wave per block of the term; writes postings at their ordinal.  ord_base[b - b0] = ordinal of
 * the block's first posting (exclusive scan of the block counts, computed by the caller). */
__global__ __launch_bounds__(XGM_WG) void xgm_decode_kernel(xgm_seg_dev seg, uint32_t term_id, uint32_t b0, uint32_t nblk,
                                                             const uint64_t* __restrict__ ord_base, uint32_t* __restrict__ out_did,
                                                             uint32_t* __restrict__ out_wdf) {
    __shared__ uint32_t stage_all[XGM_WAVES * kStageWords];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* stage = stage_all + wave * kStageWords;
    for (uint32_t i = blockIdx.x * XGM_WAVES + wave; i < nblk; i += gridDim.x * XGM_WAVES) {
        const uint32_t b = b0 + i;
        const uint32_t* payload = seg.words + seg.term_word[term_id] + seg.blk_word[b];
        DecodedPair r = decode_block<false>(payload, seg.blk_first[b], seg.blk_meta[b], stage, lane);
        uint64_t o = ord_base[i] + 2u * lane;
        if (r.v0) { out_did[o] = r.d0; out_wdf[o] = r.w0; }
        if (r.v1) { out_did[o + 1] = r.d1; out_wdf[o + 1] = r.w1; }
    }
}

/* ---------------------------------------------------------------- value sorts -- */

/* xgm_match_kernel for a search under Enquire::set_sort_by_value / _value_then_relevance / _relevance_then_value (reference
 * api/enquire.cc; comparison functions matcher/msetcmp.cc:64-101): the unit's best k documents under
 *     mode 1 (value):                   value,  docid
 *     mode 2 (value then relevance):    value,  weight, docid
 *     mode 3 (relevance then value):    weight, value,  docid
 * where "value" compares the documents' ordinals in a device column (ord[docid] = 0 for no value, else 1 + the rank of the
 * document's value among the slot's distinct values: xgm_glass_export_column) — ascending, or descending with `reverse` — weights
 * descend and docids ascend.  A candidate is ranked by two 64-bit keys, larger first: (kw, kx) = (ordinal key, weight bits), or
 * (weight bits, ordinal key) in mode 3; mode 1 carries the weight in kx without comparing it.  Every matching document is
 * weighed: with the value leading there is no weight to prune by, and the reference reports the best weight of the WHOLE match
 * (ProtoMSet::update_max_weight sees every document, protomset.h:174-183, 249-283) — published per unit in the header (c_pos =
 * weight bits, c_pad = {docid, weighted leaves} of the first document that attains it).
 * mode 4: relevance alone (weight, docid) — for a search that only collapses.
 * cord != NULL: Enquire::set_collapse_key(slot, cmax) by the ordinals of that slot's column (collapse_prune above).
 * spy_counts != NULL: also a Xapian::ValueCountMatchSpy (api/matchspy.cc:307-313) — one count per ordinal of the column spy_ord,
 * incremented for every matching document (a value-led sort shows the spies every match, protomset.h:268-275).
 * The kernel is xgm_match_kernel's own text compiled a second time (xgm_match_body.inc): every query shape that kernel handles. */
__device__ __forceinline__ bool sorted_before(uint64_t aw, uint64_t ax, uint32_t ad, uint64_t bw, uint64_t bx, uint32_t bd, bool use_x) {
    if (aw != bw) return aw > bw;
    if (use_x && ax != bx) return ax > bx;
    return ad < bd;
}

struct SortedExt {                    /* LDS behind the ordinary layout of the match kernel */
    uint64_t theta_x;
    unsigned long long max_w;
    uint32_t max_d, max_m;
    uint32_t survivors, pad;
};

/* topk_sort over (w, x, d): unused entries hold (0, 0, UINT32_MAX), which sorts last; c (the candidates' collapse ordinals) moves along */
__device__ void topk_sort_sorted(const TopK& tk, uint64_t* x, uint32_t* c, uint32_t tid, bool use_x) {
    for (uint32_t size = 2; size <= tk.cap; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (uint32_t i = tid; i < (tk.cap >> 1); i += XGM_WG) {
                const uint32_t lo = 2u * i - (i & (stride - 1u)), hi = lo + stride;
                const bool asc = ((lo & size) == 0);
                const uint64_t aw = tk.w[lo], bw = tk.w[hi], ax = x[lo], bx = x[hi];
                const uint32_t ad = tk.d[lo], bd = tk.d[hi];
                const bool swap = asc ? sorted_before(bw, bx, bd, aw, ax, ad, use_x) : sorted_before(aw, ax, ad, bw, bx, bd, use_x);
                if (swap) {
                    tk.w[lo] = bw; tk.w[hi] = aw; x[lo] = bx; x[hi] = ax;
                    tk.d[lo] = bd; tk.d[hi] = ad;
                    const uint32_t am = tk.m[lo], bm = tk.m[hi];
                    tk.m[lo] = bm; tk.m[hi] = am;
                    const uint32_t ac = c[lo], bc = c[hi];
                    c[lo] = bc; c[hi] = ac;
                }
            }
        }
    }
    __syncthreads();
}

/* Enquire::set_collapse_key on a sorted buffer (best first): an entry with a collapse key (ordinal != 0) behind cmax better entries of
 * the same key becomes a sentinel; the caller sorts again.  A document evicted here by documents of its own unit is evicted in the
 * whole match too, and a document inside the global collapsed top k is inside its unit's (per key, the global survivors ahead of it
 * are at least as many as the unit's) — so the units can collapse independently and the merge collapses once more.  Quadratic in
 * the buffer (<= a few hundred entries): first version. */
__device__ void collapse_prune(const TopK& tk, uint64_t* x, uint32_t* c, uint32_t n, uint32_t cmax, uint32_t tid) {
    bool drop[8];                                   /* the buffer holds at most 8 x XGM_WG entries (the host checks) */
    uint32_t nd = 0;
    for (uint32_t i = tid; i < n; i += XGM_WG, ++nd) {
        const uint32_t key = c[i];
        uint32_t ahead = 0;
        if (key) for (uint32_t j = 0; j < i; ++j) ahead += c[j] == key ? 1u : 0u;
        drop[nd & 7u] = key != 0u && ahead >= cmax;
    }
    __syncthreads();
    nd = 0;
    for (uint32_t i = tid; i < n; i += XGM_WG, ++nd)
        if (drop[nd & 7u]) { tk.w[i] = 0; x[i] = 0; tk.d[i] = 0xFFFFFFFFu; tk.m[i] = 0xFFFFFFFFu; c[i] = 0; }
    __syncthreads();
}

#define XGM_BODY_SORTED 1
#include "xgm_match_body.inc"
#undef XGM_BODY_SORTED

size_t match_smem_bytes(uint32_t W, uint32_t T, bool phrase, uint32_t cap, size_t tab_elem, uint32_t spg) {
    size_t off = 0;
    off += (size_t)cap * 8;
    off += (sizeof(Ctrl) + 15) & ~(size_t)15;
    off += (size_t)T * W * tab_elem;
    off += phrase ? (size_t)T * W * 4 : 0;
    off += (size_t)cap * 4 * 2;
    off += (size_t)XGM_WAVES * kStageWords * 4;
    off += (size_t)W * 2;
    off += (size_t)2 * T * spg * 4;
    return (off + 15) & ~(size_t)15;
}

}  // namespace

/* ---------------------------------------------------------------- launchers ------------------- */

/* hipFuncSetAttribute(MaxDynamicSharedMemorySize) only when a launch needs more than any before it
 * (the call costs microseconds that the single-query latency path would pay three times). */
#include <atomic>
template <class K>
static int ensure_dyn_smem(K kern, size_t smem, std::atomic<size_t>& seen) {
    if (smem <= seen.load(std::memory_order_relaxed)) return 0;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return xgm_launch_error("hipFuncSetAttribute", (int)e, hipGetErrorString(e));
    seen.store(smem, std::memory_order_relaxed);
    return 0;
}

#define XGM_HIP_CHECK(expr)                                                                     \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return xgm_launch_error(#expr, (int)e_, hipGetErrorString(e_));  \
    } while (0)

size_t xgm_match_smem_bytes(uint32_t stripe_bits, uint32_t tab_terms, bool phrase, uint32_t cap, bool wide, uint32_t stripes_per_group) {
    return match_smem_bytes(1u << stripe_bits, tab_terms, phrase, cap, wide ? 2 : 1, stripes_per_group);
}

int xgm_launch_match(const xgm_match_launch& L, hipStream_t stream) {
    dim3 grid(L.n_work), block(XGM_WG);
    if (L.sub_bits > L.seg.stripe_bits || L.seg.stripe_bits - L.sub_bits < 8u) return xgm_launch_error("match kernel", 0, "bad sub-stripe width");
    const size_t smem = xgm_match_smem_bytes(L.seg.stripe_bits - L.sub_bits, L.tab_terms, L.phrase, L.cap, L.wide, L.stripes_per_group);
    if (smem > 160u * 1024u) return xgm_launch_error("match kernel LDS budget", 0, "LDS request exceeds 160 KiB");
#define XGM_LAUNCH(TT, PH)                                                                                   \
    do {                                                                                                     \
        auto kern = xgm_match_kernel<TT, PH>;                                                                \
        static std::atomic<size_t> seen{0};                                                                  \
        if (int rc_ = ensure_dyn_smem(kern, smem, seen)) return rc_;                                         \
        hipLaunchKernelGGL(kern, grid, block, smem, stream, L.seg, L.queries, L.work, L.stripes_per_group,     \
                           L.tab_terms, L.cap, L.k_stride, L.cand, L.ghdr, L.sub_bits);                      \
    } while (0)
    if (L.wide) { if (L.phrase) XGM_LAUNCH(uint16_t, true); else XGM_LAUNCH(uint16_t, false); }
    else { if (L.phrase) XGM_LAUNCH(uint8_t, true); else XGM_LAUNCH(uint8_t, false); }
#undef XGM_LAUNCH
    XGM_HIP_CHECK(hipGetLastError());
    return 0;
}

size_t xgm_match_sorted_smem_bytes(uint32_t stripe_bits, uint32_t tab_terms, bool phrase, uint32_t cap, bool wide, uint32_t stripes_per_group) {
    return match_smem_bytes(1u << stripe_bits, tab_terms, phrase, cap, wide ? 2 : 1, stripes_per_group) + (size_t)cap * 8 + ((sizeof(SortedExt) + 15) & ~(size_t)15) + (size_t)cap * 4;
}

int xgm_launch_match_sorted(const xgm_match_launch& L, const uint32_t* ord, uint32_t mode, uint32_t reverse, const uint32_t* spy_ord, uint32_t* spy_counts,
                            const uint32_t* cord, uint32_t cmax, xgm_cand_sorted* cand, hipStream_t stream,
                            unsigned long long* all_keys, unsigned long long* all_vals, unsigned long long* all_count, unsigned long long all_cap) {
    dim3 grid(L.n_work), block(XGM_WG);
    if (L.sub_bits > L.seg.stripe_bits || L.seg.stripe_bits - L.sub_bits < 8u) return xgm_launch_error("sorted match kernel", 0, "bad sub-stripe width");
    const size_t smem = xgm_match_sorted_smem_bytes(L.seg.stripe_bits - L.sub_bits, L.tab_terms, L.phrase, L.cap, L.wide, L.stripes_per_group);
    if (smem > 160u * 1024u) return xgm_launch_error("sorted match kernel LDS budget", 0, "LDS request exceeds 160 KiB");
    if (mode < 1u || mode > 4u || (!ord && mode != 4u) || (spy_counts && !spy_ord) || (cord && cmax == 0u) || L.cap > 8u * XGM_WG ||
        (all_keys && (!all_vals || !all_count)))
        return xgm_launch_error("sorted match kernel", 0, "bad arguments");
#define XGM_LAUNCH(TT, PH)                                                                                   \
    do {                                                                                                     \
        auto kern = xgm_match_sorted_kernel<TT, PH>;                                                         \
        static std::atomic<size_t> seen{0};                                                                  \
        if (int rc_ = ensure_dyn_smem(kern, smem, seen)) return rc_;                                         \
        hipLaunchKernelGGL(kern, grid, block, smem, stream, L.seg, L.queries, L.work, L.stripes_per_group,     \
                           L.tab_terms, L.cap, L.k_stride, ord, mode, reverse, spy_ord, spy_counts, cord, cmax, cand, L.ghdr, \
                           all_keys, all_vals, all_count, all_cap, L.sub_bits, L.spy_stride);                  \
    } while (0)
    if (L.wide) { if (L.phrase) XGM_LAUNCH(uint16_t, true); else XGM_LAUNCH(uint16_t, false); }
    else { if (L.phrase) XGM_LAUNCH(uint8_t, true); else XGM_LAUNCH(uint8_t, false); }
#undef XGM_LAUNCH
    XGM_HIP_CHECK(hipGetLastError());
    return 0;
}

/* what xgm_dense_unit / xgm_flat_unit use of a wave's LDS slice (plan_batch checks it against xgm_andw_smem_bytes / XGM_WAVES before flagging a query) */
size_t xgm_body_wave_bytes(bool flat, bool phrase, uint32_t terms) { return flat ? flat_wave_bytes(phrase, terms) : dense_wave_bytes(phrase, terms); }

size_t xgm_and_smem_bytes(uint32_t stripe_bits, uint32_t tab_terms, uint32_t cap, bool wide, uint32_t stripes_per_group) {
    return and_smem_bytes(1u << stripe_bits, tab_terms, cap, wide ? 2 : 1, stripes_per_group);
}

static unsigned long long* g_phase_cycles = nullptr;        /* device buffer, diagnostics only */

int xgm_phase_cycles_fetch(unsigned long long* out8) {
    if (!g_phase_cycles) return -1;
    hipDeviceSynchronize();
    hipMemcpy(out8, g_phase_cycles, 64, hipMemcpyDeviceToHost);
    hipMemset(g_phase_cycles, 0, 64);
    return 0;
}

int xgm_launch_and(const xgm_match_launch& L, hipStream_t stream) {
    static const bool timing = getenv("XGM_PHASE_TIMING") != nullptr;
    if (timing && !g_phase_cycles) { hipMalloc((void**)&g_phase_cycles, 64); hipMemset(g_phase_cycles, 0, 64); }
    dim3 grid(L.n_work), block(XGM_WG);
    const size_t smem = xgm_and_smem_bytes(L.seg.stripe_bits, L.tab_terms, L.cap, L.wide, L.stripes_per_group);
    if (smem > 160u * 1024u) return xgm_launch_error("and kernel LDS budget", 0, "LDS request exceeds 160 KiB");
    static bool occ_printed = false;
    if (getenv("XGM_DEBUG_OCC") && !occ_printed) {
        occ_printed = true;
        int nb = -1;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)xgm_and_kernel<uint8_t>, XGM_WG, smem);
        hipFuncAttributes fa;
        hipFuncGetAttributes(&fa, (const void*)xgm_and_kernel<uint8_t>);
        fprintf(stderr, "XGM_OCC and_kernel: smem=%zu blocks/CU=%d regs=%d sharedStatic=%zu localSize=%zu maxDyn=%d n_work=%u\n", smem, nb, fa.numRegs,
                fa.sharedSizeBytes, fa.localSizeBytes, fa.maxDynamicSharedSizeBytes, L.n_work);
    }
    if (L.wide) {
        auto kern = xgm_and_kernel<uint16_t>;
        static std::atomic<size_t> seen{0};
        if (int rc_ = ensure_dyn_smem(kern, smem, seen)) return rc_;
        hipLaunchKernelGGL(kern, grid, block, smem, stream, L.seg, L.queries, L.work, L.stripes_per_group, L.tab_terms, L.cap, L.k_stride, L.cand, L.ghdr, g_phase_cycles);
    } else {
        auto kern = xgm_and_kernel<uint8_t>;
        static std::atomic<size_t> seen{0};
        if (int rc_ = ensure_dyn_smem(kern, smem, seen)) return rc_;
        hipLaunchKernelGGL(kern, grid, block, smem, stream, L.seg, L.queries, L.work, L.stripes_per_group, L.tab_terms, L.cap, L.k_stride, L.cand, L.ghdr, g_phase_cycles);
    }
    XGM_HIP_CHECK(hipGetLastError());
    return 0;
}

size_t xgm_andw_smem_bytes(uint32_t stripe_bits, uint32_t tab_terms, uint32_t cap, bool wide, uint32_t spg, bool phrase, bool sided) {
    return XGM_WAVES * andw_wave_bytes(1u << stripe_bits, tab_terms, cap, wide ? 2 : 1, spg, phrase, sided);
}

template <typename TabT, bool PHRASE, int SIDED, bool TALLY>
static int launch_andw_inst(const xgm_match_launch& L, size_t smem, hipStream_t stream) {
    const dim3 grid((L.n_work + XGM_WAVES - 1) / XGM_WAVES), block(XGM_WG);
    auto kern = xgm_andw_kernel<TabT, PHRASE, SIDED, TALLY>;
    static std::atomic<size_t> seen{0};
    if (int rc_ = ensure_dyn_smem(kern, smem, seen)) return rc_;
    XGM_LAUNCH_TIMED(L, kern, grid, block, smem, stream, L.seg, L.queries, L.work, L.n_work, L.stripes_per_group, L.tab_terms, L.cap, L.k_stride, L.cand, L.ghdr, L.hist, L.fuse);
    XGM_HIP_CHECK(hipGetLastError());
    return 0;
}

/* L.tally: the instantiation that also fills the traffic tallies of xgm_group_hdr (xgm_last_batch_traffic) — a
 * measurement build of the same code; the production instantiation carries none of it */
template <typename TabT, bool PHRASE, int SIDED>
static int launch_andw_variant(const xgm_match_launch& L, size_t smem, hipStream_t stream) {
    return L.tally ? launch_andw_inst<TabT, PHRASE, SIDED, true>(L, smem, stream) : launch_andw_inst<TabT, PHRASE, SIDED, false>(L, smem, stream);
}

int xgm_launch_andw(const xgm_match_launch& L, hipStream_t stream) {
    const size_t smem = xgm_andw_smem_bytes(L.seg.stripe_bits, L.tab_terms, L.cap, L.wide, L.stripes_per_group, L.phrase, L.sided == 2);
    if (smem > 160u * 1024u) return xgm_launch_error("andw kernel LDS budget", 0, "LDS request exceeds 160 KiB");
    if (L.phrase) return L.wide ? launch_andw_variant<uint16_t, true, 0>(L, smem, stream) : launch_andw_variant<uint8_t, true, 0>(L, smem, stream);
    if (L.sided == 2) return L.wide ? launch_andw_variant<uint16_t, false, 2>(L, smem, stream) : launch_andw_variant<uint8_t, false, 2>(L, smem, stream);
    if (L.sided == 1) return L.wide ? launch_andw_variant<uint16_t, false, 1>(L, smem, stream) : launch_andw_variant<uint8_t, false, 1>(L, smem, stream);
    return L.wide ? launch_andw_variant<uint16_t, false, 0>(L, smem, stream) : launch_andw_variant<uint8_t, false, 0>(L, smem, stream);
}

int xgm_launch_andw_all(const xgm_match_launch& L, const xgm_all_out& out, hipStream_t stream) {
    if (L.phrase || L.wide || L.sided) return xgm_launch_error("andw all kernel", 0, "plain conjunctions with one-byte wdf only");
    const size_t smem = xgm_andw_smem_bytes(L.seg.stripe_bits, L.tab_terms, L.cap, false, L.stripes_per_group, false, false);
    if (smem > 160u * 1024u) return xgm_launch_error("andw all kernel LDS budget", 0, "LDS request exceeds 160 KiB");
    const dim3 grid((L.n_work + XGM_WAVES - 1) / XGM_WAVES), block(XGM_WG);
    auto kern = xgm_andw_all_kernel;
    static std::atomic<size_t> seen{0};
    if (int rc_ = ensure_dyn_smem(kern, smem, seen)) return rc_;
    XGM_LAUNCH_TIMED(L, kern, grid, block, smem, stream, L.seg, L.queries, L.work, L.n_work, L.stripes_per_group, L.tab_terms, L.cap, L.k_stride, L.cand, L.ghdr, out);
    XGM_HIP_CHECK(hipGetLastError());
    return 0;
}

int xgm_launch_andw_list(const xgm_match_launch& L, hipStream_t stream) {
    if (!L.phrase || L.wide || L.sided) return xgm_launch_error("andw list kernel", 0, "positional batches with one-byte wdf only");
    const size_t smem = xgm_andw_smem_bytes(L.seg.stripe_bits, L.tab_terms, L.cap, false, L.stripes_per_group, true, false);
    if (smem > 160u * 1024u) return xgm_launch_error("andw list kernel LDS budget", 0, "LDS request exceeds 160 KiB");
    const dim3 grid((L.n_work + XGM_WAVES - 1) / XGM_WAVES), block(XGM_WG);
    if (L.tally) {
        auto kern = xgm_andw_list_kernel<true>;
        static std::atomic<size_t> seen{0};
        if (int rc_ = ensure_dyn_smem(kern, smem, seen)) return rc_;
        XGM_LAUNCH_TIMED(L, kern, grid, block, smem, stream, L.seg, L.queries, L.work, L.n_work, L.stripes_per_group, L.tab_terms, L.cap, L.k_stride, L.cand, L.ghdr, L.hist, L.fuse);
    } else {
        auto kern = xgm_andw_list_kernel<false>;
        static std::atomic<size_t> seen{0};
        if (int rc_ = ensure_dyn_smem(kern, smem, seen)) return rc_;
        XGM_LAUNCH_TIMED(L, kern, grid, block, smem, stream, L.seg, L.queries, L.work, L.n_work, L.stripes_per_group, L.tab_terms, L.cap, L.k_stride, L.cand, L.ghdr, L.hist, L.fuse);
    }
    XGM_HIP_CHECK(hipGetLastError());
    return 0;
}

int xgm_merge_cycles_fetch(unsigned long long* out8) {
    hipDeviceSynchronize();
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_merge_cycles), 64) != hipSuccess) return -1;
    hipMemcpyToSymbol(HIP_SYMBOL(g_merge_cycles), z, 64);
    if (XGM_MERGE_TIMERS) {
        unsigned long long mx[8];
        if (hipMemcpyFromSymbol(mx, HIP_SYMBOL(g_merge_max), 64) == hipSuccess)
            for (int i = 0; i < 5; ++i) fprintf(stderr, "merge section %d: slowest workgroup %llu cycles (a query of %llu units)\n", i, mx[i] >> 20, (mx[i] >> 4) & 0xFFFFull);
        fprintf(stderr, "merge: start of workgroup 0 to the last end (longest launch) %llu ticks of 10 ns; longest workgroup %llu ticks\n", mx[6], mx[7]);
        hipMemcpyToSymbol(HIP_SYMBOL(g_merge_max), z, 64);
    }
    return 0;
}

int xgm_launch_merge(const xgm_cand* cand, const xgm_group_hdr* ghdr, const uint32_t* goff, uint32_t k_stride_in,
                     const uint32_t* kq, uint32_t nq, uint32_t cap, uint32_t k_stride_out, xgm_hit* hits,
                     xgm_result_hdr* hdrs, const double* max_possible, const uint32_t* row_of, hipStream_t stream) {
    const size_t smem = (size_t)cap * 16 + 64 + (size_t)kMergeSel * 16;
    { static std::atomic<size_t> seen{0}; if (int rc_ = ensure_dyn_smem(xgm_merge_kernel, smem, seen)) return rc_; }
    hipLaunchKernelGGL(xgm_merge_kernel, dim3(nq), dim3(XGM_WG), smem, stream, cand, ghdr, goff, k_stride_in, kq, cap,
                       k_stride_out, hits, hdrs, max_possible, row_of);
    XGM_HIP_CHECK(hipGetLastError());
    return 0;
}

int xgm_launch_merge_parts(const xgm_hit* all_hits, const xgm_result_hdr* all_hdrs, uint32_t n_parts, uint32_t nq, uint32_t k_stride,
                           const uint32_t* kq, uint32_t cap, xgm_hit* hits, xgm_result_hdr* hdrs, const uint32_t* row_of, hipStream_t stream) {
    const size_t smem = (size_t)cap * 16 + 64;
    { static std::atomic<size_t> seen{0}; if (int rc_ = ensure_dyn_smem(xgm_merge_shards_kernel, smem, seen)) return rc_; }
    hipLaunchKernelGGL(xgm_merge_shards_kernel, dim3(nq), dim3(XGM_WG), smem, stream, all_hits, all_hdrs, n_parts, nq,
                       k_stride, kq, cap, hits, hdrs, 0u, row_of, (unsigned long long)nq * k_stride * sizeof(xgm_hit), (unsigned long long)nq * sizeof(xgm_result_hdr));
    XGM_HIP_CHECK(hipGetLastError());
    return 0;
}

int xgm_launch_merge_shards(const xgm_hit* all_hits, const xgm_result_hdr* all_hdrs, uint32_t n_shards, uint32_t nq,
                            uint32_t k_stride, const uint32_t* kq, uint32_t cap, xgm_hit* hits, xgm_result_hdr* hdrs,
                            hipStream_t stream, size_t shard_record_bytes) {
    const size_t smem = (size_t)cap * 16 + 64;
    { static std::atomic<size_t> seen{0}; if (int rc_ = ensure_dyn_smem(xgm_merge_shards_kernel, smem, seen)) return rc_; }
    hipLaunchKernelGGL(xgm_merge_shards_kernel, dim3(nq), dim3(XGM_WG), smem, stream, all_hits, all_hdrs, n_shards, nq,
                       k_stride, kq, cap, hits, hdrs, 1u, (const uint32_t*)nullptr,
                       shard_record_bytes ? (unsigned long long)shard_record_bytes : (unsigned long long)nq * k_stride * sizeof(xgm_hit),
                       shard_record_bytes ? (unsigned long long)shard_record_bytes : (unsigned long long)nq * sizeof(xgm_result_hdr));
    XGM_HIP_CHECK(hipGetLastError());
    return 0;
}

int xgm_launch_decode(const xgm_seg_dev& seg, uint32_t term_id, uint32_t b0, uint32_t nblk, const uint64_t* ord_base,
                      uint32_t* out_did, uint32_t* out_wdf, hipStream_t stream) {
    if (nblk == 0) return 0;
    uint32_t grid = (nblk + XGM_WAVES - 1) / XGM_WAVES;
    if (grid > 4096u) grid = 4096u;
    hipLaunchKernelGGL(xgm_decode_kernel, dim3(grid), dim3(XGM_WG), 0, stream, seg, term_id, b0, nblk, ord_base, out_did, out_wdf);
    XGM_HIP_CHECK(hipGetLastError());
    return 0;
}
