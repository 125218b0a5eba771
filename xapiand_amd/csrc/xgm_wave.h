/* Wave-level device primitives shared by the match kernels (gfx950, wave64): DPP scan, ballot
 * ranks, wave-private LDS fences, 64-ary search, and the K1 block decode. */
#ifndef XGM_WAVE_H
#define XGM_WAVE_H

#include <hip/hip_runtime.h>

#include "xgm_device.h"

namespace {


constexpr uint32_t kInfStripe = 0xFFFFFFFFu;
constexpr uint32_t kStageWords = 272;   /* 256 payload words + overrun, per wave */

/* ---------------------------------------------------------------- wave primitives ------------ */

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}

/* Inclusive prefix sum over the 64 lanes: 4 row_shr steps inside each 16-lane row, then
 * row_bcast:15 / row_bcast:31 to carry across rows (CDNA DPP). */
/* __builtin_amdgcn_readlane / readfirstlane return a SIGNED int: OR-ing a low half into a 64-bit value sign-extends it and
 * wipes the high half whenever bit 31 is set.  Always go through these. */
__device__ __forceinline__ uint32_t rl32(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
__device__ __forceinline__ uint32_t rfl32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t rl64(uint64_t v, uint32_t lane) { return ((uint64_t)rl32((uint32_t)(v >> 32), lane) << 32) | (uint64_t)rl32((uint32_t)v, lane); }
__device__ __forceinline__ double rl_f64(double v, uint32_t lane) { return __longlong_as_double((long long)rl64((uint64_t)__double_as_longlong(v), lane)); }

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += dpp0<0x111, 0xf>(v);
    v += dpp0<0x112, 0xf>(v);
    v += dpp0<0x114, 0xf>(v);
    v += dpp0<0x118, 0xf>(v);
    v += dpp0<0x142, 0xa>(v);
    v += dpp0<0x143, 0xc>(v);
    return v;
}

__device__ __forceinline__ uint32_t lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

__device__ __forceinline__ uint32_t mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

/* s_waitcnt vmcnt(0) (gfx9 encoding: vmcnt = simm16[15:14|3:0], expcnt [6:4] and lgkmcnt [11:8] left at their maxima): every
 * vector-memory operation this wave has issued — write-through stores included — has been acknowledged by its coherence point. */
__device__ __forceinline__ void xgm_wait_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }

/* Address spaces for the few helpers that stay out of line: a plain pointer parameter is GENERIC there, and the compiler reads LDS through flat
 * loads (slow path into the LDS, counted as memory AND LDS operations).  The host build of the emulation has one address space. */
#if defined(XGM_EMU)
#define XGM_AS_LDS
#define XGM_AS_GLOBAL
#else
#define XGM_AS_LDS __attribute__((address_space(3)))
#define XGM_AS_GLOBAL __attribute__((address_space(1)))
#endif

/* First index in [lo, hi) with arr[i] >= key (hi if none); 64-ary search, result wave-uniform. */
__device__ uint32_t wave_lower_bound(const uint32_t* __restrict__ arr_, uint32_t lo, uint32_t hi, uint32_t key,
                                     uint32_t lane) {
    const XGM_AS_GLOBAL uint32_t* arr = (const XGM_AS_GLOBAL uint32_t*)arr_;          /* (every caller searches an array of the segment in HBM) */
    while (hi - lo > 64u) {
        uint32_t step = (hi - lo + 63u) / 64u;
        uint32_t p = lo + lane * step;
        bool less = (p < hi) && (arr[p] < key);
        uint32_t c = (uint32_t)__popcll(__ballot(less));
        if (c == 0) return lo;
        uint32_t nlo = lo + (c - 1u) * step + 1u;
        uint32_t nhi = lo + c * step;
        hi = nhi < hi ? nhi : hi;
        lo = nlo;
    }
    uint32_t p = lo + lane;
    bool less = (p < hi) && (arr[p] < key);
    return lo + (uint32_t)__popcll(__ballot(less));
}

__device__ __forceinline__ uint32_t extract_bits(const uint32_t* s, uint32_t idx, uint32_t bw) {
    uint32_t bit = idx * bw;
    uint32_t w = bit >> 5;
    uint32_t v = __builtin_amdgcn_alignbit(s[w + 1], s[w], bit & 31u);
    return bw >= 32u ? v : (v & ((1u << bw) - 1u));
}

/* ---------------------------------------------------------------- K1: block decode ----------- */

struct DecodedPair {
    uint32_t d0, d1, w0, w1;      /* docids and wdfs of postings 2*lane and 2*lane+1 */
    uint32_t p0, p1;              /* position offsets (relative to the block) — PHRASE only */
    bool v0, v1;
};

/* Stage the block's payload into the wave's LDS window and unpack two postings per lane. */
template <bool WITH_POS>
__device__ __forceinline__ DecodedPair decode_block(const uint32_t* __restrict__ payload, uint32_t first_did,
                                                    uint32_t meta, uint32_t* stage, uint32_t lane) {
    const uint32_t n = XGM_META_COUNT(meta), bwg = XGM_META_BWG(meta), bww = XGM_META_BWW(meta);
    const uint32_t ngw = (n * bwg + 31u) >> 5, nww = (n * bww + 31u) >> 5;
    const uint32_t need = ngw + nww + 2u;                 /* +2: 2-word windows may overrun */
    for (uint32_t w = lane * 4u; w < need; w += 256u) {
        /* payload is 4-byte aligned; the segment is padded so this never leaves the allocation */
        uint32_t a = payload[w], b = payload[w + 1], c = payload[w + 2], d = payload[w + 3];
        stage[w] = a; stage[w + 1] = b; stage[w + 2] = c; stage[w + 3] = d;
    }
    wave_lds_fence();
    DecodedPair r;
    const uint32_t i0 = lane * 2u, i1 = i0 + 1u;
    r.v0 = i0 < n;
    r.v1 = i1 < n;
    uint32_t g0 = (r.v0 && i0 > 0u) ? extract_bits(stage, i0, bwg) + 1u : 0u;
    uint32_t g1 = r.v1 ? extract_bits(stage, i1, bwg) + 1u : 0u;
    r.w0 = r.v0 ? extract_bits(stage + ngw, i0, bww) : 0u;
    r.w1 = r.v1 ? extract_bits(stage + ngw, i1, bww) : 0u;
    uint32_t local = g0 + g1;
    uint32_t excl = wave_incl_scan(local) - local;
    r.d0 = first_did + excl + g0;
    r.d1 = r.d0 + g1;
    if (WITH_POS) {
        uint32_t lw = r.w0 + r.w1;
        uint32_t pex = wave_incl_scan(lw) - lw;
        r.p0 = pex;
        r.p1 = pex + r.w0;
    } else {
        r.p0 = r.p1 = 0;
    }
    wave_lds_fence();    /* the next block may overwrite the window */
    return r;
}

/* ---------------------------------------------------------------- K5: top-k in LDS ----------- */

/* Candidate order: weight descending, then docid ascending (msetcmp_by_relevance<true>).  Weights
 * are non-negative doubles so their bit patterns order like the values.  "a before b". */
__device__ __forceinline__ bool cand_before(uint64_t aw, uint32_t ad, uint64_t bw, uint32_t bd) {
    return aw > bw || (aw == bw && ad < bd);
}

struct __attribute__((packed, aligned(4))) Words4 { uint32_t a, b, c, d; };

/* Unpack a staged block (payload already in the wave's LDS window). */
template <bool WITH_POS>
__device__ __forceinline__ DecodedPair unpack_staged(const uint32_t* stage, uint32_t first_did, uint32_t meta, uint32_t lane) {
    const uint32_t n = XGM_META_COUNT(meta), bwg = XGM_META_BWG(meta), bww = XGM_META_BWW(meta);
    const uint32_t ngw = (n * bwg + 31u) >> 5;
    DecodedPair r;
    const uint32_t i0 = lane * 2u, i1 = i0 + 1u;
    r.v0 = i0 < n;
    r.v1 = i1 < n;
    uint32_t g0 = (r.v0 && i0 > 0u) ? extract_bits(stage, i0, bwg) + 1u : 0u;
    uint32_t g1 = r.v1 ? extract_bits(stage, i1, bwg) + 1u : 0u;
    r.w0 = r.v0 ? extract_bits(stage + ngw, i0, bww) : 0u;
    r.w1 = r.v1 ? extract_bits(stage + ngw, i1, bww) : 0u;
    uint32_t local = g0 + g1;
    uint32_t excl = wave_incl_scan(local) - local;
    r.d0 = first_did + excl + g0;
    r.d1 = r.d0 + g1;
    if (WITH_POS) {
        uint32_t lw = r.w0 + r.w1;
        uint32_t pex = wave_incl_scan(lw) - lw;
        r.p0 = pex;
        r.p1 = pex + r.w0;
    } else {
        r.p0 = r.p1 = 0;
    }
    return r;
}

__device__ __forceinline__ uint32_t payload_words(uint32_t meta) {
    const uint32_t n = XGM_META_COUNT(meta);
    return ((n * XGM_META_BWG(meta) + 31u) >> 5) + ((n * XGM_META_BWW(meta) + 31u) >> 5) + 2u;   /* +2: window overrun */
}

}  // namespace

#endif
