/* K6, the positional predicates straight from HBM — one document per lane, serial dependent reads: ExactPhrasePostList / PhrasePostList /
 * NearPostList::test_doc restated (reference src/xapian/matcher/exactphrasepostlist.cc:75-133, phrasepostlist.cc:60-90,
 * nearpostlist.cc:60-160).  Shared by the wave kernels' slow path (xgm_kernels.hip) and the dense conjunction kernel
 * (xgm_dense_and.hip), whose positional survivors are rare. */
#ifndef XGM_POSFILTER_H
#define XGM_POSFILTER_H

#include <hip/hip_runtime.h>

#include "xgm_device.h"
#include "xgm_wave.h"

namespace {

/* One document's positions of one term, in HBM: 2 or 4 bytes per entry (XGM_TF_POS16). */
struct PosList {
    const unsigned char* p; uint32_t n; uint32_t w16;
    __device__ __forceinline__ uint32_t at(uint32_t i) const {
        return w16 ? (uint32_t)reinterpret_cast<const uint16_t*>(p)[i] : reinterpret_cast<const uint32_t*>(p)[i];
    }
};

/* The three positional predicates straight from HBM, one document per lane with serial dependent reads: the SLOW
 * path — documents with more than kPosFast positions of a term, 4-byte position lists, and the workgroup kernel.
 * The wave kernel's fast path (positions staged in LDS by vector loads) is posfilter_lds below. */

/* ExactPhrasePostList::test_doc: is there a base with term i at base + phrase_index[i] for all i? */
__device__ bool phrase_exact(const PosList* pl, const uint8_t* pidx, uint32_t n_terms) {
    /* drive from the shortest list */
    uint32_t drv = 0;
    for (uint32_t t = 1; t < n_terms; ++t) if (pl[t].n < pl[drv].n) drv = t;
    uint32_t cursor[XGM_PHRASE_MAX_TERMS];
    for (uint32_t t = 0; t < XGM_PHRASE_MAX_TERMS; ++t) cursor[t] = 0;
    for (uint32_t i = 0; i < pl[drv].n; ++i) {
        uint32_t x = pl[drv].at(i);
        if (x < pidx[drv]) continue;
        uint32_t base = x - pidx[drv];
        bool ok = true;
        for (uint32_t t = 0; t < n_terms && ok; ++t) {
            if (t == drv) continue;
            uint32_t want = base + pidx[t];
            uint32_t c = cursor[t];
            while (c < pl[t].n && pl[t].at(c) < want) ++c;
            cursor[t] = c;
            ok = (c < pl[t].n) && (pl[t].at(c) == want);
        }
        if (ok) return true;
    }
    return false;
}

/* PhrasePostList::test_doc (windowed, ordered), restated with the same forward-only cursors. */
__device__ bool phrase_window(const PosList* pl_plan, const uint8_t* pidx, uint32_t n_terms, uint32_t window) {
    /* reorder to phrase order: terms[i] of the reference is the i-th word of the phrase */
    PosList pl[XGM_PHRASE_MAX_TERMS];
    for (uint32_t t = 0; t < n_terms; ++t) pl[pidx[t]] = pl_plan[t];
    uint32_t cur[XGM_PHRASE_MAX_TERMS];
    bool started[XGM_PHRASE_MAX_TERMS];
    for (uint32_t t = 0; t < XGM_PHRASE_MAX_TERMS; ++t) { cur[t] = 0; started[t] = false; }
    if (pl[0].n == 0) return false;               /* poslists[0]->next() */
    uint32_t b;
    while (true) {
        uint32_t base = pl[0].at(cur[0]);
        uint32_t pos = base;
        uint32_t i = 0;
        while (true) {
            if (++i == n_terms) return true;
            /* skip_to(pos + 1) on a forward-only list: never moves backwards */
            uint32_t c = cur[i];
            if (!started[i]) { started[i] = true; c = 0; }
            while (c < pl[i].n && pl[i].at(c) < pos + 1u) ++c;
            cur[i] = c;
            if (c >= pl[i].n) return false;
            pos = pl[i].at(c);
            b = pos + (n_terms - i);
            if (!(b - base <= window)) break;
        }
        uint32_t want = b - window;
        uint32_t c0 = cur[0];
        while (c0 < pl[0].n && pl[0].at(c0) < want) ++c0;
        cur[0] = c0;
        if (c0 >= pl[0].n) return false;
    }
}

/* NearPostList::test_doc (reference src/xapian/matcher/nearpostlist.cc:60-160) for DISTINCT terms: one position of
 * every term inside a span shorter than `window`, in any order — advance the list with the smallest head past
 * (largest head - window) until the heads fit or a list runs out. */
__device__ bool near_window(const PosList* pl, uint32_t n_terms, uint32_t window) {
    uint32_t cur[XGM_PHRASE_MAX_TERMS];
    for (uint32_t t = 0; t < XGM_PHRASE_MAX_TERMS; ++t) cur[t] = 0;
    for (uint32_t t = 0; t < n_terms; ++t) if (pl[t].n == 0) return false;
    while (true) {
        uint32_t lo = 0, lo_v = pl[0].at(cur[0]), hi_v = lo_v;
        for (uint32_t t = 1; t < n_terms; ++t) {
            const uint32_t v = pl[t].at(cur[t]);
            if (v < lo_v) { lo_v = v; lo = t; }
            if (v > hi_v) hi_v = v;
        }
        if (hi_v - lo_v < window) return true;
        const uint32_t want = hi_v - window + 1u;
        uint32_t c = cur[lo];
        while (c < pl[lo].n && pl[lo].at(c) < want) ++c;
        if (c >= pl[lo].n) return false;
        cur[lo] = c;
    }
}

/* NearPostList::test_doc IN FULL (nearpostlist.cc:70-150), for shards where distinct terms may share a position: the reference then wants
 * one occurrence of every term inside a span shorter than `window` AT PAIRWISE DISTINCT POSITIONS, and finds it greedily — the lists' heads in
 * a binary heap (smallest position on top; the heap of common/heap.h: libc++'s sift-up / sift-down, whose tie behaviour decides WHICH of two
 * equal heads moves), lists started lazily in ascending wdf order (TermCmp, nearpostlist.cc:52-58; equal wdf: query order — the sort of <= 8
 * elements is an insertion sort), the smallest head skipped forward while the span is too wide, and once every list is inside the window the
 * heads are walked in ascending position: one that sits on the previous head's position is advanced — out of the window: back to the outer
 * loop with the new maximum; otherwise it sinks to its place and the walk goes on.  Restated step by step so that the answer is the
 * reference's in every corner, not merely the predicate it approximates.  qpos[t] = plan term t's place in the query. */
__device__ bool near_colocated(const PosList* pl, const uint8_t* qpos, uint32_t n, uint32_t window) {
    uint32_t cur[XGM_PHRASE_MAX_TERMS];            /* cursor of every list; 0xFFFFFFFF = not started */
    uint32_t ord[XGM_PHRASE_MAX_TERMS], h[XGM_PHRASE_MAX_TERMS];
    for (uint32_t t = 0; t < XGM_PHRASE_MAX_TERMS; ++t) { cur[t] = 0xFFFFFFFFu; ord[t] = t; h[t] = 0; }
    /* terms by ascending wdf (= positions of the document's posting), ties in query order */
    for (uint32_t i = 1; i < n; ++i) {
        const uint32_t v = ord[i];
        uint32_t j = i;
        while (j > 0 && (pl[v].n < pl[ord[j - 1]].n || (pl[v].n == pl[ord[j - 1]].n && qpos[v] < qpos[ord[j - 1]]))) { ord[j] = ord[j - 1]; --j; }
        ord[j] = v;
    }
    auto pos = [&](uint32_t id) { return pl[id].at(cur[id]); };
    auto next = [&](uint32_t id) { cur[id] = cur[id] == 0xFFFFFFFFu ? 0u : cur[id] + 1u; return cur[id] < pl[id].n; };
    auto skip_to = [&](uint32_t id, uint32_t want) {           /* first position >= want, never backwards */
        uint32_t c = cur[id] == 0xFFFFFFFFu ? 0u : cur[id];
        while (c < pl[id].n && pl[id].at(c) < want) ++c;
        cur[id] = c;
        return c < pl[id].n;
    };
    auto above = [&](uint32_t a, uint32_t b) { return pos(a) > pos(b); };      /* Cmp: a min-heap on the heads' positions */
    auto sift_down = [&](uint32_t len, uint32_t start) {
        uint32_t child = start;
        if (len < 2u || (len - 2u) / 2u < child) return;
        child = 2u * child + 1u;
        if (child + 1u < len && above(h[child], h[child + 1u])) ++child;
        if (above(h[child], h[start])) return;
        const uint32_t top = h[start];
        const uint32_t top_pos = pos(top);
        do {
            h[start] = h[child];
            start = child;
            if ((len - 2u) / 2u < child) break;
            child = 2u * child + 1u;
            if (child + 1u < len && above(h[child], h[child + 1u])) ++child;
        } while (!(pos(h[child]) > top_pos));
        h[start] = top;
    };
    auto push = [&](uint32_t len) {                              /* h[len - 1] is new */
        if (len < 2u) return;
        uint32_t p = (len - 2u) / 2u, last = len - 1u;
        if (!above(h[p], h[last])) return;
        const uint32_t t = h[last];
        const uint32_t t_pos = pos(t);
        do {
            h[last] = h[p];
            last = p;
            if (p == 0u) break;
            p = (p - 1u) / 2u;
        } while (pos(h[p]) > t_pos);
        h[last] = t;
    };
    auto pop = [&](uint32_t len) {                               /* the top goes to h[len - 1] */
        if (len < 2u) return;
        const uint32_t t = h[0]; h[0] = h[len - 1u]; h[len - 1u] = t;
        sift_down(len - 1u, 0u);
    };
    h[0] = ord[0];
    if (!next(h[0])) return false;
    uint32_t last = pos(h[0]), end = 1u;
    while (true) {
        if (last - pos(h[0]) < window) {
            if (end != n) {
                const uint32_t id = ord[end];
                if (last < window) { if (!next(id)) return false; }
                else if (!skip_to(id, last - window + 1u)) return false;
                const uint32_t p = pos(id);
                if (p > last) last = p;
                h[end++] = id;
                push(end);
                continue;
            }
            /* every list inside the window: advance the ones that share a position with the head before them */
            uint32_t p = pos(h[0]);
            pop(end);
            uint32_t i = end - 1u;
            bool refit = false;
            while (true) {
                if (pos(h[0]) == p) {
                    if (!next(h[0])) return false;
                    const uint32_t np = pos(h[0]);
                    if (np - pos(h[end - 1u]) >= window) { last = np; refit = true; break; }
                    sift_down(i, 0u);                              /* Heap::replace */
                    continue;
                }
                p = pos(h[0]);
                pop(i);
                if (--i == 0u) return true;
            }
            (void)refit;
            if (end > 1u) for (int s = (int)((end - 2u) / 2u); s >= 0; --s) sift_down(end, (uint32_t)s);      /* Heap::make */
            continue;
        }
        if (!skip_to(h[0], last - window + 1u)) break;
        { const uint32_t p = pos(h[0]); if (p > last) last = p; }
        sift_down(end, 0u);                                        /* Heap::replace */
    }
    return false;
}

__device__ __forceinline__ bool posfilter_slow(const PosList* pl, const xgm_dev_query& q, uint32_t T) {
    if (q.flags & XGM_QF_NEAR_COLOC) return near_colocated(pl, q.phrase_index, T, q.window);
    if (q.flags & XGM_QF_NEAR) return near_window(pl, T, q.window);
    return (q.flags & XGM_QF_EXACT) ? phrase_exact(pl, q.phrase_index, T) : phrase_window(pl, q.phrase_index, T, q.window);
}

/* The serial predicates on position lists the WAVE has copied into its LDS first (list t at stage + t * stride, n_t entries of 2 or 4 bytes):
 * the slow path of xgm_dense_unit / xgm_flat_unit for a document with more than kPosFast positions of a term.  Straight from HBM (above) every
 * position costs a dependent read of ~1-2 us — a two-term phrase over lists of 20 and 8 positions took ~50 us, one lane busy, and the frequent
 * terms of C5 put such a document into every other batch of 64 survivors (tools/qcost.py phase clocks, round 4); the copy is one 16-byte load
 * per lane and list, the reads behind it cost an LDS access each.  Out of line: its registers and private cursor arrays are its own. */
__device__ __attribute__((noinline)) bool posfilter_staged(const unsigned char* stage, uint32_t stride, const xgm_dev_query& q, uint32_t T, uint32_t n0, uint32_t n1,
                                                           uint32_t n2, uint32_t n3, uint32_t w16_mask) {
    const uint32_t ns[4] = {n0, n1, n2, n3};
    PosList pl[XGM_PHRASE_MAX_TERMS];
    for (uint32_t t = 0; t < T && t < 4u; ++t) {
        pl[t].p = stage + (size_t)t * stride;
        pl[t].n = ns[t];
        pl[t].w16 = (w16_mask >> t) & 1u;
    }
    return posfilter_slow(pl, q, T);
}

/* ---- K6 fast path: positions staged in the wave's LDS ------------------------------------------------------------
 * lp[(t * kPosFast + j) * 64 + lane] = j-th position (u16) of plan term t in the lane's document; cnt(t) = how many.
 * The same three predicates as above, on LDS data with per-lane cursors packed 5 bits per term into one 64-bit
 * register; every loop over terms is unrolled (T is wave-uniform), so nothing is indexed dynamically in registers.
 * Bank conflicts: lanes l and l' collide only if (j * 32 + l / 2) = (j' * 32 + l' / 2) mod 64 with (j, l/2) != (j', l'/2),
 * which needs l/2 and l'/2 to differ by 32: never. */
constexpr uint32_t kPosFast = 16;          /* positions per (document, term) the fast path holds: 32 bytes = 2 vector loads */

struct LdsPos {
    const uint16_t* lp; uint32_t lane;
    __device__ __forceinline__ uint32_t at(uint32_t t, uint32_t j) const { return lp[(t * kPosFast + j) * 64u + lane]; }
};
/* The WIDE layout of the same area: 16 documents at a time with up to kPosWide = 64 positions of a term each — document s (0..15) owns
 * lanes' columns 4 s .. 4 s + 3 of every row, its j-th position of term t sits in row j & 15, column 4 s + (j >> 4).  For the documents the
 * first pass had to leave out (17..64 positions of a term: among the candidates that pass the WEIGHT test of a frequent-term phrase — high wdf
 * is what makes them heavy — every third one), before the serial path takes what is left (posfilter_staged). */
constexpr uint32_t kPosWide = 64;
struct LdsPosWide {
    const uint16_t* lp; uint32_t col;              /* col = 4 x the document's slot */
    __device__ __forceinline__ uint32_t at(uint32_t t, uint32_t j) const { return lp[(t * kPosFast + (j & 15u)) * 64u + col + (j >> 4)]; }
};
/* per-lane cursors, BITS per term in one 64-bit register (5: counts <= 16 ... 31; 7: counts <= 64; XGM_PHRASE_MAX_TERMS x 7 <= 64) */
template <uint32_t BITS>
__device__ __forceinline__ uint32_t cur_get(uint64_t c, uint32_t t) { return (uint32_t)(c >> (BITS * t)) & ((1u << BITS) - 1u); }
template <uint32_t BITS>
__device__ __forceinline__ uint64_t cur_set(uint64_t c, uint32_t t, uint32_t v) { return (c & ~((uint64_t)((1u << BITS) - 1u) << (BITS * t))) | ((uint64_t)v << (BITS * t)); }
static_assert(XGM_PHRASE_MAX_TERMS * 7u <= 64u, "the wide pass packs 7-bit cursors of every term into 64 bits");

template <uint32_t BITS = 5u, typename LP, typename CntF>
__device__ bool lds_phrase_exact(const LP& L, CntF cnt, const uint8_t* pidx, uint32_t T) {
    /* driven from plan term 0 (the rarest term of the collection); the predicate does not depend on the driver */
    const uint32_t n0 = cnt(0u), p0 = pidx[0];
    uint64_t cur = 0;
    for (uint32_t i = 0; i < n0; ++i) {
        const uint32_t x = L.at(0u, i);
        if (x < p0) continue;
        const uint32_t base = x - p0;
        bool ok = true;
#pragma unroll
        for (uint32_t t = 1; t < XGM_PHRASE_MAX_TERMS; ++t) {
            if (t < T && ok) {
                const uint32_t want = base + pidx[t], nt = cnt(t);
                uint32_t c = cur_get<BITS>(cur, t);
                while (c < nt && L.at(t, c) < want) ++c;
                cur = cur_set<BITS>(cur, t, c);
                ok = c < nt && L.at(t, c) == want;
            }
        }
        if (ok) return true;
    }
    return false;
}

template <uint32_t BITS = 5u, typename LP, typename CntF>
__device__ bool lds_phrase_window(const LP& L, CntF cnt, const uint8_t* pidx, uint32_t T, uint32_t window) {
    /* inv[i] = plan term that is the i-th word of the phrase (wave-uniform) */
    uint32_t inv[XGM_PHRASE_MAX_TERMS];
#pragma unroll
    for (uint32_t i = 0; i < XGM_PHRASE_MAX_TERMS; ++i) inv[i] = 0;
#pragma unroll
    for (uint32_t t = 0; t < XGM_PHRASE_MAX_TERMS; ++t)
        if (t < T) {
#pragma unroll
            for (uint32_t i = 0; i < XGM_PHRASE_MAX_TERMS; ++i) if (pidx[t] == i) inv[i] = t;
        }
    const uint32_t n0 = cnt(inv[0]);
    if (n0 == 0) return false;
    uint64_t cur = 0;                              /* cursor of phrase word i at bits 5i; all lists start at their first entry */
    while (true) {
        const uint32_t base = L.at(inv[0], cur_get<BITS>(cur, 0u));
        uint32_t pos = base, b = 0;
        bool fits = true, out = false;
#pragma unroll
        for (uint32_t i = 1; i < XGM_PHRASE_MAX_TERMS; ++i) {
            if (i < T && fits && !out) {
                const uint32_t ni = cnt(inv[i]);
                uint32_t c = cur_get<BITS>(cur, i);
                while (c < ni && L.at(inv[i], c) < pos + 1u) ++c;
                cur = cur_set<BITS>(cur, i, c);
                if (c >= ni) { out = true; }
                else {
                    pos = L.at(inv[i], c);
                    b = pos + (T - i);
                    fits = b - base <= window;
                }
            }
        }
        if (out) return false;
        if (fits) return true;
        const uint32_t want = b - window;
        uint32_t c0 = cur_get<BITS>(cur, 0u);
        while (c0 < n0 && L.at(inv[0], c0) < want) ++c0;
        if (c0 >= n0) return false;
        cur = cur_set<BITS>(cur, 0u, c0);
    }
}

template <uint32_t BITS = 5u, typename LP, typename CntF>
__device__ bool lds_near_window(const LP& L, CntF cnt, uint32_t T, uint32_t window) {
    uint64_t cur = 0;
    bool empty = false;
#pragma unroll
    for (uint32_t t = 0; t < XGM_PHRASE_MAX_TERMS; ++t) if (t < T && cnt(t) == 0u) empty = true;
    if (empty) return false;
    while (true) {
        uint32_t lo = 0, lo_v = L.at(0u, cur_get<BITS>(cur, 0u)), hi_v = lo_v;
#pragma unroll
        for (uint32_t t = 1; t < XGM_PHRASE_MAX_TERMS; ++t) {
            if (t < T) {
                const uint32_t v = L.at(t, cur_get<BITS>(cur, t));
                if (v < lo_v) { lo_v = v; lo = t; }
                if (v > hi_v) hi_v = v;
            }
        }
        if (hi_v - lo_v < window) return true;
        const uint32_t want = hi_v - window + 1u, nl = cnt(lo);
        uint32_t c = cur_get<BITS>(cur, lo);
        while (c < nl && L.at(lo, c) < want) ++c;
        if (c >= nl) return false;
        cur = cur_set<BITS>(cur, lo, c);
    }
}

struct __attribute__((packed, aligned(2))) Pos8 { uint32_t a, b, c, d; };        /* 8 u16 positions, 2-byte aligned in HBM */

}  // namespace

#endif
