/* ProtoMSet's collation walked by ONE WAVE over a list of matches in docid order (matcher/protomset.h:340-400): the kept documents in LDS — in RANK
 * order once the heap is made, so that a replacement is a ballot-counted insertion and the worst is the last entry —, 64 entries judged per step by a
 * ballot pair against (min_weight, the worst kept), four chunks of 64 in flight.  Shared by xgm_replay.hip (one query's list cut into segments) and
 * xgm_count.hip (a batch's queries, the units of the match kernel as segments). */
#ifndef XGM_REPLAY_WAVE_H
#define XGM_REPLAY_WAVE_H

#include <hip/hip_runtime.h>

#include "xgm_device.h"

namespace {

/* a ranks before b under msetcmp_by_relevance<true> (msetcmp.cc:55-62): heavier first, equal weights by ascending docid */
__device__ __forceinline__ bool rp_before(double aw, uint32_t ad, double bw, uint32_t bd) { return aw > bw || (aw == bw && ad < bd); }

constexpr uint32_t kSegMaxK = 1024u;

struct WaveState {
    double* w; uint32_t* d; uint32_t* m;      /* LDS: the kept documents — in RANK order (best first) once the heap is made */
    uint32_t size, cap; bool heap_built;        /* cap: the page size K; slot [cap] is a spare one */
    double min_w, worst_w; uint32_t worst_d;
};

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

/* the kept documents put in rank order in place (ranks by counting: each lane ranks its own entries, all reads before any write) */
__device__ __forceinline__ void wave_sort_kept(WaveState& st, uint32_t lane) {
    wave_lds_sync();
    double rw[16]; uint32_t rd[16], rm[16], rr[16];
#pragma unroll
    for (uint32_t t = 0; t < 16u; ++t) {
        const uint32_t i = lane + t * 64u;
        rr[t] = 0xFFFFFFFFu; rw[t] = 0.0; rd[t] = 0; rm[t] = 0;
        if (i < st.size) {
            rw[t] = st.w[i]; rd[t] = st.d[i]; rm[t] = st.m[i];
            uint32_t r = 0;
            for (uint32_t j = 0; j < st.size; ++j) r += rp_before(st.w[j], st.d[j], rw[t], rd[t]) ? 1u : 0u;
            rr[t] = r;
        }
    }
    wave_lds_sync();
#pragma unroll
    for (uint32_t t = 0; t < 16u; ++t) if (rr[t] != 0xFFFFFFFFu) { st.w[rr[t]] = rw[t]; st.d[rr[t]] = rd[t]; st.m[rr[t]] = rm[t]; }
    wave_lds_sync();
}

/* (ew, ed, em) ranks before the worst kept: it takes its place in the rank order, the worst drops out.  The place = the kept documents that rank
 * before it (a ballot count per 64); the entries from there on move down one slot, highest indices first. */
__device__ __forceinline__ void wave_insert_sorted(WaveState& st, double ew, uint32_t ed, uint32_t em, uint32_t lane) {
    uint32_t r = 0;
    for (uint32_t i0 = 0; i0 < st.size; i0 += 64u) {
        const uint32_t i = i0 + lane;
        r += (uint32_t)__popcll(__ballot(i < st.size && rp_before(st.w[i], st.d[i], ew, ed)));
    }
    const uint32_t last = st.size - 1u;                      /* the worst: overwritten by the shift */
    /* (no branch between the wave barriers: a lane with nothing to move moves the spare slot st.cap onto itself) */
    for (uint32_t hi = last; hi > r;) {                      /* move [r, last) to [r + 1, last], in chunks from the top */
        const uint32_t lo = hi - r > 64u ? hi - 64u : r;     /* this chunk: source indices [lo, hi) */
        const uint32_t from = lo + lane < hi ? lo + lane : st.cap, to = lo + lane < hi ? lo + lane + 1u : st.cap;
        const double w = st.w[from]; const uint32_t d = st.d[from], m = st.m[from];
        wave_lds_sync();
        st.w[to] = w; st.d[to] = d; st.m[to] = m;
        wave_lds_sync();
        hi = lo;
    }
    const uint32_t at = lane == 0u ? r : st.cap;
    st.w[at] = ew; st.d[at] = ed; st.m[at] = em;
    wave_lds_sync();
    st.worst_w = st.w[last]; st.worst_d = st.d[last];
}

/* entries [begin, end) of the list through ProtoMSet::add, one wave; returns how many reached it.  known_base: what the documents before
 * `begin` contributed to the check_at_least test (a segment that starts with the heap made passes check_at_least itself) */
__device__ __forceinline__ unsigned long long wave_replay_segment(const xgm_hit* __restrict__ list, unsigned long long begin, unsigned long long end,
                                                                  uint32_t K, unsigned long long check_at_least, unsigned long long known_base, WaveState& st,
                                                                  uint32_t lane) {
    unsigned long long known = 0;
    if (begin >= end) return 0;
    /* four chunks in flight: a wave alone on its SIMD has nothing else to hide a load's latency behind */
    xgm_hit ring[4];
#pragma unroll
    for (uint32_t r = 0; r < 4u; ++r) {
        ring[r].docid = 0; ring[r].subqs_matched = 0; ring[r].weight = 0.0;
        if (begin + r * 64u + lane < end) ring[r] = list[begin + r * 64u + lane];
    }
    /* (the ring is indexed by unrolled constants: rotating it through register copies would wait for the newest load every chunk) */
    for (unsigned long long c4 = begin; c4 < end; c4 += 256u)
#pragma unroll
    for (uint32_t rj = 0; rj < 4u; ++rj) {
        const unsigned long long c0 = c4 + rj * 64u;
        if (c0 >= end) break;
        const xgm_hit h = ring[rj];
        if (c0 + 256u + lane < end) ring[rj] = list[c0 + 256u + lane];
        const bool valid = c0 + lane < end;
        uint32_t start = 0;
        while (start < 64u) {
            const bool filling = st.size < K, steady = !filling && st.heap_built && K != 0u;
            const bool ps = valid && lane >= start && !(h.weight < st.min_w);
            const bool bt = ps && steady && rp_before(h.weight, h.docid, st.worst_w, st.worst_d);
            const unsigned long long mp = __ballot(ps), mb = __ballot(bt);
            uint32_t take = (uint32_t)__popcll(mp), ev = 64u;
            bool event = false;
            if (filling || !steady) {
                const uint32_t room = filling ? K - st.size : (K == 0u ? take : 1u);
                if (take > room || (!filling && K != 0u && take >= 1u)) {
                    unsigned long long x = mp;
                    for (uint32_t j = 1; j < room; ++j) x &= x - 1ull;
                    ev = (uint32_t)__ffsll((long long)x) - 1u;
                    take = room;
                    event = !filling;
                }
            } else if (mb) {
                ev = (uint32_t)__ffsll((long long)mb) - 1u;
                event = true;
                take = (uint32_t)__popcll(mp & (ev >= 63u ? ~0ull : ((2ull << ev) - 1ull)));
            }
            known += take;
            if (filling) {
                if (ps && lane <= ev) {
                    const uint32_t r = st.size + (uint32_t)__popcll(mp & ((1ull << lane) - 1ull));
                    st.w[r] = h.weight; st.d[r] = h.docid; st.m[r] = h.subqs_matched;
                }
                st.size += take;
            }
            start = ev >= 64u ? 64u : ev + 1u;
            if (!event) continue;
            const double ew = __shfl(h.weight, (int)ev);
            const uint32_t ed = (uint32_t)__shfl((int)h.docid, (int)ev), em = (uint32_t)__shfl((int)h.subqs_matched, (int)ev);
            if (!st.heap_built) {
                st.heap_built = true;
                wave_sort_kept(st, lane);                                     /* (Heap::make's counterpart: from here on the kept set stays in rank order) */
                st.worst_w = st.w[st.size - 1u]; st.worst_d = st.d[st.size - 1u];
                if (known + known_base >= check_at_least) st.min_w = st.worst_w;
            }
            if (rp_before(ew, ed, st.worst_w, st.worst_d)) {
                wave_insert_sorted(st, ew, ed, em, lane);
                if (known + known_base >= check_at_least) st.min_w = st.worst_w;
            }
        }
    }
    return known;
}

/* the kept documents in rank order → out[0 .. size) */
__device__ __forceinline__ void wave_write_sorted(WaveState& st, xgm_hit* __restrict__ out, uint32_t lane) {
    if (!st.heap_built) wave_sort_kept(st, lane);                            /* (a list that never filled the page) */
    wave_lds_sync();
    for (uint32_t i = lane; i < st.size; i += 64u) {
        xgm_hit h; h.docid = st.d[i]; h.subqs_matched = st.m[i]; h.weight = st.w[i];
        out[i] = h;
    }
}

__device__ __forceinline__ WaveState wave_state_carve(unsigned char* smem, uint32_t K) {
    WaveState st;
    st.w = reinterpret_cast<double*>(smem);
    st.d = reinterpret_cast<uint32_t*>(st.w + K + 1u);
    st.m = st.d + K + 1u;
    st.cap = K; st.size = 0; st.heap_built = false; st.min_w = 0.0; st.worst_w = 0.0; st.worst_d = 0;
    return st;
}

/* entries of the sorted arr that rank before (w, d) */
__device__ __forceinline__ uint32_t count_before(const xgm_hit* arr, uint32_t n, double w, uint32_t d) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (rp_before(arr[mid].weight, arr[mid].docid, w, d)) lo = mid + 1u; else hi = mid;
    }
    return lo;
}


}  // namespace

#endif
