/* The end of a work unit of the wave kernels: its sorted candidate list and header go to HBM — and, when the launch finishes its
 * queries itself (xgm_fuse.arrive), the unit reports in on its query's arrival counter; the LAST unit of a query to arrive merges
 * all the units' lists through its own top-k buffer and writes the final hits and header: what xgm_merge_kernel does in a launch of
 * its own (ProtoMSet::finalise, protomset.h:657 — the same order: weight descending, docid ascending), without the launch and the
 * idle gaps around it (31 us + 5.5 us of a 398 us step on C2).
 *
 * Visibility between units without flushing the L2: a release fence at agent scope costs an L2 write-back per unit on this part
 * (measured: the conjunction kernel 0.35 -> 1.43 ms with one per unit), so the lists and headers are written THROUGH
 * (agent-scope atomic stores, 8 bytes each, `global_store ... sc1`), the wave waits for their acknowledgement with an EXPLICIT
 * `s_waitcnt vmcnt(0)` (a workgroup-scope release fence alone emits no vmcnt wait on gfx950 outside tgsplit mode — ADVICE r3: the
 * counter and the lists live in different L2 channels, the counter could become visible first), then lane 0 bumps the counter
 * (agent-scope atomic); the last unit reads the others' lists with agent-scope atomic loads (`sc1`), which the compiler may not hoist
 * above the counter read (acquire fence at workgroup scope = a compiler barrier, no cache invalidate).  tools/isa_contract.py checks
 * the ISA of every instantiation for exactly this sequence, so a compiler upgrade cannot silently drop it.  Measured cost of the
 * write-through stores + the arrival: none (0.351-0.368 vs 0.351-0.364 ms).
 */
#ifndef XGM_UNIT_FINISH_H
#define XGM_UNIT_FINISH_H

#include "xgm_device.h"
#include "xgm_wave.h"

static_assert(sizeof(xgm_cand) == 16 && sizeof(xgm_group_hdr) % 8 == 0, "written 8 bytes at a time");

__device__ __forceinline__ void xgm_store_cand(bool through, xgm_cand* dst, const xgm_cand& c) {
    if (!through) { *dst = c; return; }
    unsigned long long v[2];
    __builtin_memcpy(v, &c, 16);
    unsigned long long* d = reinterpret_cast<unsigned long long*>(dst);
    __hip_atomic_store(d, v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d + 1, v[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void xgm_store_hdr(bool through, xgm_group_hdr* dst, const xgm_group_hdr& h) {
    if (!through) { *dst = h; return; }
    unsigned long long v[sizeof(xgm_group_hdr) / 8];
    __builtin_memcpy(v, &h, sizeof h);
    unsigned long long* d = reinterpret_cast<unsigned long long*>(dst);
#pragma unroll
    for (uint32_t i = 0; i < sizeof(xgm_group_hdr) / 8; ++i) __hip_atomic_store(d + i, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/* Call with the whole wave after the unit's list and header have been stored (xgm_store_cand / xgm_store_hdr with through = true).
 * tk_w / tk_d (/ tk_m: per-candidate weighted-leaf counts, HAS_M) = the wave's top-k buffer of tk_cap entries (a power of two
 * >= k + 64), free again; sort() = the bitonic sort of that buffer, best first.  subqs_const: the weighted-leaf count of every
 * candidate when !HAS_M. */
template <bool HAS_M, class SortFn>
__device__ __forceinline__ void xgm_unit_arrive(const xgm_fuse& F, uint32_t qi, uint32_t k, uint32_t subqs_const, uint64_t* tk_w, uint32_t* tk_d,
                                                uint8_t* tk_m, uint32_t tk_cap, const xgm_cand* cand_all, const xgm_group_hdr* ghdr_all,
                                                uint32_t k_stride, uint32_t lane, SortFn sort) {
    qi = rfl32(qi);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");         /* compiler: nothing of the unit's list moves below this point */
    xgm_wait_vmem();                                               /* hardware: every write-through store of this wave has been acknowledged */
    __builtin_amdgcn_wave_barrier();
    uint32_t old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(&F.arrive[qi], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = rfl32(old);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");         /* compiler: no list load moves above the counter read */
    const uint32_t g0 = F.goff[qi], U = F.goff[qi + 1] - g0;
    if (old + 1u != U) return;
    /* ---- the query's last unit: every list is in memory ---- */
    if (lane == 0) __hip_atomic_store(&F.arrive[qi], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      /* ready for the next launch */
    for (uint32_t i = lane; i < tk_cap; i += 64u) { tk_w[i] = 0; tk_d[i] = 0xFFFFFFFFu; if (HAS_M) tk_m[i] = 0; }
    wave_lds_fence();
    uint32_t tkn = 0;
    bool theta_valid = false;
    uint64_t theta_w = 0;
    uint32_t theta_d = 0;
    unsigned long long matches = 0;                                /* per lane; summed at the end */
    uint32_t fill = 0;
    bool lower = false;
    for (uint32_t u0 = 0; u0 < U; u0 += 64u) {
        /* lane = unit.  The lists are sorted best first: a unit whose next candidate cannot enter the top k is done */
        const uint32_t u = u0 + lane;
        const bool valid = u < U;
        /* one round trip for the unit's header and its first kAhead candidates (nearly every unit is done after one or two) */
        constexpr uint32_t kAhead = 4u;
        const unsigned long long* hp = reinterpret_cast<const unsigned long long*>(&ghdr_all[g0 + (valid ? u : 0u)]);
        const unsigned long long* cp = reinterpret_cast<const unsigned long long*>(cand_all + (size_t)(g0 + (valid ? u : 0u)) * k_stride);
        const unsigned long long m = __hip_atomic_load(hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long ncw = __hip_atomic_load(hp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      /* n_cand | pad << 32 */
        unsigned long long pw[kAhead], pd[kAhead];
#pragma unroll
        for (uint32_t j = 0; j < kAhead; ++j) {
            const uint32_t jj = j < k_stride ? j : k_stride - 1u;                                               /* (stay inside the unit's slots) */
            pw[j] = __hip_atomic_load(cp + 2u * jj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pd[j] = __hip_atomic_load(cp + 2u * jj + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const uint32_t nc = valid ? (uint32_t)ncw : 0u;
        if (valid) {
            matches += m & ~XGM_MATCHES_LOWER_BOUND;
            lower = lower || (m & XGM_MATCHES_LOWER_BOUND) != 0ull;
            fill += nc;
        }
        uint32_t r = 0;
        bool alive = nc > 0u;
        while (__ballot(alive)) {
            if (tkn + 64u > tk_cap || (!theta_valid && tkn >= k)) {        /* (a threshold as soon as k candidates are held: the units' heads usually suffice) */
                sort();
                tkn = tkn < k ? tkn : k;
                if (tkn == k) { theta_valid = true; theta_w = tk_w[k - 1u]; theta_d = tk_d[k - 1u]; }
                wave_lds_fence();
                for (uint32_t i = tkn + lane; i < tk_cap; i += 64u) { tk_w[i] = 0; tk_d[i] = 0xFFFFFFFFu; if (HAS_M) tk_m[i] = 0; }
                wave_lds_fence();
            }
            uint64_t wb = 0;
            unsigned long long ds = 0;
            if (r < kAhead) {
#pragma unroll
                for (uint32_t j = 0; j < kAhead; ++j) if (r == j) { wb = pw[j]; ds = pd[j]; }
            } else if (alive) {
                wb = __hip_atomic_load(cp + 2u * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ds = __hip_atomic_load(cp + 2u * r + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const uint32_t did = (uint32_t)ds, sq = (uint32_t)(ds >> 32);
            const bool take = alive && (!theta_valid || cand_before(wb, did, theta_w, theta_d));
            const uint64_t tm = __ballot(take);
            if (take) { const uint32_t p = tkn + mbcnt(tm); tk_w[p] = wb; tk_d[p] = did; if (HAS_M) tk_m[p] = (uint8_t)sq; }
            tkn += (uint32_t)__popcll(tm);
            wave_lds_fence();
            ++r;
            alive = take && r < nc;
        }
    }
    sort();
    for (int sh = 32; sh > 0; sh >>= 1) {
        matches += (unsigned long long)__shfl_xor((long long)matches, sh);
        fill += (uint32_t)__shfl_xor((int)fill, sh);
    }
    const bool any_lower = __ballot(lower) != 0ull;
    const uint32_t n = fill < k ? fill : k;
    const uint32_t orow = F.row_of ? F.row_of[qi] : qi;
    for (uint32_t i = lane; i < n; i += 64u) {
        xgm_hit hit;
        hit.docid = tk_d[i]; hit.subqs_matched = HAS_M ? (uint32_t)tk_m[i] : subqs_const; hit.weight = __longlong_as_double((long long)tk_w[i]);
        F.hits[(size_t)orow * F.k_stride_out + i] = hit;
    }
    if (lane == 0) {
        xgm_result_hdr r;
        r.n_hits = n;
        r.max_weight_subqs_matched = n ? (HAS_M ? (uint32_t)tk_m[0] : subqs_const) : 0u;
        r.matches_exact = matches | (any_lower ? XGM_MATCHES_LOWER_BOUND : 0ull);
        r.max_attained = n ? __longlong_as_double((long long)tk_w[0]) : 0.0;
        r.max_possible = F.max_possible ? F.max_possible[qi] : 0.0;
        F.hdrs[orow] = r;
    }
}

#endif
