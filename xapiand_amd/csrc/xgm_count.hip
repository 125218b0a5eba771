/* ProtoMSet's known_matching_docs for a whole BATCH of plain conjunctions (include/xgm.h, XGM_REPLAY_BATCH_COUNT): what xgm_search_replay(XGM_REPLAY_COUNT)
 * gives one query at a time.  The reference's matcher shows ProtoMSet::add a document when its weight is >= min_weight (matcher.cc:500-505), and
 * min_weight — once first + maxitems = K documents are held and check_at_least (= K here: what Enquire::get_mset clamps Xapiand's 0 to) are counted — is
 * the weight of the worst document kept (protomset.h:340-400): how many documents are shown depends on the ORDER they come in, docid order.
 *
 * xgm_andw_all_kernel's units have left (a) their own top-K lists, as always, (b) EVERY match with its weight, in docid order (xgm_all_out).  Once the heap
 * is made ProtoMSet's state is a function of the PREFIX alone — the kept set is the prefix's top K, min_weight its worst — so:
 *   xgm_count_scan_kernel    one wave per query: an exclusive scan of the units' top-K lists under "merge, keep the best K" in stripe order; unit u gets the
 *                            state the reference's walk has when it arrives there (a version of the K kept documents, how many documents came before); the
 *                            scan's total is the query's page (ProtoMSet::finalise, protomset.h:657) — this kernel is also the merge;
 *   xgm_count_units_kernel   one wave per unit: ProtoMSet's walk over the unit's own matches from that state (xgm_replay_wave.h: 64 documents judged per step
 *                            against min_weight and the worst kept, the kept set in rank order in LDS); what it counts is added to the query's figure.
 * K <= 64 (the bodies that list serve pages of at most 64): one kept document per lane in the scan.  HBM-bound streaming of 16 bytes per match. */
#include <hip/hip_runtime.h>

#include "xgm_launch.h"
#include "xgm_wave.h"
#include "xgm_replay_wave.h"

namespace {

/* unit_ver [n_work]: which VERSION of the query's kept set a unit arrives with; states [(n_work + nq)][k_stride_c]: the versions, query qi's from row
 * goff[qi] + qi on (at most one more than it has units).  The kept set changes only at a unit whose best document beats the worst kept (or while fewer than K
 * are kept): the units are looked at 64 at a time — headers and best candidates in one round trip — and only those are merged, one after the other
 * (measured: a merge per unit, each behind two dependent loads, took 0.73 ms per batch of C2; the heaviest query has 1 221 units). */
__global__ __launch_bounds__(64) void xgm_count_scan_kernel(const xgm_dev_query* __restrict__ queries, uint32_t nq, const uint32_t* __restrict__ goff,
                                                            const xgm_cand* __restrict__ cand, const xgm_group_hdr* __restrict__ ghdr, uint32_t k_stride_c,
                                                            xgm_cand* __restrict__ states, uint32_t* __restrict__ unit_ver, unsigned long long* __restrict__ unit_before,
                                                            const double* __restrict__ max_possible, const uint32_t* __restrict__ row_of,
                                                            xgm_hit* __restrict__ hits, xgm_result_hdr* __restrict__ hdrs, unsigned long long* __restrict__ extra,
                                                            uint32_t k_stride_out) {
    __shared__ xgm_cand merged[64];
    const uint32_t qi = blockIdx.x, lane = threadIdx.x;
    if (qi >= nq) return;
    const xgm_dev_query& q = queries[qi];
    const uint32_t K = rfl32(q.k);
    const uint32_t g0 = goff[qi], U = goff[qi + 1] - g0;
    const uint32_t orow = row_of ? row_of[qi] : qi;
    xgm_cand* my_states = states + (size_t)(g0 + qi) * k_stride_c;
    /* lane i < n_cur: the i-th best document of the units walked so far */
    uint64_t cw = 0; uint32_t cd = 0xFFFFFFFFu, cm = 0;
    uint32_t n_cur = 0, ver = 0;
    unsigned long long before = 0;
    bool fallback = K > 64u || K == 0u;
    for (uint32_t u0 = 0; u0 < U; u0 += 64u) {
        const uint32_t u = u0 + lane;
        const bool valid = u < U;
        /* lane = unit: its header and its best candidate */
        unsigned long long hm = 0; uint32_t hn = 0, hf = 0;
        xgm_cand first; first.wbits = 0; first.did = 0xFFFFFFFFu; first.subqs = 0;
        if (valid) {
            const xgm_group_hdr& h = ghdr[g0 + u];
            hm = h.matches & ~XGM_MATCHES_LOWER_BOUND; hn = h.n_cand < K ? h.n_cand : K; hf = h.c_pad[1];
            first = cand[(size_t)(g0 + u) * k_stride_c];           /* (read whatever n_cand says: inside the unit's slots) */
        }
        fallback = fallback || __ballot(valid && (hf & (XGM_ALL_OVERFLOW | XGM_ALL_DECLINED)) != 0u) != 0ull;
        /* how many documents the walk has seen when it reaches each unit (a unit holds < 2^19 matches: the wave's sum fits 32 bits) */
        const uint32_t m32 = (uint32_t)hm, incl = wave_incl_scan(m32);
        if (valid) unit_before[g0 + u] = before + (incl - m32);
        before += rl32(incl, 63u);
        uint64_t pending = __ballot(valid && hn > 0u);
        uint32_t my_ver = ver;
        while (pending) {
            /* the first pending unit that changes the kept set: it is not full yet, or the unit's best beats the worst kept */
            const uint64_t ww = rl64(cw, n_cur ? n_cur - 1u : 0u); const uint32_t wd = rl32(cd, n_cur ? n_cur - 1u : 0u);
            const uint64_t chg = pending & __ballot(n_cur < K || cand_before(first.wbits, first.did, ww, wd));
            if (!chg) break;
            const uint32_t L = (uint32_t)__builtin_ctzll(chg);
            /* (the units up to and including L arrive with the current version) */
            const uint32_t nc = rl32(hn, L);
            xgm_cand o; o.wbits = 0; o.did = 0xFFFFFFFFu; o.subqs = 0;
            if (lane < nc) o = cand[(size_t)(g0 + u0 + L) * k_stride_c + lane];
            uint32_t r_c = lane, r_o = lane;
            const uint32_t n_max = n_cur > nc ? n_cur : nc;
            for (uint32_t j = 0; j < n_max; ++j) {
                const uint64_t ow = rl64(o.wbits, j), kw = rl64(cw, j);
                const uint32_t od = rl32(o.did, j), kd = rl32(cd, j);
                r_c += (j < nc && cand_before(ow, od, cw, cd)) ? 1u : 0u;
                r_o += (j < n_cur && cand_before(kw, kd, o.wbits, o.did)) ? 1u : 0u;
            }
            wave_lds_sync();
            if (lane < n_cur && r_c < K) { xgm_cand c; c.wbits = cw; c.did = cd; c.subqs = cm; merged[r_c] = c; }
            if (lane < nc && r_o < K) merged[r_o] = o;
            wave_lds_sync();
            n_cur = n_cur + nc < K ? n_cur + nc : K;
            if (lane < n_cur) { cw = merged[lane].wbits; cd = merged[lane].did; cm = merged[lane].subqs; }
            ++ver;
            if (lane < n_cur) { xgm_cand c; c.wbits = cw; c.did = cd; c.subqs = cm; my_states[(size_t)ver * k_stride_c + lane] = c; }
            if (lane > L) my_ver = ver;                            /* the units behind L arrive with the new version (or a later one) */
            pending &= ~((2ull << L) - 1ull);
        }
        if (valid) unit_ver[g0 + u] = my_ver;
    }
    /* ---- the page (ProtoMSet::finalise: rank order) and the header; the count is added by xgm_count_units_kernel ---- */
    if (lane < n_cur && !fallback) {
        xgm_hit hit; hit.docid = cd; hit.subqs_matched = cm; hit.weight = __longlong_as_double((long long)cw);
        hits[(size_t)orow * k_stride_out + lane] = hit;
    }
    const uint64_t w0 = rl64(cw, 0u);
    const uint32_t m0 = rl32(cm, 0u);
    if (lane == 0u) {
        xgm_result_hdr r;
        r.n_hits = fallback ? 0u : n_cur;
        r.max_weight_subqs_matched = (n_cur && !fallback) ? m0 : 0u;
        r.matches_exact = fallback ? 0ull : before;
        r.max_attained = (n_cur && !fallback) ? __longlong_as_double((long long)w0) : 0.0;      /* (true weights: the heaviest document is always shown) */
        r.max_possible = max_possible ? max_possible[qi] : 0.0;
        hdrs[orow] = r;
        extra[orow] = fallback ? XGM_EXTRA_FALLBACK : 0ull;
    }
}

__global__ __launch_bounds__(64) void xgm_count_units_kernel(const xgm_dev_query* __restrict__ queries, const xgm_work* __restrict__ work, uint32_t n_work,
                                                             const uint32_t* __restrict__ goff, const xgm_group_hdr* __restrict__ ghdr, uint32_t k_stride_c, xgm_all_out lists,
                                                             const xgm_cand* __restrict__ states, const uint32_t* __restrict__ unit_ver,
                                                             const unsigned long long* __restrict__ unit_before,
                                                             const uint32_t* __restrict__ row_of, unsigned long long* __restrict__ extra) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x;
    if (blockIdx.x >= n_work) return;
    const xgm_work wk = work[blockIdx.x];
    const uint32_t qi = rfl32(wk.qi), slot = rfl32(wk.slot);
    const uint32_t orow = row_of ? row_of[qi] : qi;
    const uint32_t K = rfl32(queries[qi].k);
    const xgm_group_hdr& h = ghdr[slot];
    const unsigned long long n_u = h.matches & ~XGM_MATCHES_LOWER_BOUND;
    if (n_u == 0ull || K == 0u || K > 64u) return;
    if (__hip_atomic_load(&extra[orow], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & XGM_EXTRA_FALLBACK) return;      /* (set by the scan kernel, a launch earlier) */
    const unsigned long long before = unit_before[slot];
    /* the state the walk arrives with: the prefix's top K kept, in rank order */
    WaveState st = wave_state_carve(smem, K);
    const uint32_t n_cur = before < K ? (uint32_t)before : K;
    if (lane < n_cur) {
        const xgm_cand c = states[((size_t)goff[qi] + qi + unit_ver[slot]) * k_stride_c + lane];
        st.w[lane] = __longlong_as_double((long long)c.wbits); st.d[lane] = c.did; st.m[lane] = c.subqs;
    }
    st.size = n_cur;
    wave_lds_sync();
    if (before > K) {                                              /* the heap is made and check_at_least (= K) counted: min_weight is the worst kept document's */
        st.heap_built = true;
        st.worst_w = st.w[K - 1u]; st.worst_d = st.d[K - 1u];
        st.min_w = st.worst_w;
    }
    unsigned long long known = 0, left = n_u;
    for (uint32_t c = 0; c < XGM_ALL_CHUNKS && left; ++c) {
        const unsigned long long size = (unsigned long long)(XGM_ALL_CHUNK0 << c), len = left < size ? left : size;
        const xgm_hit* seg = lists.arena + lists.chunk_tab[(size_t)slot * XGM_ALL_CHUNKS + c];
        known += wave_replay_segment(seg, 0ull, len, K, (unsigned long long)K, before + known, st, lane);
        left -= len;
    }
    if (lane == 0u && known) atomicAdd(&extra[orow], known);
}

}  // namespace

int xgm_launch_count_finish(const xgm_dev_query* queries, uint32_t nq, const xgm_work* work, uint32_t n_work, const uint32_t* goff, const xgm_cand* cand,
                            const xgm_group_hdr* ghdr, uint32_t k_stride_c, const xgm_all_out& lists, xgm_cand* states, uint32_t* unit_ver, unsigned long long* unit_before,
                            const double* max_possible, const uint32_t* row_of, xgm_hit* hits, xgm_result_hdr* hdrs, unsigned long long* extra, uint32_t k_stride_out,
                            hipStream_t stream) {
    if (nq == 0 || n_work == 0) return 0;
    if (k_stride_c > 64u) return xgm_launch_error("xgm_count kernels", 0, "pages of at most 64");
    hipLaunchKernelGGL(xgm_count_scan_kernel, dim3(nq), dim3(64), 0, stream, queries, nq, goff, cand, ghdr, k_stride_c, states, unit_ver, unit_before, max_possible, row_of,
                       hits, hdrs, extra, k_stride_out);
    hipLaunchKernelGGL(xgm_count_units_kernel, dim3(n_work), dim3(64), (size_t)(64 + 1) * 16, stream, queries, work, n_work, goff, ghdr, k_stride_c, lists, states,
                       unit_ver, unit_before, row_of, extra);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xgm_launch_error("xgm_count kernels", (int)e, hipGetErrorString(e));
    return 0;
}
