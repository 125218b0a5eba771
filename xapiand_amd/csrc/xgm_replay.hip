/* xgm_search_replay's second half: ProtoMSet's sequential collation replayed ON THE DEVICE over the whole match in docid order
 * (the list xgm_all.hip packs) — so that the answers whose exact value depends on the reference's traversal need neither a download
 * of 16 bytes per match nor a loop on the host:
 *
 *   known_matching_docs (protomset.h:340-400 behind MSet::get_matches_lower_bound / _estimated, i.e. Xapiand's HTTP "total"): a
 *     document reaches ProtoMSet::add when its weight is >= min_weight (matcher.cc:500-505); min_weight is 0 until max_size documents
 *     are held, is set when the heap is made (the (max_size + 1)-th document) and afterwards only by a document that REPLACES the worst
 *     one kept, each time only if known_matching_docs >= check_at_least;
 *   the frozen weight of positional queries (selectpostlist.cc:28-55): once min_weight is positive SelectPostList::vet weighs the next
 *     document of the underlying CONJUNCTION — before test_doc() — caches that weight and serves it for every later match; the list then
 *     carries the conjunction's documents that fail the positional test too, flagged XGM_ALL_NOT_A_MATCH.
 *
 * The collation is sequential by nature (min_weight depends on everything before), but it changes state only at EVENTS — the heap's
 * making and the replacements, O(k log(n / k)) of them — and between two events every document is judged against the same two numbers.
 * One workgroup streams the list in blocks of kBlock entries (all waves load and judge in parallel: one ballot pair per wave and slice),
 * finds the block's first event, counts up to it, lets the event happen (replace the worst kept, find the new worst in parallel) and
 * judges the rest of the block again.  Uniform state is held redundantly by every thread; LDS carries the ballots (double-buffered:
 * one barrier per event-free block), the kept documents and the reductions.  HBM-bound streaming integer / fp64-compare work. */
#include <hip/hip_runtime.h>

#include "xgm_launch.h"
#include "xgm_replay_wave.h"

namespace {

constexpr uint32_t kThreads = 512u, kWaves = kThreads / 64u, kPer = 4u, kBlock = kThreads * kPer, kMasks = kBlock / 64u;

struct ReplayLds {
    unsigned long long pass[2][kMasks], better[2][kMasks];
    double red_w[kWaves];
    uint32_t red_d[kWaves], red_i[kWaves];
    unsigned long long red_g[kWaves];
    uint32_t red_m[kWaves];
    double ev_w; uint32_t ev_d, ev_m;              /* the document an event is about */
};


__global__ __launch_bounds__(kThreads) void xgm_replay_kernel(const xgm_hit* __restrict__ list, unsigned long long n, uint32_t max_size,
                                                              unsigned long long check_at_least, uint32_t frozen_mode, unsigned long long total_matches,
                                                              xgm_hit* __restrict__ out_hits, xgm_replay_out* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ReplayLds& L = *reinterpret_cast<ReplayLds*>(smem);
    double* res_w = reinterpret_cast<double*>(smem + ((sizeof(ReplayLds) + 15) & ~(size_t)15));
    uint32_t* res_d = reinterpret_cast<uint32_t*>(res_w + max_size);
    uint32_t* res_m = res_d + max_size;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;

    /* uniform state, identical in every thread */
    uint32_t size = 0;
    bool heap_built = false, frozen = false, have_star = false, stop = false;
    double min_w = 0.0, worst_w = 0.0, w_star = 0.0;
    uint32_t worst_d = 0, worst_i = 0;
    unsigned long long known = 0;
    uint32_t it = 0;                                   /* ballot buffer parity */
    /* private: the heaviest document this thread showed ProtoMSet, the earliest such (update_max_weight replaces only on >, protomset.h:174-183) */
    double best_w = 0.0; unsigned long long best_g = ~0ull; uint32_t best_m = 0;

    /* the worst document kept (last under the ranking): parallel scan + two-level reduction; every thread leaves with the same answer */
    auto find_worst = [&]() {
        __syncthreads();                                                      /* the kept documents as they are now */
        double w = 0.0; uint32_t d = 0, i = 0xFFFFFFFFu;
        for (uint32_t j = tid; j < size; j += kThreads)
            if (i == 0xFFFFFFFFu || rp_before(w, d, res_w[j], res_d[j])) { w = res_w[j]; d = res_d[j]; i = j; }
        for (int sh = 32; sh > 0; sh >>= 1) {                                 /* butterfly: docids are distinct, every lane ends with the same one */
            const double ow = __shfl_xor(w, sh); const uint32_t od = (uint32_t)__shfl_xor((int)d, sh), oi = (uint32_t)__shfl_xor((int)i, sh);
            if (oi != 0xFFFFFFFFu && (i == 0xFFFFFFFFu || rp_before(w, d, ow, od))) { w = ow; d = od; i = oi; }
        }
        if (lane == 0u) { L.red_w[wave] = w; L.red_d[wave] = d; L.red_i[wave] = i; }
        __syncthreads();
        w = 0.0; d = 0; i = 0xFFFFFFFFu;
        for (uint32_t v = 0; v < kWaves; ++v)
            if (L.red_i[v] != 0xFFFFFFFFu && (i == 0xFFFFFFFFu || rp_before(w, d, L.red_w[v], L.red_d[v]))) { w = L.red_w[v]; d = L.red_d[v]; i = L.red_i[v]; }
        worst_w = w; worst_d = d; worst_i = i;
        __syncthreads();                                                      /* (the reduction slots are free again) */
    };

    const unsigned long long n_blocks = (n + kBlock - 1u) / kBlock;
    /* this thread's entries of a block: slice e, position e * kThreads + tid — positions ascend with (slice, wave, lane) */
    xgm_hit cur[kPer], nxt[kPer];
    auto load = [&](unsigned long long b, xgm_hit* dst) {
#pragma unroll
        for (uint32_t e = 0; e < kPer; ++e) {
            const unsigned long long g = b * kBlock + (unsigned long long)e * kThreads + tid;
            xgm_hit h; h.docid = 0; h.subqs_matched = XGM_ALL_NOT_A_MATCH; h.weight = 0.0;
            if (g < n) h = list[g];
            dst[e] = h;
        }
    };
    if (n_blocks) load(0, nxt);
    for (unsigned long long b = 0; b < n_blocks && !stop; ++b) {
#pragma unroll
        for (uint32_t e = 0; e < kPer; ++e) cur[e] = nxt[e];
        if (b + 1u < n_blocks) load(b + 1u, nxt);                            /* in flight while this block is judged */
        const unsigned long long g0 = b * kBlock;
        uint32_t start = 0;                                                   /* positions below it are done */
        while (start < kBlock && !stop) {
            /* ---- judge: who reaches ProtoMSet::add, who would replace the worst kept ---- */
            const bool filling = size < max_size, steady = !filling && heap_built && max_size != 0u;
            const bool phase2 = frozen;
            const bool star_event = phase2 && steady && rp_before(w_star, 0xFFFFFFFFu, worst_w, worst_d);   /* a later docid never wins a tie */
#pragma unroll
            for (uint32_t e = 0; e < kPer; ++e) {
                const uint32_t pos = e * kThreads + tid;
                const bool valid = g0 + pos < n && pos >= start && !(cur[e].subqs_matched & XGM_ALL_NOT_A_MATCH);
                const double w = phase2 ? w_star : cur[e].weight;
                const bool ps = valid && !(w < min_w);
                const bool bt = ps && steady && (phase2 ? star_event : rp_before(w, cur[e].docid, worst_w, worst_d));
                const unsigned long long mp = __ballot(ps), mb = __ballot(bt);
                if (lane == 0u) { L.pass[it & 1u][e * kWaves + wave] = mp; L.better[it & 1u][e * kWaves + wave] = mb; }
            }
            __syncthreads();
            const unsigned long long* P = L.pass[it & 1u];
            const unsigned long long* B = L.better[it & 1u];
            ++it;
            /* ---- where this step ends: the first event, the heap's making, the end of the room while filling, or the block's end ---- */
            uint32_t total = 0;
            for (uint32_t m = 0; m < kMasks; ++m) total += (uint32_t)__popcll(P[m]);
            uint32_t take = total;                   /* passing documents this step shows ProtoMSet */
            uint32_t ev_pos = kBlock;                /* position of the document the step ends with (kBlock: none, the block is done) */
            bool event = false;
            if (filling || !steady) {
                /* filling: the next (max_size - size) go straight in.  full but heap not made (or max_size == 0): the next one makes it */
                const uint32_t room = filling ? max_size - size : (max_size == 0u ? total : 1u);
                if (total > room || (!filling && max_size != 0u && total >= 1u)) {
                    take = room;
                    uint32_t left = room;            /* position of the room-th passing document */
                    for (uint32_t m = 0; m < kMasks && ev_pos == kBlock; ++m) {
                        const uint32_t c = (uint32_t)__popcll(P[m]);
                        if (left > c) { left -= c; continue; }
                        unsigned long long x = P[m];
                        for (uint32_t j = 1; j < left; ++j) x &= x - 1ull;
                        ev_pos = m * 64u + (uint32_t)__ffsll((long long)x) - 1u;
                    }
                    event = !filling;                /* (the heap's making) */
                }
            } else {
                for (uint32_t m = 0; m < kMasks; ++m)
                    if (B[m]) { ev_pos = m * 64u + (uint32_t)__ffsll((long long)B[m]) - 1u; event = true; break; }
                if (event) {
                    take = 0;
                    for (uint32_t m = 0; m < kMasks; ++m) {
                        const uint32_t lo = m * 64u;
                        if (lo > ev_pos) break;
                        const unsigned long long keep = ev_pos - lo >= 63u ? ~0ull : ((2ull << (ev_pos - lo)) - 1ull);
                        take += (uint32_t)__popcll(P[m] & keep);
                    }
                }
            }
            /* ---- the documents of this step: counted, shown to update_max_weight, appended while filling ---- */
            known += take;
#pragma unroll
            for (uint32_t e = 0; e < kPer; ++e) {
                const uint32_t pos = e * kThreads + tid, m = e * kWaves + wave;
                if (!((P[m] >> lane) & 1ull) || pos > ev_pos) continue;
                const double w = phase2 ? w_star : cur[e].weight;
                const unsigned long long g = g0 + pos;
                if (w > best_w || (w == best_w && best_g != ~0ull && g < best_g)) { best_w = w; best_g = g; best_m = cur[e].subqs_matched; }
                if (filling) {
                    uint32_t r = (uint32_t)__popcll(P[m] & ((1ull << lane) - 1ull));
                    for (uint32_t mm = 0; mm < m; ++mm) r += (uint32_t)__popcll(P[mm]);
                    res_w[size + r] = w; res_d[size + r] = cur[e].docid; res_m[size + r] = cur[e].subqs_matched;
                }
                if (event && pos == ev_pos) { L.ev_w = w; L.ev_d = cur[e].docid; L.ev_m = cur[e].subqs_matched; }
            }
            if (filling) size += take;
            start = ev_pos == kBlock ? kBlock : ev_pos + 1u;
            if (!event) continue;
            /* ---- the event (protomset.h:363-398): the heap is made if it is not, the document is compared with the worst kept ---- */
            bool moved = false;
            if (!heap_built) {
                heap_built = true;
                find_worst();
                if (known >= check_at_least) { min_w = worst_w; moved = true; }
            } else {
                __syncthreads();                                              /* (L.ev_* written) */
            }
            const double ew = L.ev_w; const uint32_t ed = L.ev_d, em = L.ev_m;
            if (rp_before(ew, ed, worst_w, worst_d)) {
                if (tid == 0u) { res_w[worst_i] = ew; res_d[worst_i] = ed; res_m[worst_i] = em; }
                find_worst();
                if (known >= check_at_least) { min_w = worst_w; moved = true; }
            }
            /* ---- positional queries: min_weight turned positive — the weight of the NEXT document of the conjunction is what every
             * later match is given (SelectPostList::vet + get_weight, selectpostlist.cc:28-55) ---- */
            if (frozen_mode && moved && !frozen && min_w > 0.0) {
                frozen = true;
                const unsigned long long gn = g0 + ev_pos + 1ull;
                if (gn < n) { w_star = list[gn].weight; have_star = true; } else stop = true;       /* no document left: the loop ends */
            }
            if (frozen && have_star && w_star < min_w) stop = true;          /* vet() rejects every later document untested */
            /* ... or none can replace a kept one any more (the frozen weight does not beat the worst kept; a later docid never wins a tie) while
             * every one still reaches ProtoMSet::add: the rest of the loop only counts, and the count is known — every match so far was shown
             * (min_weight was <= 0 until the freeze, the frozen weight >= min_weight since), so known_matching_docs ends at the match count */
            if (frozen && have_star && !stop && !rp_before(w_star, 0xFFFFFFFFu, worst_w, worst_d)) { known = total_matches; stop = true; }
        }
    }
    __syncthreads();
    /* ---- the best weight shown and who showed it first ---- */
    {
        double w = best_w; unsigned long long g = best_g; uint32_t m = best_m;
        for (int sh = 32; sh > 0; sh >>= 1) {
            const double ow = __shfl_xor(w, sh); const unsigned long long og = (unsigned long long)__shfl_xor((long long)g, sh); const uint32_t om = (uint32_t)__shfl_xor((int)m, sh);
            if (og != ~0ull && (g == ~0ull || ow > w || (ow == w && og < g))) { w = ow; g = og; m = om; }
        }
        if (lane == 0u) { L.red_w[wave] = w; L.red_g[wave] = g; L.red_m[wave] = m; }
        __syncthreads();
        if (tid == 0u) {
            w = 0.0; g = ~0ull; m = 0;
            for (uint32_t v = 0; v < kWaves; ++v)
                if (L.red_g[v] != ~0ull && (g == ~0ull || L.red_w[v] > w || (L.red_w[v] == w && L.red_g[v] < g))) { w = L.red_w[v]; g = L.red_g[v]; m = L.red_m[v]; }
            xgm_replay_out o;
            o.known_matching_docs = known;
            o.max_weight = g == ~0ull ? 0.0 : w;
            o.max_weight_subqs = g == ~0ull ? 0u : m;
            o.n_hits = size;
            o.frozen = frozen ? 1u : 0u;
            o.reserved = 0;
            o.frozen_weight = have_star ? w_star : 0.0;
            *out = o;
        }
    }
    /* ---- the page: the kept documents in rank order (ProtoMSet::finalise sorts them, protomset.h:657) — ranks by counting ---- */
    for (uint32_t i = tid; i < size; i += kThreads) {
        const double w = res_w[i]; const uint32_t d = res_d[i];
        uint32_t r = 0;
        for (uint32_t j = 0; j < size; ++j) r += rp_before(res_w[j], res_d[j], w, d) ? 1u : 0u;
        xgm_hit h; h.docid = d; h.subqs_matched = res_m[i]; h.weight = w;
        out_hits[r] = h;
    }
}

}  // namespace

int xgm_launch_replay(const xgm_hit* list, uint64_t n, uint32_t max_size, uint64_t check_at_least, bool frozen_mode, uint64_t total_matches,
                      xgm_hit* out_hits, xgm_replay_out* out, hipStream_t stream) {
    if (max_size > XGM_MAX_K) return xgm_launch_error("xgm_replay_kernel", 0, "first + maxitems beyond XGM_MAX_K");
    const size_t smem = ((sizeof(ReplayLds) + 15) & ~(size_t)15) + (size_t)max_size * 16 + 16;
    hipLaunchKernelGGL(xgm_replay_kernel, dim3(1), dim3(kThreads), smem, stream, list, (unsigned long long)n, max_size, (unsigned long long)check_at_least,
                       frozen_mode ? 1u : 0u, (unsigned long long)total_matches, out_hits, out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xgm_launch_error("xgm_replay_kernel", (int)e, hipGetErrorString(e));
    return 0;
}

/* ================================================================================================================================
 * The same collation in PARALLEL, for check_at_least <= first + maxitems (what Xapiand asks: 0, clamped to the page) and no frozen weight.
 * Once the heap is made — at the (k + 1)-th document, and then known_matching_docs >= check_at_least for good — ProtoMSet's state is a
 * function of the PREFIX alone: the kept set is the prefix's top k, min_weight the weight of its worst.  So the list is cut into S
 * segments, one WAVE each (no barriers anywhere):
 *   1. xgm_replay_segtop_kernel   every segment's own top k (the event loop from an empty state), sorted;
 *   2. xgm_replay_prefix_kernel   an exclusive scan under "top k of the union" — a merge of two sorted lists by ranks —, in two levels: groups
 *                                 of kReplayGroup segments in parallel (start[s] = top k of s's group before s; a total per group), then the
 *                                 group totals (gpre[g] = top k of the groups before g): 16 + 32 sequential merges instead of 512;
 *   3. xgm_replay_segcount_kernel every segment merges gpre[its group] with start[s] = the exact state a sequential walk has on arrival,
 *                                 replays its entries and counts; the counts add up to known_matching_docs, the last segment ends with the page.
 * Inside a wave the kept set stays in RANK order in LDS (a replacement is a ballot-counted insertion, the worst is the last entry) and four
 * chunks of 64 entries are in flight.  An OR-5 at 10 M documents (3.7 M matches, k = 100): 11.2 ms in one workgroup — each event costs four
 * workgroup barriers —, 1.97 ms this way, of which 0.39 + 2 x 0.08 + 0.36 ms are these kernels. */
namespace {

__global__ __launch_bounds__(64) void xgm_replay_segtop_kernel(const xgm_hit* __restrict__ list, unsigned long long n, unsigned long long seg_len, uint32_t K,
                                                               xgm_hit* __restrict__ seg_top, uint32_t* __restrict__ seg_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x, s = blockIdx.x;
    WaveState st = wave_state_carve(smem, K);
    const unsigned long long b = (unsigned long long)s * seg_len, e = b + seg_len < n ? b + seg_len : n;
    (void)wave_replay_segment(list, b < n ? b : n, e, K, 0ull, 0ull, st, lane);
    wave_write_sorted(st, seg_top + (size_t)s * K, lane);
    if (lane == 0u) seg_n[s] = st.size;
}

/* Exclusive scan of top-K lists under "merge and keep the best K", one wave per GROUP of `gs` lists: start[s] = top K of the lists of s's group
 * before s; total[g] = top K of the whole group (when asked for).  An entry's place in a merge = its index + the entries of the other list that
 * rank before it (a binary search).  Two levels (groups of segments, then the groups' totals with one group of all of them) keep the sequential
 * depth at gs + S / gs merges instead of S. */
__global__ __launch_bounds__(64) void xgm_replay_prefix_kernel(const xgm_hit* __restrict__ seg_top, const uint32_t* __restrict__ seg_n, uint32_t S, uint32_t gs,
                                                               uint32_t K, xgm_hit* __restrict__ start, uint32_t* __restrict__ start_n,
                                                               xgm_hit* __restrict__ total, uint32_t* __restrict__ total_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    xgm_hit* cur = reinterpret_cast<xgm_hit*>(smem);
    xgm_hit* oth = cur + K;
    xgm_hit* out = oth + K;
    const uint32_t lane = threadIdx.x, g = blockIdx.x;
    const uint32_t s0 = g * gs, s1 = s0 + gs < S ? s0 + gs : S;
    uint32_t n_cur = 0;
    /* the next list is requested while this one is merged (two entries per lane: K <= 128 in one go; a longer tail is read when needed) */
    xgm_hit pre0, pre1; pre0.docid = 0; pre0.subqs_matched = 0; pre0.weight = 0.0; pre1 = pre0;
    uint32_t pre_n = s0 < s1 ? seg_n[s0] : 0u;
    if (lane < pre_n) pre0 = seg_top[(size_t)s0 * K + lane];
    if (lane + 64u < pre_n) pre1 = seg_top[(size_t)s0 * K + lane + 64u];
    for (uint32_t s = s0; s < s1; ++s) {
        for (uint32_t i = lane; i < n_cur; i += 64u) start[(size_t)s * K + i] = cur[i];
        if (lane == 0u) start_n[s] = n_cur;
        if (s + 1u == s1 && total == nullptr) break;
        const uint32_t n_b = pre_n;
        if (lane < n_b) oth[lane] = pre0;
        if (lane + 64u < n_b) oth[lane + 64u] = pre1;
        for (uint32_t i = lane + 128u; i < n_b; i += 64u) oth[i] = seg_top[(size_t)s * K + i];
        if (s + 1u < s1) {
            pre_n = seg_n[s + 1u];
            if (lane < pre_n) pre0 = seg_top[(size_t)(s + 1u) * K + lane];
            if (lane + 64u < pre_n) pre1 = seg_top[(size_t)(s + 1u) * K + lane + 64u];
        }
        wave_lds_sync();
        for (uint32_t i = lane; i < n_cur; i += 64u) {
            const uint32_t r = i + count_before(oth, n_b, cur[i].weight, cur[i].docid);
            if (r < K) out[r] = cur[i];
        }
        for (uint32_t i = lane; i < n_b; i += 64u) {
            const uint32_t r = i + count_before(cur, n_cur, oth[i].weight, oth[i].docid);
            if (r < K) out[r] = oth[i];
        }
        wave_lds_sync();
        n_cur = n_cur + n_b < K ? n_cur + n_b : K;
        xgm_hit* t = cur; cur = out; out = t;
    }
    if (total != nullptr) {
        for (uint32_t i = lane; i < n_cur; i += 64u) total[(size_t)g * K + i] = cur[i];
        if (lane == 0u) total_n[g] = n_cur;
    }
}

__global__ __launch_bounds__(64) void xgm_replay_segcount_kernel(const xgm_hit* __restrict__ list, unsigned long long n, unsigned long long seg_len, uint32_t K,
                                                                 unsigned long long check_at_least, const xgm_hit* __restrict__ start,
                                                                 const uint32_t* __restrict__ start_n, const xgm_hit* __restrict__ gpre,
                                                                 const uint32_t* __restrict__ gpre_n, uint32_t gs, uint32_t S, xgm_hit* __restrict__ out_hits,
                                                                 xgm_replay_out* __restrict__ out, unsigned long long* __restrict__ known_sum) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x, s = blockIdx.x;
    WaveState st = wave_state_carve(smem, K);
    unsigned long long base = 0;
    if (s != 0u) {
        /* the state a sequential walk has when it reaches this segment: the prefix's top K kept — the merge of its group's lists before it (start[s])
         * and the groups before (gpre[g]) —, the heap made (the prefix holds more than K documents: seg_len > K), min_weight = the worst kept
         * (check_at_least <= K was passed when the heap was made) */
        xgm_hit* la = reinterpret_cast<xgm_hit*>(smem + (((size_t)K + 1u) * 16u));
        xgm_hit* lb = la + K;
        const uint32_t g = s / gs, n_a = gpre_n[g], n_b = start_n[s];
        for (uint32_t i = lane; i < n_a; i += 64u) la[i] = gpre[(size_t)g * K + i];
        for (uint32_t i = lane; i < n_b; i += 64u) lb[i] = start[(size_t)s * K + i];
        wave_lds_sync();
        for (uint32_t i = lane; i < n_a; i += 64u) {
            const xgm_hit h = la[i];
            const uint32_t r = i + count_before(lb, n_b, h.weight, h.docid);
            if (r < K) { st.w[r] = h.weight; st.d[r] = h.docid; st.m[r] = h.subqs_matched; }
        }
        for (uint32_t i = lane; i < n_b; i += 64u) {
            const xgm_hit h = lb[i];
            const uint32_t r = i + count_before(la, n_a, h.weight, h.docid);
            if (r < K) { st.w[r] = h.weight; st.d[r] = h.docid; st.m[r] = h.subqs_matched; }
        }
        st.size = n_a + n_b < K ? n_a + n_b : K;
        st.heap_built = true;                                                  /* (in rank order: the worst is the last) */
        wave_lds_sync();
        if (st.size) { st.worst_w = st.w[st.size - 1u]; st.worst_d = st.d[st.size - 1u]; }
        st.min_w = K != 0u ? st.worst_w : 0.0;
        base = check_at_least;
    }
    const unsigned long long b = (unsigned long long)s * seg_len, e = b + seg_len < n ? b + seg_len : n;
    const unsigned long long known = wave_replay_segment(list, b < n ? b : n, e, K, check_at_least, base, st, lane);
    if (lane == 0u && known) atomicAdd(known_sum, known);
    if (s + 1u == S) {
        wave_write_sorted(st, out_hits, lane);
        if (lane == 0u) {
            xgm_replay_out o;
            o.known_matching_docs = 0;            /* (the host adds *known_sum once every segment has finished) */
            o.max_weight = 0.0; o.frozen_weight = 0.0; o.max_weight_subqs = 0; o.n_hits = st.size; o.frozen = 0; o.reserved = 1;
            *out = o;
        }
    }
}

}  // namespace

constexpr uint32_t kReplayGroup = 16u;      /* segments per group of the two-level scan */

/* bytes of device scratch the parallel replay needs for S segments of a page of K */
size_t xgm_replay_parallel_bytes(uint32_t S, uint32_t K) {
    const size_t k = K ? K : 1u, G = (S + kReplayGroup - 1u) / kReplayGroup;
    return (2 * (size_t)S + 2 * G) * k * sizeof(xgm_hit) + (2 * (size_t)S + 2 * G) * 4 + 64;
}

int xgm_launch_replay_parallel(const xgm_hit* list, uint64_t n, uint32_t K, uint64_t check_at_least, uint32_t S, uint64_t seg_len, void* scratch,
                               xgm_hit* out_hits, xgm_replay_out* out, unsigned long long* known_sum, hipStream_t stream) {
    if (K == 0u || K > kSegMaxK || check_at_least > K || S == 0u || seg_len <= K) return xgm_launch_error("parallel replay", 0, "bad arguments");
    const uint32_t G = (S + kReplayGroup - 1u) / kReplayGroup;
    xgm_hit* seg_top = (xgm_hit*)scratch;
    xgm_hit* start = seg_top + (size_t)S * K;
    xgm_hit* gtop = start + (size_t)S * K;
    xgm_hit* gpre = gtop + (size_t)G * K;
    uint32_t* seg_n = (uint32_t*)(gpre + (size_t)G * K);
    uint32_t* start_n = seg_n + S;
    uint32_t* gtop_n = start_n + S;
    uint32_t* gpre_n = gtop_n + G;
    hipError_t e = hipMemsetAsync(known_sum, 0, 8, stream);
    if (e != hipSuccess) return xgm_launch_error("hipMemsetAsync", (int)e, hipGetErrorString(e));
    const size_t lds_state = ((size_t)K + 1u) * 16 + (size_t)2 * K * sizeof(xgm_hit), lds_prefix = (size_t)3 * K * sizeof(xgm_hit) + 16;
    hipLaunchKernelGGL(xgm_replay_segtop_kernel, dim3(S), dim3(64), lds_state, stream, list, (unsigned long long)n, (unsigned long long)seg_len, K, seg_top, seg_n);
    hipLaunchKernelGGL(xgm_replay_prefix_kernel, dim3(G), dim3(64), lds_prefix, stream, seg_top, seg_n, S, kReplayGroup, K, start, start_n, gtop, gtop_n);
    hipLaunchKernelGGL(xgm_replay_prefix_kernel, dim3(1), dim3(64), lds_prefix, stream, gtop, gtop_n, G, G, K, gpre, gpre_n, (xgm_hit*)nullptr, (uint32_t*)nullptr);
    hipLaunchKernelGGL(xgm_replay_segcount_kernel, dim3(S), dim3(64), lds_state, stream, list, (unsigned long long)n, (unsigned long long)seg_len, K,
                       (unsigned long long)check_at_least, start, start_n, gpre, gpre_n, kReplayGroup, S, out_hits, out, known_sum);
    e = hipGetLastError();
    if (e != hipSuccess) return xgm_launch_error("parallel replay kernels", (int)e, hipGetErrorString(e));
    return 0;
}
