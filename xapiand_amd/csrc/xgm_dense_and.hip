/* xgm_dense_kernel — the conjunction whose every term has probe containers, as a kernel of its own (gfx950, wave64).
 *
 * Why it exists (round 3): xgm_andw_kernel serves every conjunction shape from one body — block-decoded rarest terms, sided
 * operators, the positional filter with its LDS staging — and needs 128 VGPRs (+ spills) for the plain AND-3 the headline is
 * measured on, 168 for the positional instantiation: 4 and 3 waves per SIMD.  The path that dominates both C2 and C5 is much
 * simpler: every term is dense, so the candidates of a stripe ARE the bits of the AND of the terms' bitmaps, and a candidate
 * costs one wdf byte per term and its document length.  Measured on the MI355X, occupancy beats everything else on this
 * latency-bound path (4 waves with spills 0.40 ms, 3 waves without 0.43 ms, 2 waves 4.0 vs 3.2 ms for the phrase class), so this
 * kernel is written for 8 waves per SIMD: <= 64 VGPRs, ~3 KB of LDS per wave, T <= 4 terms unrolled, k <= 64.
 *
 * One wave per work unit (a query, <= 32 stripes), no workgroup barrier:
 *   producer  per stripe: one 16-byte load per lane and term of the containers' bitmaps (next stripe's in flight), AND; the set
 *             bits go to a ring of docids in LDS, two per lane and pass (wave prefix sum of the counts) — the order in which
 *             candidates are weighed does not matter to a top-k under a total order;
 *   consumer  64 candidates per round, every lane busy: T one-byte probes + the document length, BM25 in fp64 with the
 *             reference's operation order (BM25Weight::get_sumpart, bm25weight.cc:170-181; MultiAndPostList::get_weight,
 *             multiandpostlist.cc:150-160), threshold test against the unit's k-th best, survivors into the wave's top-k buffer
 *             (bitonic selection when it fills: ProtoMSet::add, protomset.h:340-400, order of msetcmp.cc:55-62);
 *   PHRASE / NEAR with positional pruning (XGM_QF_POSPRUNE: Xapiand's check_at_least = 0): a candidate is weighed first and its
 *             positions are tested only if it can still enter the top k (unit's k-th best and the query-wide histogram of the
 *             matches taken so far) — survivors are rare, so they take the serial predicates straight from HBM
 *             (xgm_posfilter.h: ExactPhrasePostList / PhrasePostList / NearPostList::test_doc).
 * Replaces, for these shapes, the queue path of xgm_andw_kernel; same inputs (work units of plan_batch), same outputs
 * (xgm_cand / xgm_group_hdr for xgm_merge_kernel).  Compile with -ffp-contract=off.
 */
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>

#include "xgm_device.h"
#include "xgm_launch.h"
#include "xgm_wave.h"
#include "xgm_posfilter.h"

namespace {

constexpr uint32_t kDenseT = 4;            /* terms, unrolled */
constexpr uint32_t kDenseSpg = 32;         /* stripes per unit (plan_batch's bound for the wave kernels) */
constexpr uint32_t kDenseCap = 128;        /* top-k buffer: k <= 64 kept + a round of 64 */
constexpr uint32_t kDenseRing = 256;       /* candidate ring (docids) */

constexpr uint32_t kDenseSq = 128;         /* positional survivors waiting for their test (< 64 left by a drain + <= 64 of a round) */

__host__ __device__ inline size_t dense_wave_bytes(bool phrase) {
    return (size_t)kDenseCap * 8 + (size_t)kDenseCap * 4 + (size_t)kDenseT * kDenseSpg * 4 + (size_t)kDenseRing * 4 + (phrase ? (size_t)kDenseSq * 16 : 0);
}

typedef uint32_t dense_u4 __attribute__((ext_vector_type(4)));

/* bitonic sort of kDenseCap candidates by one wave; best first */
__device__ __forceinline__ void dense_topk_sort(uint64_t* w, uint32_t* d, uint32_t lane) {
    for (uint32_t size = 2; size <= kDenseCap; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            wave_lds_fence();
#pragma unroll
            for (uint32_t i0 = 0; i0 < (kDenseCap >> 1); i0 += 64u) {
                const uint32_t i = i0 + lane;
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

/* The positional test of ONE candidate (rare: only what can still enter the top k gets here), out of line so that its registers and
 * private arrays do not count against the weighing loop.  off_t = the candidate's containers (16-byte units), w1_t = its wdf + 1
 * bytes.  Where the document's positions of a term start: the base of its 64-slot bucket + the wdf of the bucket's earlier
 * documents (the candidate's 64-byte sector of wdf + 1 bytes) — then ExactPhrasePostList / PhrasePostList / NearPostList::test_doc. */
__device__ __attribute__((noinline)) bool dense_positions_ok(const xgm_seg_dev& seg, const xgm_dev_query& q, uint32_t T, uint32_t off0, uint32_t off1,
                                                             uint32_t off2, uint32_t off3, uint32_t slot, uint32_t w10, uint32_t w11, uint32_t w12,
                                                             uint32_t w13) {
    const uint32_t W = 1u << seg.stripe_bits, NW = W / 32u;
    const uint32_t offs[kDenseT] = {off0, off1, off2, off3}, w1s[kDenseT] = {w10, w11, w12, w13};
    PosList pl[XGM_PHRASE_MAX_TERMS];
    for (uint32_t t = 0; t < T && t < kDenseT; ++t) {
        const unsigned char* wbytes = seg.dense_data + (size_t)offs[t] * 16 + (size_t)NW * 4;
        const uint32_t* sp = reinterpret_cast<const uint32_t*>(wbytes + (slot & ~63u));
        uint32_t pos = reinterpret_cast<const uint32_t*>(wbytes + W)[slot >> 6];
        const uint32_t kk = slot & 63u;
        for (uint32_t j = 0; j < 16u && 4u * j < kk; ++j) {
            const uint32_t xw = sp[j];
            const uint32_t nz1 = ((xw | ((xw & 0x7F7F7F7Fu) + 0x7F7F7F7Fu)) >> 7) & 0x01010101u;     /* 1 in every present slot's byte */
            const uint32_t rel = kk - 4u * j;                                                          /* bytes of this word before the slot */
            const uint32_t m = rel >= 4u ? 0xFFFFFFFFu : ((1u << (8u * rel)) - 1u);
            pos = __builtin_amdgcn_sad_u8((xw - nz1) & m, 0u, pos);                                    /* += Σ (wdf + 1 - 1) of those bytes */
        }
        const uint32_t id = q.term_id[t];
        const uint32_t w16 = seg.term_flags[id] & XGM_TF_POS16;
        pl[t].p = seg.positions + seg.term_pos[id] + (size_t)pos * (w16 ? 2u : 4u);
        pl[t].n = w1s[t] - 1u;
        pl[t].w16 = w16;
    }
    return posfilter_slow(pl, q, T);
}

#ifndef XGM_DENSE_WAVES
#define XGM_DENSE_WAVES 6          /* waves per SIMD the register allocator must allow (A/B: tools/ab_build.sh) */
#endif
#ifndef XGM_DENSE_PHRASE_WAVES
#define XGM_DENSE_PHRASE_WAVES 6
#endif

template <bool PHRASE, bool TALLY>
__global__ __launch_bounds__(XGM_WG, PHRASE ? XGM_DENSE_PHRASE_WAVES : XGM_DENSE_WAVES) void xgm_dense_kernel(xgm_seg_dev seg, const xgm_dev_query* __restrict__ queries,
                                                              const xgm_work* __restrict__ work, uint32_t n_work, uint32_t k_stride,
                                                              xgm_cand* __restrict__ cand_out, xgm_group_hdr* __restrict__ ghdr_out,
                                                              uint32_t* __restrict__ hist_all) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t unit = blockIdx.x * XGM_WAVES + wave;
    if (unit >= n_work) return;                                    /* no barriers below: early exit is safe */
    /* the unit and everything of its query the loops use are wave-uniform: pinned to scalar registers (the compiler cannot prove
     * threadIdx.x >> 6 uniform and would otherwise keep the query's doubles in vector registers, lane by lane) */
    const xgm_work wk0 = work[unit];
    xgm_work wk;
    wk.qi = rfl32(wk0.qi); wk.s_begin = rfl32(wk0.s_begin); wk.s_end = rfl32(wk0.s_end); wk.slot = rfl32(wk0.slot);
    const xgm_dev_query& q = queries[wk.qi];
    const uint32_t SB = seg.stripe_bits, W = 1u << SB, NW = W / 32u;
    const uint32_t T = rfl32(q.n_terms), k = rfl32(q.k);
    const double u_len_factor = rl_f64(q.len_factor, 0), u_min_normlen = rl_f64(q.min_normlen, 0), u_k1 = rl_f64(q.k1, 0), u_b = rl_f64(q.b, 0);
    const double u_omb = 1.0 - u_b;                                 /* (1 - param_b), bm25weight.cc:176 */
    double u_tw[kDenseT];
#pragma unroll
    for (uint32_t t = 0; t < kDenseT; ++t) u_tw[t] = rl_f64(q.termweight[t], 0);
    const float f_len_factor = (float)u_len_factor, f_min_normlen = (float)u_min_normlen, f_k1 = (float)u_k1, f_b = (float)u_b, f_omb = (float)u_omb;
    float f_tw[kDenseT];
#pragma unroll
    for (uint32_t t = 0; t < kDenseT; ++t) f_tw[t] = (float)u_tw[t];
    const unsigned long long t_unit_start = __builtin_readcyclecounter();

    unsigned char* base = smem + (size_t)wave * dense_wave_bytes(PHRASE);
    uint64_t* tk_w = reinterpret_cast<uint64_t*>(base);
    uint32_t* tk_d = reinterpret_cast<uint32_t*>(base + (size_t)kDenseCap * 8);
    uint32_t* rs = reinterpret_cast<uint32_t*>(base + (size_t)kDenseCap * 12);
    uint32_t* ring = reinterpret_cast<uint32_t*>(base + (size_t)kDenseCap * 12 + (size_t)kDenseT * kDenseSpg * 4);
    uint64_t* sq_w = reinterpret_cast<uint64_t*>(base + dense_wave_bytes(false));           /* PHRASE only: the survivor queue */
    uint32_t* sq_d = reinterpret_cast<uint32_t*>(base + dense_wave_bytes(false) + (size_t)kDenseSq * 8);
    uint32_t* sq_v = sq_d + kDenseSq;

    const uint32_t s_begin = wk.s_begin, s_end = wk.s_end;
    const bool empty = (q.flags & XGM_QF_EMPTY) || s_begin >= s_end || k == 0;
    const uint32_t n_local = empty ? 0u : (s_end - s_begin < kDenseSpg ? s_end - s_begin : kDenseSpg);

    /* traffic tallies (xgm_group_hdr), wave-uniform */
    uint32_t cn_bmpw = 0, cn_probe = 0, cn_dl = 0, cn_aux = 0, cn_probe_raw = 0, cn_dl_raw = 0, q_cands = 0;
    unsigned long long cn_pos = 0;
    auto tally_sectors = [&](bool valid, uint32_t key, uint32_t sh) {
        const uint32_t prev = (uint32_t)__shfl_up((int)key, 1);
        return (uint32_t)__popcll(__ballot(valid && (lane == 0u || (prev >> sh) != (key >> sh))));
    };

#pragma unroll
    for (uint32_t i = 0; i < kDenseCap; i += 64u) { tk_w[i + lane] = 0; tk_d[i + lane] = 0xFFFFFFFFu; }
    /* container offsets of every (term, stripe of the unit); 0 = the term has no posting in that stripe */
#pragma unroll
    for (uint32_t i = 0; i < kDenseT * kDenseSpg; i += 64u) {
        const uint32_t e = i + lane, t = e / kDenseSpg, x = e % kDenseSpg;
        uint32_t off = 0;
        if (t < T && x < n_local) {
            const uint32_t id = q.term_id[t];
            const uint32_t dn = id != 0xFFFFFFFFu ? seg.dense_id[id] : 0xFFFFFFFFu;
            if (dn != 0xFFFFFFFFu) off = seg.dense_dir[(size_t)dn * seg.n_stripes + (s_begin + x)];
        }
        rs[e] = off;
    }
    if (TALLY) { cn_aux += T * n_local; }
    wave_lds_fence();

    /* PHRASE: the units of a query share what they learn (see xgm_andw_kernel): a 256-bucket histogram of the weights of the
     * positional matches taken so far; the highest bucket with >= k matches at or above it bounds the final k-th weight from below */
    uint32_t* hist_g = nullptr;
    int hbase = 0;
    uint64_t theta_glob = 0;
    if (PHRASE && hist_all && !empty) {
        double mp = 0.0;
        for (uint32_t t = 0; t < T; ++t) mp += q.ub[t];
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
        const uint64_t okm = __ballot(suf >= k);
        if (okm) {
            const uint32_t Lh = 63u - (uint32_t)__builtin_clzll(okm);
            const uint32_t cum = rl32(suf, Lh) - rl32(s4, Lh);
            const uint32_t c3 = rl32(hc[3], Lh), c2 = rl32(hc[2], Lh), c1 = rl32(hc[1], Lh);
            uint32_t bsel = 4u * Lh;
            if (cum + c3 >= k) bsel = 4u * Lh + 3u;
            else if (cum + c3 + c2 >= k) bsel = 4u * Lh + 2u;
            else if (cum + c3 + c2 + c1 >= k) bsel = 4u * Lh + 1u;
            if (bsel > 0u) {
                const uint64_t tb = (uint64_t)((uint32_t)hbase + bsel) << 47;
                theta_glob = tb > theta_glob ? tb : theta_glob;
            }
        }
    };

    uint32_t tkn = 0;                                              /* wave-uniform top-k state */
    bool theta_valid = false;
    uint64_t theta_w = 0;
    uint32_t theta_d = 0;
    uint32_t matches = 0;                                          /* per lane, reduced at the end (a unit holds < 2^18 documents) */
    bool pos_pruned = false;

    auto next_q = [&](uint32_t from) {                             /* next stripe where every term has a container */
        uint32_t x = from;
        for (; x < n_local; ++x) {
            bool all = true;
#pragma unroll
            for (uint32_t t = 0; t < kDenseT; ++t) if (t < T) all = all && rs[t * kDenseSpg + x] != 0u;
            if (all) break;
        }
        return x;
    };
    auto load_and = [&](uint32_t x) {                              /* AND of the terms' bitmaps of local stripe x: this lane's 128 documents */
        /* four loads whatever T is (a query of fewer terms reads its last term's bitmap again: the AND does not change), so that
         * they are issued back to back and waited for once */
        dense_u4 r[kDenseT];
#pragma unroll
        for (uint32_t t = 0; t < kDenseT; ++t) {
            const uint32_t off = rfl32(rs[(t < T ? t : T - 1u) * kDenseSpg + x]);
            const dense_u4* bmp = reinterpret_cast<const dense_u4*>(seg.dense_data + (size_t)off * 16);
            r[t] = bmp[lane * 4u < NW ? lane : 0u];
        }
        dense_u4 a = r[0] & r[1] & r[2] & r[3];
        if (lane * 4u >= NW) a = dense_u4{0u, 0u, 0u, 0u};
        if (TALLY) { cn_bmpw += T * NW; }
        return a;
    };

    /* top-k buffer: room for a round of insertions (bitonic selection when it fills; positional pruning wants its threshold as
     * soon as k matches are held) */
    auto make_room = [&]() {
        if (tkn + 64u > kDenseCap || (PHRASE && !theta_valid && tkn >= k)) {
            dense_topk_sort(tk_w, tk_d, lane);
            tkn = tkn < k ? tkn : k;
            if (tkn == k) { theta_valid = true; theta_w = tk_w[k - 1]; theta_d = tk_d[k - 1]; }
#pragma unroll
            for (uint32_t i = 0; i < kDenseCap; i += 64u) if (i + lane >= tkn) { tk_w[i + lane] = 0; tk_d[i + lane] = 0xFFFFFFFFu; }
            wave_lds_fence();
        }
    };
    auto insert = [&](bool take, uint64_t wb, uint32_t did) {
        const uint64_t tm = __ballot(take);
        if (take) { const uint32_t p = tkn + mbcnt(tm); tk_w[p] = wb; tk_d[p] = did; }
        tkn += (uint32_t)__popcll(tm);
        wave_lds_fence();
    };

    /* PHRASE: candidates that passed the weight test wait here for their positional test, so that the tests run 64 at a time —
     * one in twenty-five candidates of C5 gets this far, i.e. one or two per round: tested where they fall, their serial chain of
     * dependent reads would be paid by nearly every round with one lane busy */
    uint32_t sq_n = 0;
    uint32_t n_tested = 0;                                         /* diagnostics: positional tests made (per lane) */
    auto survivors = [&](uint32_t n) {
        make_room();
        const bool valid = lane < n;
        const uint32_t e = sq_n - n + (valid ? lane : 0u);
        const uint32_t did = sq_d[e], wv4 = sq_v[e];
        const uint64_t wb = sq_w[e];
        sq_n -= n;
        /* the threshold may have moved on since the candidate was queued */
        bool take = valid && (!theta_valid || cand_before(wb, did, theta_w, theta_d)) && !(wb < theta_glob);
        if (valid && !take) pos_pruned = true;
        if (take) {
            ++n_tested;
            const uint32_t x = (did >> SB) - s_begin, slot = did & (W - 1u);
            uint32_t roff[kDenseT];
#pragma unroll
            for (uint32_t t = 0; t < kDenseT; ++t) roff[t] = t < T ? rs[t * kDenseSpg + x] : 0u;
#ifdef XGM_DENSE_SKIP_POS           /* timing experiment only: every positional test fails without being made (wrong results) */
            take = roff[0] == 0xFFFFFFFFu;
#else
            take = dense_positions_ok(seg, q, T, roff[0], roff[1], roff[2], roff[3], slot, wv4 & 0xFFu, (wv4 >> 8) & 0xFFu, (wv4 >> 16) & 0xFFu, wv4 >> 24);
#endif
            if (TALLY) { for (uint32_t t = 0; t < T; ++t) cn_pos += ((wv4 >> (8u * t)) & 0xFFu) - 1u; }
        }
        if (take) ++matches;
        if (hist_g && take) {
            int b = (int)(wb >> 47) - hbase;
            b = b < 0 ? 0 : (b > (int)XGM_OR_HIST - 1 ? (int)XGM_OR_HIST - 1 : b);
            atomicAdd(&hist_g[b], 1u);
        }
        insert(take, wb, did);
    };

    /* One round of 64 candidates, software-pipelined: `issue` takes the next n candidates off the ring and requests their T wdf
     * bytes and document length; `weigh` consumes what the PREVIOUS issue requested — the gathers of round r + 1 are in flight
     * while round r is weighed. */
    uint32_t head = 0, pend = 0;
    uint32_t p_n = 0, p_did = 0, p_dlen = 0;                       /* the round in flight: its size, this lane's candidate, what was loaded */
    uint32_t p_wv[kDenseT] = {1u, 1u, 1u, 1u};
    uint32_t n_rounds = 0;
    auto weigh = [&]() {
        if (p_n == 0u) return;
        if (!PHRASE) make_room();
        /* what the query's other units have learned meanwhile: every other round (the units of a frequent-term phrase start without
         * a threshold of their own; looked at once per stripe, eight times as many positional tests were made) */
        if (PHRASE && hist_g && (n_rounds++ & 1u) == 0u) look_at_histogram();
        const uint32_t n = p_n, did = p_did;
        const bool valid = lane < n;
        /* Once a threshold is in force nearly every candidate falls short of it by far: a single-precision estimate of the weight
         * (relative error < 1e-6: ~12 roundings of 2^-24 per term, positive terms) settles those for a fifth of the instructions of
         * the three IEEE double divisions; whatever comes within 1e-5 of the threshold — and every round before there is one — is
         * weighed exactly.  The estimate never decides a result: it only spares the exact computation where its outcome is known. */
        bool maybe = valid;
        if (theta_valid || (PHRASE && theta_glob)) {
            const float th = (float)__longlong_as_double((long long)(theta_valid ? (PHRASE && theta_glob > theta_w ? theta_glob : theta_w) : theta_glob)) * 0.99999f;
            const float lenf = (float)p_dlen;
            float nl = lenf * f_len_factor;
            nl = nl > f_min_normlen ? nl : f_min_normlen;
            const float dl = f_k1 * (nl * f_b + f_omb);
            float wf = 0.0f;
#pragma unroll
            for (uint32_t t = 0; t < kDenseT; ++t) {
                if (t < T) {
                    const float wdf = (float)(p_wv[t] - 1u);
                    wf += f_tw[t] * (wdf * __builtin_amdgcn_rcpf(dl + wdf));
                }
            }
            maybe = valid && wf >= th;
        }
        uint64_t wb = 0;
        if (__ballot(maybe)) {
            /* BM25Weight::get_sumpart, bm25weight.cc:170-181 — same operations, same order; MultiAndPostList::get_weight: ((0 + w0) + w1) + ... */
            const double len = (double)p_dlen;
            double normlen = len * u_len_factor;
            normlen = normlen > u_min_normlen ? normlen : u_min_normlen;
            const double denom_len = u_k1 * (normlen * u_b + u_omb);
            double weight = 0.0;
#pragma unroll
            for (uint32_t t = 0; t < kDenseT; ++t) {
                if (t < T) {
                    const double wdf = (double)(p_wv[t] - 1u);
                    const double denom = denom_len + wdf;
                    weight = weight + u_tw[t] * (wdf / denom);
                }
            }
            wb = (uint64_t)__double_as_longlong(weight);
        }
        bool take = maybe && (!theta_valid || cand_before(wb, did, theta_w, theta_d));
        if (PHRASE) {
            /* weighed first: a candidate that cannot enter the top k is dropped with its positions untested (the match count
             * becomes a lower bound, XGM_MATCHES_LOWER_BOUND); the others queue up for the positional test */
            if (take && wb < theta_glob) take = false;
            if (valid && !take) pos_pruned = true;
            const uint64_t tm = __ballot(take);
            if (take) {
                const uint32_t e = sq_n + mbcnt(tm);
                sq_d[e] = did; sq_w[e] = wb; sq_v[e] = p_wv[0] | (p_wv[1] << 8) | (p_wv[2] << 16) | (p_wv[3] << 24);
            }
            sq_n += (uint32_t)__popcll(tm);
            wave_lds_fence();
            if (sq_n >= 64u) survivors(64u);
        } else {
            if (valid) ++matches;                                  /* a bit of the AND is a match */
            insert(take, wb, did);
        }
        p_n = 0u;
    };
    auto issue = [&](uint32_t n) {
        const bool valid = lane < n;
        /* lanes beyond n repeat the round's first candidate: every lane loads, no branch around the gathers */
        const uint32_t did = ring[(head + (valid ? lane : 0u)) & (kDenseRing - 1u)];
        head = (head + n) & (kDenseRing - 1u);
        pend -= n;
        const uint32_t x = (did >> SB) - s_begin, slot = did & (W - 1u);
        uint32_t wv[kDenseT];
#pragma unroll
        for (uint32_t t = 0; t < kDenseT; ++t)
            wv[t] = seg.dense_data[(size_t)rs[(t < T ? t : T - 1u) * kDenseSpg + x] * 16 + (size_t)NW * 4 + slot];
        const uint32_t dlen = seg.doclen[did];
        if (TALLY) {
            const uint32_t sec = tally_sectors(valid, did, 6u);
            cn_probe += T * sec; cn_probe_raw += T * n;
            cn_dl += tally_sectors(valid, did, 4u); cn_dl_raw += n;
        }
        weigh();                                                   /* the previous round, while these loads are in flight */
        p_n = n; p_did = did; p_dlen = dlen;
#pragma unroll
        for (uint32_t t = 0; t < kDenseT; ++t) p_wv[t] = wv[t];
    };

    /* producer state: the stripe being enumerated.  Candidates enter the ring in DOCID order (lane l owns documents 128 l .. 128 l + 127
     * of the stripe; a wave prefix sum of the lanes' bit counts gives every lane its first ordinal): a round of 64 consecutive
     * candidates then touches few memory sectors — for the dense matches of frequent terms 64 candidates span ~200 documents, 4
     * sectors of wdf bytes per term, not 64.  (The first version took two bits per lane and pass: simpler, and 2x slower on C5 for
     * exactly that reason.)  The next stripe's bitmaps are requested as soon as this one's have arrived. */
    uint32_t sl = next_q(0);
    dense_u4 m = dense_u4{0u, 0u, 0u, 0u}, m_next = dense_u4{0u, 0u, 0u, 0u};
    uint32_t sl_next = n_local;
    if (sl < n_local) {
        m = load_and(sl);
        sl_next = next_q(sl + 1u);
        if (sl_next < n_local) m_next = load_and(sl_next);
    }
    uint32_t o = 0u, n_total = 0u, done = 0u, tail = 0u;
    bool fresh = true;                                             /* m holds a stripe whose bits have not been counted yet */
    while (sl < n_local || pend) {
        if (sl < n_local && pend + 64u <= kDenseRing) {
            if (fresh) {
                const uint32_t cnt = (uint32_t)(__popc(m.x) + __popc(m.y) + __popc(m.z) + __popc(m.w));
                const uint32_t incl = wave_incl_scan(cnt);
                n_total = rl32(incl, 63);
                o = incl - cnt;                                    /* this lane's next ordinal in the stripe */
                done = 0u;
                fresh = false;
            }
            const uint32_t room = kDenseRing - pend;
            const uint32_t take_n = n_total - done < room ? n_total - done : room;
            const uint32_t lim = done + take_n;
            const uint32_t wbase = tail - done;                    /* ring position of ordinal 0 (mod the ring) */
            const uint32_t dbase = ((s_begin + sl) << SB) + lane * 128u;
#define XGM_DENSE_EMIT(MW, I)                                                                             \
            while (MW && o < lim) {                                                                       \
                const uint32_t bit = (uint32_t)__ffs(MW) - 1u;                                            \
                ring[(wbase + o) & (kDenseRing - 1u)] = dbase + (I) * 32u + bit;                          \
                MW &= MW - 1u;                                                                            \
                ++o;                                                                                      \
            }
            XGM_DENSE_EMIT(m.x, 0u)
            XGM_DENSE_EMIT(m.y, 1u)
            XGM_DENSE_EMIT(m.z, 2u)
            XGM_DENSE_EMIT(m.w, 3u)
#undef XGM_DENSE_EMIT
            tail = (tail + take_n) & (kDenseRing - 1u);
            pend += take_n;
            if (TALLY) { q_cands += take_n; }
            done = lim;
            if (done == n_total) {
                sl = sl_next;
                m = m_next;
                if (sl < n_local) {
                    sl_next = next_q(sl + 1u);
                    if (sl_next < n_local) m_next = load_and(sl_next);
                }
                fresh = true;
            }
            wave_lds_fence();
        }
        if (pend >= 64u || (sl >= n_local && pend)) issue(pend < 64u ? pend : 64u);
    }
    weigh();                                                       /* the last round in flight */
    if (PHRASE) { while (sq_n) survivors(sq_n < 64u ? sq_n : 64u); }

    /* ---- unit epilogue ---- */
    dense_topk_sort(tk_w, tk_d, lane);
    for (int sh = 32; sh > 0; sh >>= 1) matches += (uint32_t)__shfl_xor((int)matches, sh);
    if (PHRASE) { for (int sh = 32; sh > 0; sh >>= 1) n_tested += (uint32_t)__shfl_xor((int)n_tested, sh); }
    if (PHRASE && TALLY) { for (int sh = 32; sh > 0; sh >>= 1) cn_pos += (unsigned long long)__shfl_xor((long long)cn_pos, sh); }
    const bool any_pruned = PHRASE && __ballot(pos_pruned) != 0ull;
    const uint32_t n_out = tkn < k ? tkn : k;
    xgm_cand* out = cand_out + (size_t)wk.slot * k_stride;
    for (uint32_t i = lane; i < n_out; i += 64u) {
        xgm_cand c;
        c.wbits = tk_w[i]; c.did = tk_d[i]; c.subqs = (uint32_t)__popc(q.score_mask);      /* the weighted leaves all match */
        out[i] = c;
    }
    if (lane == 0) {
        xgm_group_hdr h;
        h.matches = (unsigned long long)matches | (any_pruned ? XGM_MATCHES_LOWER_BOUND : 0ull); h.n_cand = n_out; h.pad = TALLY ? q_cands : (PHRASE ? n_tested : 0u);
        h.t_start = t_unit_start; h.t_end = __builtin_readcyclecounter();
        h.c_pos = cn_pos; h.c_bmp_words = cn_bmpw; h.c_probes = cn_probe; h.c_blk_words = 0; h.c_hdrs = 0;
        h.c_doclen = cn_dl; h.c_aux_words = cn_aux; h.c_probes_raw = cn_probe_raw; h.c_doclen_raw = cn_dl_raw; h.c_pad[0] = 0; h.c_pad[1] = 0;
        ghdr_out[wk.slot] = h;
    }
}

}  // namespace

int xgm_launch_error(const char* what, int code, const char* msg);

size_t xgm_dense_smem_bytes(bool phrase) { return XGM_WAVES * dense_wave_bytes(phrase); }
uint32_t xgm_dense_max_terms() { return kDenseT; }
uint32_t xgm_dense_max_k() { return kDenseCap - 64u; }
uint32_t xgm_dense_max_stripes() { return kDenseSpg; }

template <bool PHRASE, bool TALLY>
static int launch_dense_inst(const xgm_match_launch& L, hipStream_t stream) {
    const dim3 grid((L.n_work + XGM_WAVES - 1) / XGM_WAVES), block(XGM_WG);
    hipLaunchKernelGGL((xgm_dense_kernel<PHRASE, TALLY>), grid, block, xgm_dense_smem_bytes(PHRASE), stream, L.seg, L.queries, L.work, L.n_work, L.k_stride,
                       L.cand, L.ghdr, L.hist);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xgm_launch_error("xgm_dense_kernel", (int)e, hipGetErrorString(e));
    return 0;
}

/* every query of the batch: a conjunction (or, L.phrase, a positional query that prunes by weight) of <= xgm_dense_max_terms() terms
 * that all have containers, k <= xgm_dense_max_k(); units of <= xgm_dense_max_stripes() stripes */
int xgm_launch_dense(const xgm_match_launch& L, hipStream_t stream) {
    if (L.phrase) return L.tally ? launch_dense_inst<true, true>(L, stream) : launch_dense_inst<true, false>(L, stream);
    return L.tally ? launch_dense_inst<false, true>(L, stream) : launch_dense_inst<false, false>(L, stream);
}
