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

#include "xgm_dense_body.inc"

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
    xgm_dense_unit<PHRASE, TALLY>(seg, queries, work[unit], smem + (size_t)wave * dense_wave_bytes(PHRASE), lane, k_stride, cand_out, ghdr_out, hist_all);
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
