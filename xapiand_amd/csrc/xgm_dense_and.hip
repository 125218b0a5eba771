/* xgm_dense_kernel — xgm_dense_unit (xgm_dense_body.inc: the conjunction whose every term has probe containers) as a kernel of its
 * own, plain and positional.  AN EXPERIMENT KEPT FOR A/B RUNS, OFF BY DEFAULT (XGM_DENSE_KERNEL=1 switches it on; run_batch then cuts
 * such queries into a class of their own).  The product runs the same body INSIDE xgm_andw_kernel, one launch for the whole batch.
 *
 * What it showed on the MI355X (round 3): at 6 waves per SIMD without spills C2's all-container class takes 266 us
 * against ~303 us through xgm_andw_kernel's queue path — but a batch then needs two match + two merge launches and loses more than
 * that; the positional instantiation (survivors' positions tested 64 at a time by the serial predicates of xgm_posfilter.h) halves
 * the instructions of xgm_andw_kernel<..., true, ...> and is still slower (2.96 vs 2.59 ms on its class): that path is bound by the
 * latency chain of a wave's round, not by issue slots.  Same inputs (work units of plan_batch), same outputs (xgm_cand /
 * xgm_group_hdr for xgm_merge_kernel); parity-green at 10 M documents (tests/test_gpu_variants.py runs the suite with it on).
 * Compile with -ffp-contract=off.
 */
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>

#include "xgm_device.h"
#include "xgm_launch.h"
#include "xgm_wave.h"
#include "xgm_posfilter.h"
#include "xgm_unit_finish.h"

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
                                                              uint32_t* __restrict__ hist_all, const xgm_fuse* __restrict__ fuse) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t unit = blockIdx.x * XGM_WAVES + wave;
    if (unit >= n_work) return;                                    /* no barriers below: early exit is safe */
    xgm_dense_unit<PHRASE, TALLY>(seg, queries, work[unit], smem + (size_t)wave * dense_wave_bytes(PHRASE), lane, k_stride, cand_out, ghdr_out, hist_all, fuse);
}

}  // namespace

int xgm_launch_error(const char* what, int code, const char* msg);

size_t xgm_dense_smem_bytes(bool phrase) { return XGM_WAVES * dense_wave_bytes(phrase); }
uint32_t xgm_dense_max_terms() { return kDenseT; }
uint32_t xgm_dense_max_k() { return kDenseCap - 64u; }
uint32_t xgm_dense_max_stripes() { return kDenseSpg; }
bool xgm_dense_word_major() { return XGM_DENSE_WORD_MAJOR != 0; }     /* (the positional body's bitmaps: a component of a lane's words = a quarter of the stripe) */

template <bool PHRASE, bool TALLY>
static int launch_dense_inst(const xgm_match_launch& L, hipStream_t stream) {
    const dim3 grid((L.n_work + XGM_WAVES - 1) / XGM_WAVES), block(XGM_WG);
    hipLaunchKernelGGL((xgm_dense_kernel<PHRASE, TALLY>), grid, block, xgm_dense_smem_bytes(PHRASE), stream, L.seg, L.queries, L.work, L.n_work, L.k_stride,
                       L.cand, L.ghdr, L.hist, L.fuse);
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
