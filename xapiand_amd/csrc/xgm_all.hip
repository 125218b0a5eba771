/* xgm_search_all's second half: the match list the workgroup kernel appended in no particular order (xgm_match_body.inc, all_keys /
 * all_vals) put into ascending docid order and packed as xgm_hit records.
 *
 * The order is what the reference's matcher loop walks (Matcher::get_local_mset, matcher.cc:482-536: the posting-list tree hands
 * out documents by ascending docid), which is what the matcher hook replays for known_matching_docs, the frozen weight of positional
 * queries and the snapshot's collapser (integration/xgm_matcher_hook.cc).
 *
 * Docids are unique and bounded by the shard's lastdocid, so the order is a RANK, not a comparison sort: set bit `did` of a bitmap,
 * prefix-count the bitmap's words, and an entry's place is the number of set bits below its own — three streaming passes over
 * lastdocid / 8 bytes and one scatter of the n entries (HBM-bound integer work: 1.25 MB of bitmap for a 10 M-document shard, 16 B per
 * match).  Plain kernels of our own: the CPU emulation (tests/emu) runs them as they are. */
#include <hip/hip_runtime.h>

#include "xgm_launch.h"

namespace {

constexpr uint32_t kScanThreads = 256u, kWordsPerThread = 16u, kWordsPerBlock = kScanThreads * kWordsPerThread;

__global__ __launch_bounds__(256) void xgm_all_mark_kernel(const unsigned long long* __restrict__ keys, size_t n, uint32_t* __restrict__ bitmap) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t did = (uint32_t)(keys[i] >> 32);
    atomicOr(&bitmap[did >> 5], 1u << (did & 31u));
}

/* per block of 4 096 words: wprefix[w] = set bits in the block's words before w; block_sum[b] = the block's set bits */
__global__ __launch_bounds__(256) void xgm_all_scan_kernel(const uint32_t* __restrict__ bitmap, uint32_t n_words, uint32_t* __restrict__ wprefix,
                                                           uint32_t* __restrict__ block_sum) {
    __shared__ uint32_t part[kScanThreads];
    const uint32_t tid = threadIdx.x;
    const uint32_t w0 = blockIdx.x * kWordsPerBlock + tid * kWordsPerThread;
    uint32_t cnt[kWordsPerThread];
    uint32_t total = 0;
#pragma unroll
    for (uint32_t j = 0; j < kWordsPerThread; ++j) {
        const uint32_t w = w0 + j;
        cnt[j] = w < n_words ? (uint32_t)__popc(bitmap[w]) : 0u;
        total += cnt[j];
    }
    part[tid] = total;
    __syncthreads();
    for (uint32_t d = 1; d < kScanThreads; d <<= 1) {               /* inclusive scan of the threads' totals */
        const uint32_t v = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - total;
#pragma unroll
    for (uint32_t j = 0; j < kWordsPerThread; ++j) {
        const uint32_t w = w0 + j;
        if (w < n_words) wprefix[w] = run;
        run += cnt[j];
    }
    if (tid == kScanThreads - 1u) block_sum[blockIdx.x] = part[tid];
}

/* exclusive scan of the block sums in place (one workgroup; a 10 M-document shard has 77 blocks) */
__global__ __launch_bounds__(256) void xgm_all_block_offsets_kernel(uint32_t* __restrict__ block_sum, uint32_t n_blocks) {
    __shared__ uint32_t part[kScanThreads];
    __shared__ uint32_t carry;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += kScanThreads) {
        const uint32_t b = b0 + tid;
        const uint32_t mine = b < n_blocks ? block_sum[b] : 0u;
        part[tid] = mine;
        __syncthreads();
        for (uint32_t d = 1; d < kScanThreads; d <<= 1) {
            const uint32_t v = tid >= d ? part[tid - d] : 0u;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        const uint32_t base = carry;
        if (b < n_blocks) block_sum[b] = base + part[tid] - mine;
        __syncthreads();
        if (tid == kScanThreads - 1u) carry = base + part[tid];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void xgm_all_place_kernel(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ vals, size_t n,
                                                            const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ wprefix,
                                                            const uint32_t* __restrict__ block_off, xgm_hit* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    const uint32_t did = (uint32_t)(k >> 32), w = did >> 5;
    const uint32_t rank = block_off[w / kWordsPerBlock] + wprefix[w] + (uint32_t)__popc(bitmap[w] & ((1u << (did & 31u)) - 1u));
    xgm_hit h;
    h.docid = did;
    h.subqs_matched = (uint32_t)k;
    h.weight = __longlong_as_double((long long)vals[i]);
    if (rank < n) out[rank] = h;                     /* (always: docids are unique; a duplicate would leave a hole, never write out of bounds) */
}

}  // namespace

/* device bytes of the ordering's own storage for a shard whose largest docid is lastdocid: bitmap + word prefixes + block sums */
size_t xgm_all_order_bytes(uint32_t lastdocid) {
    const size_t n_words = ((size_t)lastdocid >> 5) + 1u, n_blocks = (n_words + kWordsPerBlock - 1u) / kWordsPerBlock;
    return ((n_words * 4 + 255) & ~(size_t)255) * 2 + ((n_blocks * 4 + 255) & ~(size_t)255);
}

int xgm_all_order_pack(void* tmp, uint32_t lastdocid, const unsigned long long* keys, const unsigned long long* vals, size_t n, xgm_hit* out,
                       hipStream_t stream) {
    if (n == 0) return 0;
    const size_t n_words = ((size_t)lastdocid >> 5) + 1u, n_blocks = (n_words + kWordsPerBlock - 1u) / kWordsPerBlock;
    const size_t b_words = (n_words * 4 + 255) & ~(size_t)255;
    uint32_t* bitmap = (uint32_t*)tmp;
    uint32_t* wprefix = (uint32_t*)((unsigned char*)tmp + b_words);
    uint32_t* block_sum = (uint32_t*)((unsigned char*)tmp + 2 * b_words);
    hipError_t e = hipMemsetAsync(bitmap, 0, n_words * 4, stream);
    if (e != hipSuccess) return xgm_launch_error("hipMemsetAsync(match bitmap)", (int)e, hipGetErrorString(e));
    const dim3 per_entry((unsigned)((n + 255u) / 256u)), wg(256);
    hipLaunchKernelGGL(xgm_all_mark_kernel, per_entry, wg, 0, stream, keys, n, bitmap);
    hipLaunchKernelGGL(xgm_all_scan_kernel, dim3((unsigned)n_blocks), wg, 0, stream, bitmap, (uint32_t)n_words, wprefix, block_sum);
    hipLaunchKernelGGL(xgm_all_block_offsets_kernel, dim3(1), wg, 0, stream, block_sum, (uint32_t)n_blocks);
    hipLaunchKernelGGL(xgm_all_place_kernel, per_entry, wg, 0, stream, keys, vals, n, bitmap, wprefix, block_sum, out);
    e = hipGetLastError();
    if (e != hipSuccess) return xgm_launch_error("xgm_search_all ordering kernels", (int)e, hipGetErrorString(e));
    return 0;
}
