/* xgm_search_all's second half: the match list the workgroup kernel appended in no particular order (xgm_match_body.inc, all_keys /
 * all_vals) put into ascending docid order and packed as xgm_hit records.
 *
 * The order is what the reference's matcher loop walks (Matcher::get_local_mset, matcher.cc:482-536: the posting-list tree hands
 * out documents by ascending docid), which is what the matcher hook replays for known_matching_docs, the frozen weight of positional
 * queries and the snapshot's collapser (integration/xgm_matcher_hook.cc).  A radix sort over the docid half of the key (rocPRIM's
 * device-wide sort: 4 passes of 8 bits) — HBM-bound streaming, ~1 ms for the 5 M matches of a disjunction over a 10 M-document
 * shard; no MFMA, no LDS tricks of our own. */
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "xgm_launch.h"

namespace {

__global__ __launch_bounds__(256) void xgm_all_pack_kernel(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ vals,
                                                           size_t n, xgm_hit* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    xgm_hit h;
    h.docid = (uint32_t)(k >> 32);
    h.subqs_matched = (uint32_t)k;
    h.weight = __longlong_as_double((long long)vals[i]);
    out[i] = h;
}

}  // namespace

size_t xgm_all_sort_temp_bytes(size_t n) {
    size_t bytes = 0;
    rocprim::double_buffer<unsigned long long> k(nullptr, nullptr), v(nullptr, nullptr);
    if (rocprim::radix_sort_pairs(nullptr, bytes, k, v, n ? n : 1, 32, 64, nullptr) != hipSuccess) return 0;
    return bytes ? bytes : 16;
}

int xgm_all_sort_pack(void* tmp, size_t tmp_bytes, unsigned long long* keys, unsigned long long* keys_alt, unsigned long long* vals,
                      unsigned long long* vals_alt, size_t n, xgm_hit* out, hipStream_t stream) {
    if (n == 0) return 0;
    rocprim::double_buffer<unsigned long long> k(keys, keys_alt), v(vals, vals_alt);
    hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, k, v, n, 32, 64, stream);
    if (e != hipSuccess) return xgm_launch_error("radix_sort_pairs", (int)e, hipGetErrorString(e));
    hipLaunchKernelGGL(xgm_all_pack_kernel, dim3((unsigned)((n + 255u) / 256u)), dim3(256), 0, stream, k.current(), v.current(), n, out);
    e = hipGetLastError();
    if (e != hipSuccess) return xgm_launch_error("xgm_all_pack_kernel", (int)e, hipGetErrorString(e));
    return 0;
}
