/* GPU segment builder for the synthetic corpus of tools/xgm_corpus.h (bench / test tooling).
 *
 * Index-build side, not the query path: it exists because the benchmark corpus (10 M documents,
 * ~10^9 tokens per shard) cannot be generated, inverted and block-encoded on the host in bench
 * time.  It must produce, for the same documents, sections bit-identical to the host builder
 * (xgm_segment_build.cc) — tests/test_gpu_builder.py checks that.
 *
 * Pipeline (all device-side, rocPRIM/hipCUB for the sort and the scans):
 *   doc lengths → token offsets → one 64-bit key per token (term sort-id | local docid | position)
 *   → radix sort → posting boundaries (wdf = run length, positions = low byte, already ordered)
 *   → (term, stripe) runs → blocks of <= 128 postings → per-block bit widths → word offsets
 *   → bit-packing → dictionary compaction (terms with df > 0).
 */
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstring>
#include <numeric>
#include <string>

#include "../../tools/xgm_corpus.h"
#include "xgm_internal.h"
#include "xgm_launch.h"

namespace {

#define SY_TRY(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) { rc = xgm_launch_error(#expr, (int)e_, hipGetErrorString(e_)); goto fail; } \
    } while (0)

constexpr int TB = 256;
inline unsigned grid_for(uint64_t n) { return (unsigned)std::min<uint64_t>((n + TB - 1) / TB, 1u << 20); }

/* key layout: [sort id : 24+][local docid : 32][position : 8] */
__device__ __forceinline__ uint32_t key_sid(uint64_t k) { return (uint32_t)(k >> 40); }
__device__ __forceinline__ uint32_t key_doc(uint64_t k) { return (uint32_t)(k >> 8); }
__device__ __forceinline__ uint64_t key_posting(uint64_t k) { return k >> 8; }

__global__ void k_doclen(xgm_corpus_params cp, uint32_t n_local, uint32_t n_shards, uint32_t shard, uint32_t* doclen /*[n_local+1]*/,
                         uint64_t* len64 /*[n_local]*/) {
    for (uint64_t i = blockIdx.x * (uint64_t)TB + threadIdx.x; i < n_local; i += (uint64_t)gridDim.x * TB) {
        uint64_t g = (uint64_t)i * n_shards + shard + 1;
        uint32_t l = xgm_doc_len(&cp, g);
        doclen[i + 1] = l;
        len64[i] = l;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) doclen[0] = 0;
}

__global__ void k_tokens(xgm_corpus_params cp, const uint64_t* __restrict__ thr, const uint32_t* __restrict__ rank2sid, uint32_t n_local,
                         uint32_t n_shards, uint32_t shard, const uint32_t* __restrict__ doclen, const uint64_t* __restrict__ tok_start,
                         uint64_t* __restrict__ keys) {
    const uint32_t slots = cp.len_hi;
    const uint64_t total = (uint64_t)n_local * slots;
    for (uint64_t x = blockIdx.x * (uint64_t)TB + threadIdx.x; x < total; x += (uint64_t)gridDim.x * TB) {
        uint32_t i = (uint32_t)(x / slots), pos = (uint32_t)(x % slots) + 1;
        if (pos > doclen[i + 1]) continue;
        uint64_t g = (uint64_t)i * n_shards + shard + 1;
        uint32_t rank = xgm_token(&cp, thr, g, pos);
        keys[tok_start[i] + pos - 1] = ((uint64_t)rank2sid[rank - 1] << 40) | ((uint64_t)(i + 1) << 8) | pos;
    }
}

__global__ void k_posting_flags(const uint64_t* __restrict__ keys, uint64_t n, uint32_t* __restrict__ flag) {
    for (uint64_t i = blockIdx.x * (uint64_t)TB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TB)
        flag[i] = (i == 0 || key_posting(keys[i]) != key_posting(keys[i - 1])) ? 1u : 0u;
}

/* per posting: (sid, doc) and the index of its first token */
__global__ void k_postings(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pidx, uint64_t n,
                           uint32_t* __restrict__ p_sid, uint32_t* __restrict__ p_doc, uint64_t* __restrict__ p_tok, uint16_t* __restrict__ positions) {
    for (uint64_t i = blockIdx.x * (uint64_t)TB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TB) {
        uint64_t k = keys[i];
        if (positions) positions[i] = (uint16_t)(k & 0xFFu);       /* synthetic documents are < 256 tokens: every term is XGM_TF_POS16 */
        if (flag[i]) {
            uint32_t p = pidx[i];
            p_sid[p] = key_sid(k);
            p_doc[p] = key_doc(k);
            p_tok[p] = i;
        }
    }
}

__global__ void k_wdf(const uint64_t* __restrict__ p_tok, uint64_t n_post, uint64_t n_tok, uint32_t* __restrict__ wdf) {
    for (uint64_t p = blockIdx.x * (uint64_t)TB + threadIdx.x; p < n_post; p += (uint64_t)gridDim.x * TB)
        wdf[p] = (uint32_t)((p + 1 < n_post ? p_tok[p + 1] : n_tok) - p_tok[p]);
}

/* term_pstart[s] = first posting of sort-id s (n_post for ids after the last), s in [0, V] */
__global__ void k_term_starts(const uint32_t* __restrict__ p_sid, uint64_t n_post, uint32_t V, uint64_t* __restrict__ term_pstart) {
    for (uint64_t p = blockIdx.x * (uint64_t)TB + threadIdx.x; p <= n_post; p += (uint64_t)gridDim.x * TB) {
        uint32_t lo = p == 0 ? 0u : p_sid[p - 1] + 1u;
        uint32_t hi = p == n_post ? V : p_sid[p];            /* inclusive */
        if (p == 0 || p == n_post || p_sid[p] != p_sid[p - 1])
            for (uint32_t s = lo; s <= hi && s <= V; ++s) term_pstart[s] = p;
    }
}

/* run start marker: posting index where the (sid, stripe) run begins, else 0 (max-scanned later) */
__global__ void k_run_marks(const uint32_t* __restrict__ p_sid, const uint32_t* __restrict__ p_doc, uint64_t n_post, uint32_t sb,
                            uint64_t* __restrict__ mark) {
    for (uint64_t p = blockIdx.x * (uint64_t)TB + threadIdx.x; p < n_post; p += (uint64_t)gridDim.x * TB) {
        bool start = p == 0 || p_sid[p] != p_sid[p - 1] || (p_doc[p] >> sb) != (p_doc[p - 1] >> sb);
        mark[p] = start ? p : 0;
    }
}

__global__ void k_block_flags(const uint64_t* __restrict__ run_start, uint64_t n_post, uint32_t* __restrict__ bflag) {
    for (uint64_t p = blockIdx.x * (uint64_t)TB + threadIdx.x; p < n_post; p += (uint64_t)gridDim.x * TB)
        bflag[p] = ((p - run_start[p]) % XGM_BLOCK == 0) ? 1u : 0u;
}

__global__ void k_block_first_posting(const uint32_t* __restrict__ bflag, const uint32_t* __restrict__ bidx, uint64_t n_post,
                                      uint64_t* __restrict__ blk_p) {
    for (uint64_t p = blockIdx.x * (uint64_t)TB + threadIdx.x; p < n_post; p += (uint64_t)gridDim.x * TB)
        if (bflag[p]) blk_p[bidx[p]] = p;
}

struct MaxOp {
    __device__ __forceinline__ uint64_t operator()(uint64_t a, uint64_t b) const { return a > b ? a : b; }
};

__device__ __forceinline__ uint32_t bits_needed(uint32_t v) { return v ? 32u - (uint32_t)__clz(v) : 0u; }

/* one wave per block: header + word count */
__global__ void k_block_meta(const uint64_t* __restrict__ blk_p, uint64_t n_blk, uint64_t n_post, const uint32_t* __restrict__ p_doc,
                             const uint32_t* __restrict__ wdf, uint32_t* __restrict__ blk_first, uint32_t* __restrict__ blk_meta,
                             uint64_t* __restrict__ blk_nwords) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t b = (blockIdx.x * (uint64_t)TB + threadIdx.x) >> 6; b < n_blk; b += ((uint64_t)gridDim.x * TB) >> 6) {
        uint64_t p0 = blk_p[b];
        uint32_t n = (uint32_t)((b + 1 < n_blk ? blk_p[b + 1] : n_post) - p0);
        uint32_t mg = 0, mw = 0;
        for (uint32_t j = lane; j < n; j += 64u) {
            if (j) mg = max(mg, p_doc[p0 + j] - p_doc[p0 + j - 1] - 1u);
            mw = max(mw, wdf[p0 + j]);
        }
        for (int o = 32; o > 0; o >>= 1) { mg = max(mg, (uint32_t)__shfl_xor((int)mg, o)); mw = max(mw, (uint32_t)__shfl_xor((int)mw, o)); }
        if (lane == 0) {
            uint32_t bwg = bits_needed(mg), bww = bits_needed(mw);
            blk_first[b] = p_doc[p0];
            blk_meta[b] = XGM_META(n, bwg, bww);
            blk_nwords[b] = (n * bwg + 31u) / 32u + (n * bww + 31u) / 32u;
        }
    }
}

/* one wave per block: pack gaps and wdfs through LDS */
__global__ __launch_bounds__(TB) void k_block_pack(const uint64_t* __restrict__ blk_p, uint64_t n_blk, uint64_t n_post,
                                                   const uint32_t* __restrict__ p_doc, const uint32_t* __restrict__ wdf,
                                                   const uint32_t* __restrict__ blk_meta, const uint64_t* __restrict__ blk_goff,
                                                   uint32_t* __restrict__ words) {
    __shared__ uint32_t lds[TB / 64][260];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* buf = lds[wave];
    for (uint64_t b = (blockIdx.x * (uint64_t)TB + threadIdx.x) >> 6; b < n_blk; b += ((uint64_t)gridDim.x * TB) >> 6) {
        const uint64_t p0 = blk_p[b];
        const uint32_t meta = blk_meta[b];
        const uint32_t n = XGM_META_COUNT(meta), bwg = XGM_META_BWG(meta), bww = XGM_META_BWW(meta);
        const uint32_t ngw = (n * bwg + 31u) / 32u, nww = (n * bww + 31u) / 32u;
        for (uint32_t w = lane; w < ngw + nww + 1u; w += 64u) buf[w] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t j = lane; j < n; j += 64u) {
            if (j && bwg) {
                uint32_t v = p_doc[p0 + j] - p_doc[p0 + j - 1] - 1u;
                uint32_t bit = j * bwg, w = bit >> 5, sh = bit & 31u;
                atomicOr(&buf[w], v << sh);
                if (sh + bwg > 32u) atomicOr(&buf[w + 1], v >> (32u - sh));
            }
            if (bww) {
                uint32_t v = wdf[p0 + j];
                uint32_t bit = j * bww, w = ngw + (bit >> 5), sh = bit & 31u;
                atomicOr(&buf[w], v << sh);
                if (sh + bww > 32u) atomicOr(&buf[w + 1], v >> (32u - sh));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t* out = words + blk_goff[b];
        for (uint32_t w = lane; w < ngw + nww; w += 64u) out[w] = buf[w];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

/* dictionary: compact sort-ids with df > 0 */
__global__ void k_term_flags(const uint64_t* __restrict__ term_pstart, uint32_t V, uint32_t* __restrict__ has) {
    for (uint32_t s = blockIdx.x * TB + threadIdx.x; s < V; s += gridDim.x * TB) has[s] = term_pstart[s + 1] > term_pstart[s] ? 1u : 0u;
}

__global__ void k_term_tables(const uint64_t* __restrict__ term_pstart, const uint32_t* __restrict__ has, const uint32_t* __restrict__ newid,
                              uint32_t V, uint32_t T, uint64_t n_post, uint64_t n_tok, uint64_t n_blk, uint64_t n_words,
                              const uint64_t* __restrict__ p_tok, const uint32_t* __restrict__ wdf, const uint32_t* __restrict__ pblk /*block of posting*/,
                              const uint64_t* __restrict__ blk_goff, uint32_t wdf_ub_db, int with_pos,
                              uint32_t* __restrict__ t_df, uint32_t* __restrict__ t_cf, uint32_t* __restrict__ t_wdfub, uint32_t* __restrict__ t_flags,
                              uint64_t* __restrict__ t_blk, uint64_t* __restrict__ t_word, uint64_t* __restrict__ t_pos) {
    for (uint32_t s = blockIdx.x * TB + threadIdx.x; s < V; s += gridDim.x * TB) {
        if (!has[s]) continue;
        uint32_t t = newid[s];
        uint64_t a = term_pstart[s], b = term_pstart[s + 1];
        uint64_t ta = p_tok[a], tb = b < n_post ? p_tok[b] : n_tok;
        uint64_t cf = tb - ta;
        uint32_t df = (uint32_t)(b - a);
        uint32_t cf32 = cf > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cf;
        uint32_t first_wdf = wdf[a];
        uint32_t ub = (cf32 == 0 || df == 1) ? cf32 : max(cf32 - first_wdf, first_wdf);
        t_df[t] = df;
        t_cf[t] = cf32;
        t_wdfub[t] = min(ub, wdf_ub_db);
        t_flags[t] = with_pos ? (XGM_TF_POS_OK | XGM_TF_POS16) : 0u;
        uint64_t blk = pblk[a];
        t_blk[t] = blk;
        t_word[t] = blk_goff[blk];
        t_pos[t] = with_pos ? ta * 2u : 0;                    /* byte offset of the term's u16 position array */
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        t_blk[T] = n_blk;
        t_word[T] = n_words;
        t_pos[T] = with_pos ? n_tok * 2u : 0;
    }
}

/* relative block offsets + remap */
__global__ void k_block_relative(const uint64_t* __restrict__ blk_p, uint64_t n_blk, const uint32_t* __restrict__ p_sid,
                                 const uint32_t* __restrict__ newid, const uint64_t* __restrict__ t_word, const uint64_t* __restrict__ t_pos,
                                 const uint64_t* __restrict__ blk_goff, const uint64_t* __restrict__ p_tok, int with_pos,
                                 uint32_t* __restrict__ blk_word, uint32_t* __restrict__ blk_pos, uint32_t* __restrict__ overflow) {
    for (uint64_t b = blockIdx.x * (uint64_t)TB + threadIdx.x; b < n_blk; b += (uint64_t)gridDim.x * TB) {
        uint64_t p0 = blk_p[b];
        uint32_t t = newid[p_sid[p0]];
        uint64_t wo = blk_goff[b] - t_word[t];
        uint64_t po = with_pos ? p_tok[p0] - t_pos[t] / 2u : 0;    /* entries */
        if (wo > 0xFFFFFFFFull || po > 0xFFFFFFFFull) *overflow = 1;
        blk_word[b] = (uint32_t)wo;
        blk_pos[b] = (uint32_t)po;
    }
}

struct DevBuf {
    void* p = nullptr;
    template <class T> T* as() { return (T*)p; }
    ~DevBuf() { if (p) hipFree(p); }
    hipError_t alloc(size_t bytes) { if (p) hipFree(p); p = nullptr; return hipMalloc(&p, bytes ? bytes : 8); }
    void* release() { void* r = p; p = nullptr; return r; }
    void reset() { if (p) hipFree(p); p = nullptr; }
};

template <class T>
hipError_t excl_sum(const T* in, T* out, uint64_t n, DevBuf& tmp) {
    size_t bytes = 0;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, n);
    if (e != hipSuccess) return e;
    if ((e = tmp.alloc(bytes)) != hipSuccess) return e;
    return hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, in, out, n);
}

}  // namespace

extern "C" int xgm_index_build_synthetic(const xgm_synth_params* sp, int device, xgm_index** out) {
    if (!sp || !out) return xgm_set_error(XGM_E_INVALID, "null argument");
    *out = nullptr;
    if (sp->vocab == 0 || sp->vocab >= (1u << 24) || sp->len_lo == 0 || sp->len_hi < sp->len_lo || sp->len_hi > 255 ||
        sp->n_shards == 0 || sp->shard >= sp->n_shards || sp->n_docs_global == 0)
        return xgm_set_error(XGM_E_INVALID, "bad synthetic corpus parameters");
    uint32_t sb = sp->stripe_bits ? sp->stripe_bits : XGM_DEFAULT_STRIPE_BITS;
    if (sb < XGM_MIN_STRIPE_BITS || sb > XGM_MAX_STRIPE_BITS) return xgm_set_error(XGM_E_INVALID, "stripe_bits out of range");
    const uint64_t n_local64 = (sp->n_docs_global - sp->shard + sp->n_shards - 1) / sp->n_shards;
    if (n_local64 == 0 || n_local64 >= 0xFFFFFFFFull) return xgm_set_error(XGM_E_INVALID, "shard has no documents or too many");
    const uint32_t n_local = (uint32_t)n_local64;
    const uint32_t V = sp->vocab;
    const int with_pos = sp->with_positions ? 1 : 0;
    {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) return xgm_set_error(XGM_E_NO_DEVICE, "no HIP device available (%s)", hipGetErrorString(e));
        if (device < 0 || device >= n) return xgm_set_error(XGM_E_NO_DEVICE, "HIP device %d out of range", device);
        e = hipSetDevice(device);
        if (e != hipSuccess) return xgm_launch_error("hipSetDevice", (int)e, hipGetErrorString(e));
    }

    /* host: Zipf thresholds and the bytewise order of the names "t<rank>" */
    xgm_corpus_params cp{sp->seed, sp->vocab, sp->len_lo, sp->len_hi};
    std::vector<uint64_t> thr(V);
    xgm_zipf_thresholds(V, thr.data());
    std::vector<std::string> names(V);
    for (uint32_t r = 0; r < V; ++r) names[r] = "t" + std::to_string(r + 1);
    std::vector<uint32_t> by_name(V);
    std::iota(by_name.begin(), by_name.end(), 0u);
    std::sort(by_name.begin(), by_name.end(), [&](uint32_t a, uint32_t b) { return names[a] < names[b]; });
    std::vector<uint32_t> rank2sid(V);
    for (uint32_t s = 0; s < V; ++s) rank2sid[by_name[s]] = s;

    int rc = XGM_OK;
    xgm_index* idx = new xgm_index();
    idx->device = device;
    idx->sections_owned = true;
    DevBuf d_thr, d_r2s, d_len64, d_tokstart, d_keys, d_keys2, d_tmp, d_flag, d_pidx, d_psid, d_pdoc, d_ptok, d_wdf, d_tps, d_mark,
        d_bflag, d_bidx, d_blkp, d_nwords, d_goff, d_has, d_newid, d_pblk, d_scalar;
    uint64_t n_tok = 0, n_post = 0, n_blk = 0, n_words = 0;
    uint32_t T = 0, wdf_ub_db = 0, doclen_lb = 0, doclen_ub = 0, overflow = 0;
    uint64_t total_len = 0;
    void** S = idx->d_sections;

    SY_TRY(d_thr.alloc((size_t)V * 8));
    SY_TRY(d_r2s.alloc((size_t)V * 4));
    SY_TRY(hipMemcpy(d_thr.p, thr.data(), (size_t)V * 8, hipMemcpyHostToDevice));
    SY_TRY(hipMemcpy(d_r2s.p, rank2sid.data(), (size_t)V * 4, hipMemcpyHostToDevice));
    SY_TRY(hipMalloc(&S[XGM_S_DOCLEN], ((size_t)n_local + 1) * 4));
    SY_TRY(d_len64.alloc((size_t)n_local * 8));
    SY_TRY(d_tokstart.alloc(((size_t)n_local + 1) * 8));
    hipLaunchKernelGGL(k_doclen, dim3(grid_for(n_local)), dim3(TB), 0, 0, cp, n_local, sp->n_shards, sp->shard, (uint32_t*)S[XGM_S_DOCLEN],
                       d_len64.as<uint64_t>());
    SY_TRY(excl_sum(d_len64.as<uint64_t>(), d_tokstart.as<uint64_t>(), (uint64_t)n_local, d_tmp));
    {
        uint64_t last_start = 0, last_len = 0;
        SY_TRY(hipMemcpy(&last_start, d_tokstart.as<uint64_t>() + (n_local - 1), 8, hipMemcpyDeviceToHost));
        SY_TRY(hipMemcpy(&last_len, d_len64.as<uint64_t>() + (n_local - 1), 8, hipMemcpyDeviceToHost));
        n_tok = last_start + last_len;
        total_len = n_tok;
        /* min document length */
        size_t bytes = 0;
        SY_TRY(d_scalar.alloc(64));
        SY_TRY(hipcub::DeviceReduce::Min(nullptr, bytes, d_len64.as<uint64_t>(), d_scalar.as<uint64_t>(), (int)n_local));
        SY_TRY(d_tmp.alloc(bytes));
        SY_TRY(hipcub::DeviceReduce::Min(d_tmp.p, bytes, d_len64.as<uint64_t>(), d_scalar.as<uint64_t>(), (int)n_local));
        uint64_t mn = 0;
        SY_TRY(hipMemcpy(&mn, d_scalar.p, 8, hipMemcpyDeviceToHost));
        doclen_lb = (uint32_t)mn;
        /* max document length (the wdf bound of an OP_SYNONYM) */
        SY_TRY(hipcub::DeviceReduce::Max(nullptr, bytes, d_len64.as<uint64_t>(), d_scalar.as<uint64_t>(), (int)n_local));
        SY_TRY(d_tmp.alloc(bytes));
        SY_TRY(hipcub::DeviceReduce::Max(d_tmp.p, bytes, d_len64.as<uint64_t>(), d_scalar.as<uint64_t>(), (int)n_local));
        SY_TRY(hipMemcpy(&mn, d_scalar.p, 8, hipMemcpyDeviceToHost));
        doclen_ub = (uint32_t)mn;
    }
    if (n_tok >= 0x7FFFFFFFull) { rc = xgm_set_error(XGM_E_INVALID, "shard too large for the synthetic builder (%llu tokens)", (unsigned long long)n_tok); goto fail; }
    d_len64.reset();

    /* tokens → sorted keys */
    SY_TRY(d_keys.alloc((size_t)n_tok * 8));
    SY_TRY(d_keys2.alloc((size_t)n_tok * 8));
    hipLaunchKernelGGL(k_tokens, dim3(grid_for((uint64_t)n_local * cp.len_hi)), dim3(TB), 0, 0, cp, d_thr.as<uint64_t>(), d_r2s.as<uint32_t>(),
                       n_local, sp->n_shards, sp->shard, (const uint32_t*)S[XGM_S_DOCLEN], d_tokstart.as<uint64_t>(), d_keys.as<uint64_t>());
    SY_TRY(hipGetLastError());
    {
        size_t bytes = 0;
        SY_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, d_keys.as<uint64_t>(), d_keys2.as<uint64_t>(), (int)n_tok, 0, 64));
        SY_TRY(d_tmp.alloc(bytes));
        SY_TRY(hipcub::DeviceRadixSort::SortKeys(d_tmp.p, bytes, d_keys.as<uint64_t>(), d_keys2.as<uint64_t>(), (int)n_tok, 0, 64));
    }
    d_keys.reset();
    d_tokstart.reset();

    /* postings */
    SY_TRY(d_flag.alloc((size_t)n_tok * 4));
    SY_TRY(d_pidx.alloc((size_t)n_tok * 4));
    hipLaunchKernelGGL(k_posting_flags, dim3(grid_for(n_tok)), dim3(TB), 0, 0, d_keys2.as<uint64_t>(), n_tok, d_flag.as<uint32_t>());
    SY_TRY(excl_sum(d_flag.as<uint32_t>(), d_pidx.as<uint32_t>(), n_tok, d_tmp));
    {
        uint32_t lf = 0, lp = 0;
        SY_TRY(hipMemcpy(&lf, d_flag.as<uint32_t>() + (n_tok - 1), 4, hipMemcpyDeviceToHost));
        SY_TRY(hipMemcpy(&lp, d_pidx.as<uint32_t>() + (n_tok - 1), 4, hipMemcpyDeviceToHost));
        n_post = (uint64_t)lp + lf;
    }
    SY_TRY(d_psid.alloc((size_t)n_post * 4));
    SY_TRY(d_pdoc.alloc((size_t)n_post * 4));
    SY_TRY(d_ptok.alloc(((size_t)n_post + 1) * 8));
    if (with_pos) {
        SY_TRY(hipMalloc(&S[XGM_S_POSITIONS], (size_t)n_tok * 2 + XGM_POS_PAD));
        SY_TRY(hipMemset((char*)S[XGM_S_POSITIONS] + (size_t)n_tok * 2, 0, XGM_POS_PAD));
    }
    hipLaunchKernelGGL(k_postings, dim3(grid_for(n_tok)), dim3(TB), 0, 0, d_keys2.as<uint64_t>(), d_flag.as<uint32_t>(), d_pidx.as<uint32_t>(), n_tok,
                       d_psid.as<uint32_t>(), d_pdoc.as<uint32_t>(), d_ptok.as<uint64_t>(), with_pos ? (uint16_t*)S[XGM_S_POSITIONS] : nullptr);
    SY_TRY(hipGetLastError());
    SY_TRY(hipDeviceSynchronize());
    d_keys2.reset(); d_flag.reset(); d_pidx.reset();
    SY_TRY(d_wdf.alloc((size_t)n_post * 4));
    hipLaunchKernelGGL(k_wdf, dim3(grid_for(n_post)), dim3(TB), 0, 0, d_ptok.as<uint64_t>(), n_post, n_tok, d_wdf.as<uint32_t>());
    {
        size_t bytes = 0;
        SY_TRY(hipcub::DeviceReduce::Max(nullptr, bytes, d_wdf.as<uint32_t>(), d_scalar.as<uint32_t>(), (int)n_post));
        SY_TRY(d_tmp.alloc(bytes));
        SY_TRY(hipcub::DeviceReduce::Max(d_tmp.p, bytes, d_wdf.as<uint32_t>(), d_scalar.as<uint32_t>(), (int)n_post));
        SY_TRY(hipMemcpy(&wdf_ub_db, d_scalar.p, 4, hipMemcpyDeviceToHost));
    }

    /* term starts, runs, blocks */
    SY_TRY(d_tps.alloc(((size_t)V + 1) * 8));
    hipLaunchKernelGGL(k_term_starts, dim3(grid_for(n_post + 1)), dim3(TB), 0, 0, d_psid.as<uint32_t>(), n_post, V, d_tps.as<uint64_t>());
    SY_TRY(d_mark.alloc((size_t)n_post * 8));
    hipLaunchKernelGGL(k_run_marks, dim3(grid_for(n_post)), dim3(TB), 0, 0, d_psid.as<uint32_t>(), d_pdoc.as<uint32_t>(), n_post, sb, d_mark.as<uint64_t>());
    {
        size_t bytes = 0;
        SY_TRY(hipcub::DeviceScan::InclusiveScan(nullptr, bytes, d_mark.as<uint64_t>(), d_mark.as<uint64_t>(), MaxOp(), (int)n_post));
        SY_TRY(d_tmp.alloc(bytes));
        SY_TRY(hipcub::DeviceScan::InclusiveScan(d_tmp.p, bytes, d_mark.as<uint64_t>(), d_mark.as<uint64_t>(), MaxOp(), (int)n_post));
    }
    SY_TRY(d_bflag.alloc((size_t)n_post * 4));
    SY_TRY(d_bidx.alloc((size_t)n_post * 4));
    hipLaunchKernelGGL(k_block_flags, dim3(grid_for(n_post)), dim3(TB), 0, 0, d_mark.as<uint64_t>(), n_post, d_bflag.as<uint32_t>());
    SY_TRY(excl_sum(d_bflag.as<uint32_t>(), d_bidx.as<uint32_t>(), n_post, d_tmp));
    {
        uint32_t lf = 0, lb = 0;
        SY_TRY(hipMemcpy(&lf, d_bflag.as<uint32_t>() + (n_post - 1), 4, hipMemcpyDeviceToHost));
        SY_TRY(hipMemcpy(&lb, d_bidx.as<uint32_t>() + (n_post - 1), 4, hipMemcpyDeviceToHost));
        n_blk = (uint64_t)lb + lf;
    }
    d_mark.reset();
    SY_TRY(d_blkp.alloc(((size_t)n_blk + 1) * 8));
    hipLaunchKernelGGL(k_block_first_posting, dim3(grid_for(n_post)), dim3(TB), 0, 0, d_bflag.as<uint32_t>(), d_bidx.as<uint32_t>(), n_post, d_blkp.as<uint64_t>());
    /* inclusive block index per posting = bidx + bflag - ... : block of posting p is bidx[p] if bflag[p] else bidx[p]-1;
     * only needed for the first posting of each term (always a block start), so bidx suffices */
    d_bflag.reset();
    SY_TRY(hipMalloc(&S[XGM_S_BLK_FIRST], (size_t)n_blk * 4));
    SY_TRY(hipMalloc(&S[XGM_S_BLK_META], (size_t)n_blk * 4));
    SY_TRY(hipMalloc(&S[XGM_S_BLK_WORD], (size_t)n_blk * 4));
    SY_TRY(hipMalloc(&S[XGM_S_BLK_POS], (size_t)n_blk * 4));
    SY_TRY(d_nwords.alloc((size_t)n_blk * 8));
    SY_TRY(d_goff.alloc(((size_t)n_blk + 1) * 8));
    hipLaunchKernelGGL(k_block_meta, dim3(grid_for(n_blk * 64)), dim3(TB), 0, 0, d_blkp.as<uint64_t>(), n_blk, n_post, d_pdoc.as<uint32_t>(),
                       d_wdf.as<uint32_t>(), (uint32_t*)S[XGM_S_BLK_FIRST], (uint32_t*)S[XGM_S_BLK_META], d_nwords.as<uint64_t>());
    SY_TRY(excl_sum(d_nwords.as<uint64_t>(), d_goff.as<uint64_t>(), n_blk, d_tmp));
    {
        uint64_t lo = 0, ln = 0;
        SY_TRY(hipMemcpy(&lo, d_goff.as<uint64_t>() + (n_blk - 1), 8, hipMemcpyDeviceToHost));
        SY_TRY(hipMemcpy(&ln, d_nwords.as<uint64_t>() + (n_blk - 1), 8, hipMemcpyDeviceToHost));
        n_words = lo + ln;
    }
    d_nwords.reset();
    SY_TRY(hipMalloc(&S[XGM_S_WORDS], ((size_t)n_words + XGM_WORD_PAD) * 4));
    SY_TRY(hipMemset(S[XGM_S_WORDS], 0, ((size_t)n_words + XGM_WORD_PAD) * 4));
    hipLaunchKernelGGL(k_block_pack, dim3(grid_for(n_blk * 64)), dim3(TB), 0, 0, d_blkp.as<uint64_t>(), n_blk, n_post, d_pdoc.as<uint32_t>(),
                       d_wdf.as<uint32_t>(), (const uint32_t*)S[XGM_S_BLK_META], d_goff.as<uint64_t>(), (uint32_t*)S[XGM_S_WORDS]);
    SY_TRY(hipGetLastError());

    /* dictionary */
    SY_TRY(d_has.alloc((size_t)V * 4));
    SY_TRY(d_newid.alloc((size_t)V * 4));
    hipLaunchKernelGGL(k_term_flags, dim3(grid_for(V)), dim3(TB), 0, 0, d_tps.as<uint64_t>(), V, d_has.as<uint32_t>());
    SY_TRY(excl_sum(d_has.as<uint32_t>(), d_newid.as<uint32_t>(), (uint64_t)V, d_tmp));
    {
        uint32_t lh = 0, ln = 0;
        SY_TRY(hipMemcpy(&lh, d_has.as<uint32_t>() + (V - 1), 4, hipMemcpyDeviceToHost));
        SY_TRY(hipMemcpy(&ln, d_newid.as<uint32_t>() + (V - 1), 4, hipMemcpyDeviceToHost));
        T = ln + lh;
    }
    SY_TRY(hipMalloc(&S[XGM_S_TERM_DF], (size_t)T * 4));
    SY_TRY(hipMalloc(&S[XGM_S_TERM_CF], (size_t)T * 4));
    SY_TRY(hipMalloc(&S[XGM_S_TERM_WDFUB], (size_t)T * 4));
    SY_TRY(hipMalloc(&S[XGM_S_TERM_FLAGS], (size_t)T * 4));
    SY_TRY(hipMalloc(&S[XGM_S_TERM_BLK], ((size_t)T + 1) * 8));
    SY_TRY(hipMalloc(&S[XGM_S_TERM_WORD], ((size_t)T + 1) * 8));
    SY_TRY(hipMalloc(&S[XGM_S_TERM_POS], ((size_t)T + 1) * 8));
    hipLaunchKernelGGL(k_term_tables, dim3(grid_for(V)), dim3(TB), 0, 0, d_tps.as<uint64_t>(), d_has.as<uint32_t>(), d_newid.as<uint32_t>(), V, T, n_post,
                       n_tok, n_blk, n_words, d_ptok.as<uint64_t>(), d_wdf.as<uint32_t>(), d_bidx.as<uint32_t>(), d_goff.as<uint64_t>(), wdf_ub_db,
                       with_pos, (uint32_t*)S[XGM_S_TERM_DF], (uint32_t*)S[XGM_S_TERM_CF], (uint32_t*)S[XGM_S_TERM_WDFUB],
                       (uint32_t*)S[XGM_S_TERM_FLAGS], (uint64_t*)S[XGM_S_TERM_BLK], (uint64_t*)S[XGM_S_TERM_WORD], (uint64_t*)S[XGM_S_TERM_POS]);
    SY_TRY(hipMemset(d_scalar.p, 0, 64));
    hipLaunchKernelGGL(k_block_relative, dim3(grid_for(n_blk)), dim3(TB), 0, 0, d_blkp.as<uint64_t>(), n_blk, d_psid.as<uint32_t>(), d_newid.as<uint32_t>(),
                       (const uint64_t*)S[XGM_S_TERM_WORD], (const uint64_t*)S[XGM_S_TERM_POS], d_goff.as<uint64_t>(), d_ptok.as<uint64_t>(), with_pos,
                       (uint32_t*)S[XGM_S_BLK_WORD], (uint32_t*)S[XGM_S_BLK_POS], d_scalar.as<uint32_t>());
    SY_TRY(hipGetLastError());
    SY_TRY(hipMemcpy(&overflow, d_scalar.p, 4, hipMemcpyDeviceToHost));
    if (overflow) { rc = xgm_set_error(XGM_E_INVALID, "a term is too large for 32-bit block offsets"); goto fail; }
    if (n_blk >= 0xFFFFFFFFull) { rc = xgm_set_error(XGM_E_INVALID, "too many blocks"); goto fail; }

    /* host dictionary */
    {
        std::vector<uint32_t> has(V);
        SY_TRY(hipMemcpy(has.data(), d_has.p, (size_t)V * 4, hipMemcpyDeviceToHost));
        idx->term_df.resize(T); idx->term_cf.resize(T); idx->term_wdfub.resize(T); idx->term_flags.resize(T);
        idx->term_blk.resize((size_t)T + 1); idx->term_word.resize((size_t)T + 1);
        SY_TRY(hipMemcpy(idx->term_df.data(), S[XGM_S_TERM_DF], (size_t)T * 4, hipMemcpyDeviceToHost));
        SY_TRY(hipMemcpy(idx->term_cf.data(), S[XGM_S_TERM_CF], (size_t)T * 4, hipMemcpyDeviceToHost));
        SY_TRY(hipMemcpy(idx->term_wdfub.data(), S[XGM_S_TERM_WDFUB], (size_t)T * 4, hipMemcpyDeviceToHost));
        SY_TRY(hipMemcpy(idx->term_flags.data(), S[XGM_S_TERM_FLAGS], (size_t)T * 4, hipMemcpyDeviceToHost));
        SY_TRY(hipMemcpy(idx->term_blk.data(), S[XGM_S_TERM_BLK], ((size_t)T + 1) * 8, hipMemcpyDeviceToHost));
        SY_TRY(hipMemcpy(idx->term_word.data(), S[XGM_S_TERM_WORD], ((size_t)T + 1) * 8, hipMemcpyDeviceToHost));
        idx->str_off.assign(1, 0);
        for (uint32_t s = 0; s < V; ++s) {
            if (!has[s]) continue;
            const std::string& nm = names[by_name[s]];
            idx->str_bytes.insert(idx->str_bytes.end(), nm.begin(), nm.end());
            idx->str_off.push_back(idx->str_bytes.size());
        }
    }
    {
        xgm_seg_header& h = idx->hdr;
        memset(&h, 0, sizeof h);
        memcpy(h.magic, XGM_SEG_MAGIC, 8);
        h.version = XGM_SEG_VERSION; h.stripe_bits = sb; h.block_size = XGM_BLOCK; h.n_terms = T;
        h.lastdocid = n_local; h.doccount = n_local; h.has_positions = (uint32_t)with_pos; h.doclen_lower_bound = doclen_lb;
        h.doclen_upper_bound = doclen_ub;
        h.wdf_upper_bound = wdf_ub_db; h.total_length = total_len; h.revision = 1; h.n_postings = n_post;
        h.n_positions = with_pos ? n_tok : 0; h.n_blocks = n_blk; h.n_words = n_words;
        uint64_t* sz = h.sec_bytes;
        sz[XGM_S_DOCLEN] = ((uint64_t)n_local + 1) * 4;
        sz[XGM_S_TERM_DF] = sz[XGM_S_TERM_CF] = sz[XGM_S_TERM_WDFUB] = sz[XGM_S_TERM_FLAGS] = (uint64_t)T * 4;
        sz[XGM_S_TERM_BLK] = sz[XGM_S_TERM_WORD] = sz[XGM_S_TERM_POS] = ((uint64_t)T + 1) * 8;
        sz[XGM_S_BLK_FIRST] = sz[XGM_S_BLK_META] = sz[XGM_S_BLK_WORD] = sz[XGM_S_BLK_POS] = n_blk * 4;
        sz[XGM_S_WORDS] = (n_words + XGM_WORD_PAD) * 4;
        sz[XGM_S_POSITIONS] = with_pos ? n_tok * 2 + XGM_POS_PAD : 0;
        sz[XGM_S_STR_OFF] = ((uint64_t)T + 1) * 8;
        sz[XGM_S_STR_BYTES] = idx->str_bytes.size();
        idx->device_bytes = 0;
        for (int s = 0; s < XGM_S_COUNT; ++s) if (s != XGM_S_STR_OFF && s != XGM_S_STR_BYTES) idx->device_bytes += sz[s];
    }
    if (!with_pos) SY_TRY(hipMalloc(&S[XGM_S_POSITIONS], 64));
    SY_TRY(hipDeviceSynchronize());
    {
        xgm_seg_dev& v = idx->view;
        v.doclen = (const uint32_t*)S[XGM_S_DOCLEN];
        v.term_blk = (const uint64_t*)S[XGM_S_TERM_BLK];
        v.term_word = (const uint64_t*)S[XGM_S_TERM_WORD];
        v.term_pos = (const uint64_t*)S[XGM_S_TERM_POS];
        v.blk_first = (const uint32_t*)S[XGM_S_BLK_FIRST];
        v.blk_meta = (const uint32_t*)S[XGM_S_BLK_META];
        v.blk_word = (const uint32_t*)S[XGM_S_BLK_WORD];
        v.blk_pos = (const uint32_t*)S[XGM_S_BLK_POS];
        v.words = (const uint32_t*)S[XGM_S_WORDS];
        v.positions = (const unsigned char*)S[XGM_S_POSITIONS];
        v.term_flags = (const uint32_t*)S[XGM_S_TERM_FLAGS];
        v.stripe_bits = sb;
        v.lastdocid = n_local;
    }
    if ((rc = xgm_build_dense(idx))) goto fail;
    *out = idx;
    return XGM_OK;

fail:
    for (int s = 0; s < XGM_S_COUNT; ++s) if (idx->d_sections[s]) hipFree(idx->d_sections[s]);
    delete idx;
    return rc;
}
