/* Probe containers for dense terms (see xgm_segment.h).  Built on the device from the block-encoded
 * posting lists when an index is opened — an acceleration structure of the query path, like a
 * bitmap index beside the compressed lists; the reference's closest analogue is the B-tree skip
 * (find_entry) that GlassPostList::skip_to uses to avoid scanning (glass_postlist.cc:959-991). */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "xgm_device.h"
#include "xgm_internal.h"
#include "xgm_launch.h"

namespace {

#define DN_TRY(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) { rc = xgm_launch_error(#expr, (int)e_, hipGetErrorString(e_)); goto fail; } \
    } while (0)

constexpr uint32_t kStage = 272;

__device__ __forceinline__ uint32_t dn_extract(const uint32_t* s, uint32_t idx, uint32_t bw) {
    uint32_t bit = idx * bw, w = bit >> 5;
    uint32_t v = __builtin_amdgcn_alignbit(s[w + 1], s[w], bit & 31u);
    return bw >= 32u ? v : (v & ((1u << bw) - 1u));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dn_dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ uint32_t dn_scan(uint32_t v) {
    v += dn_dpp<0x111, 0xf>(v); v += dn_dpp<0x112, 0xf>(v); v += dn_dpp<0x114, 0xf>(v); v += dn_dpp<0x118, 0xf>(v);
    v += dn_dpp<0x142, 0xa>(v); v += dn_dpp<0x143, 0xc>(v);
    return v;
}

/* postings per (dense term, stripe) */
__global__ void k_dense_count(xgm_seg_dev seg, const uint32_t* __restrict__ dense_terms, uint32_t n_stripes, uint32_t* __restrict__ cnt) {
    const uint32_t d = blockIdx.y, t = dense_terms[d];
    const uint64_t b0 = seg.term_blk[t], b1 = seg.term_blk[t + 1];
    for (uint64_t b = b0 + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < b1; b += (uint64_t)gridDim.x * blockDim.x)
        atomicAdd(&cnt[(size_t)d * n_stripes + (seg.blk_first[b] >> seg.stripe_bits)], XGM_META_COUNT(seg.blk_meta[b]));
}

/* one workgroup per (stripe, dense term): decode the run, write the bitmap and the per-slot wdf+1 bytes */
__global__ __launch_bounds__(256) void k_dense_fill(xgm_seg_dev seg, const uint32_t* __restrict__ dense_terms, uint32_t n_stripes,
                                                    const uint32_t* __restrict__ dir, unsigned char* __restrict__ data, int with_pos,
                                                    uint32_t* __restrict__ wdf_max) {
    __shared__ uint32_t bitmap[256];
    __shared__ uint32_t bitmap2[256];     /* wdf >= 2 */
    __shared__ uint32_t wmax_s;
    __shared__ uint32_t pbase[128];       /* with_pos: position-entry offset (relative to the term) of the first posting of every 64-slot bucket */
    __shared__ uint32_t stage_all[4 * kStage];
    __shared__ uint32_t run[2];
    const uint32_t s = blockIdx.x, d = blockIdx.y, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t off = dir[(size_t)d * n_stripes + s];
    if (off == 0) return;
    const uint32_t t = dense_terms[d];
    const uint32_t SB = seg.stripe_bits, W = 1u << SB, NW = W / 32u;
    const uint32_t b0 = (uint32_t)seg.term_blk[t], b1 = (uint32_t)seg.term_blk[t + 1];
    unsigned char* cont = data + (size_t)off * 16;
    unsigned char* wdf_out = cont + (size_t)NW * 4;
    if (tid < NW) { bitmap[tid] = 0; bitmap2[tid] = 0; }
    if (tid == 0) wmax_s = 0;
    if (tid < 128u) pbase[tid] = 0xFFFFFFFFu;
    for (uint32_t i = tid; i < W / 16u; i += 256u) reinterpret_cast<uint4*>(wdf_out)[i] = make_uint4(0, 0, 0, 0);
    if (tid == 0) {                       /* the run's blocks: binary search on the term's block firsts */
        uint32_t lo = b0, hi = b1;
        const uint32_t key0 = s << SB;
        while (lo < hi) { uint32_t mid = lo + (hi - lo) / 2; if (seg.blk_first[mid] < key0) lo = mid + 1; else hi = mid; }
        run[0] = lo;
        uint32_t e = lo;
        while (e < b1 && (seg.blk_first[e] >> SB) == s) ++e;
        run[1] = e;
    }
    __syncthreads();
    const uint32_t rb = run[0], nb = run[1] - rb;
    uint32_t* stage = stage_all + wave * kStage;
    uint32_t wmax = 0;
    for (uint32_t j = wave; j < nb; j += 4u) {
        const uint32_t b = rb + j, meta = seg.blk_meta[b], first = seg.blk_first[b];
        const uint32_t n = XGM_META_COUNT(meta), bwg = XGM_META_BWG(meta), bww = XGM_META_BWW(meta);
        const uint32_t ngw = (n * bwg + 31u) >> 5, nww = (n * bww + 31u) >> 5;
        const uint32_t* payload = seg.words + seg.term_word[t] + seg.blk_word[b];
        for (uint32_t w = lane; w < ngw + nww + 2u; w += 64u) stage[w] = payload[w];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t i0 = lane * 2u, i1 = i0 + 1u;
        const bool v0 = i0 < n, v1 = i1 < n;
        const uint32_t g0 = (v0 && i0 > 0u) ? dn_extract(stage, i0, bwg) + 1u : 0u;
        const uint32_t g1 = v1 ? dn_extract(stage, i1, bwg) + 1u : 0u;
        const uint32_t w0 = v0 ? dn_extract(stage + ngw, i0, bww) : 0u;
        const uint32_t w1 = v1 ? dn_extract(stage + ngw, i1, bww) : 0u;
        const uint32_t local = g0 + g1;
        const uint32_t excl = dn_scan(local) - local;
        const uint32_t d0 = first + excl + g0, d1 = d0 + g1;
        /* a posting's positions start at entry blk_pos + Σ wdf of the earlier postings of its block */
        const uint32_t lw = w0 + w1;
        const uint32_t pex = with_pos ? seg.blk_pos[b] + dn_scan(lw) - lw : 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (v0) { const uint32_t sl = d0 & (W - 1u); atomicOr(&bitmap[sl >> 5], 1u << (sl & 31u)); if (w0 >= 2u) atomicOr(&bitmap2[sl >> 5], 1u << (sl & 31u)); wdf_out[sl] = (unsigned char)(w0 + 1u); if (with_pos) atomicMin(&pbase[sl >> 6], pex); }
        if (v1) { const uint32_t sl = d1 & (W - 1u); atomicOr(&bitmap[sl >> 5], 1u << (sl & 31u)); if (w1 >= 2u) atomicOr(&bitmap2[sl >> 5], 1u << (sl & 31u)); wdf_out[sl] = (unsigned char)(w1 + 1u); if (with_pos) atomicMin(&pbase[sl >> 6], pex + w0); }
        wmax = max(wmax, max(w0, w1));
    }
    if (wmax) atomicMax(&wmax_s, wmax);
    __syncthreads();
    if (tid < NW) reinterpret_cast<uint32_t*>(cont)[tid] = bitmap[tid];
    if (with_pos && tid < W / 64u) reinterpret_cast<uint32_t*>(wdf_out + W)[tid] = pbase[tid];
    if (tid < NW) reinterpret_cast<uint32_t*>(wdf_out + W + (with_pos ? NW * 2u : 0u))[tid] = bitmap2[tid];
    if (tid == 0 && wmax_s) atomicMax(&wdf_max[d], wmax_s);            /* the term's true largest wdf: the disjunction's pruning bound */
}

/* *misfit is raised when a length does not fit T above the base (a header whose bounds do not cover its own lengths): the narrow
 * array is then not used — a truncated length would change BM25 weights silently */
/* flat posting arrays: one wave per term without containers; lane j decodes block j of the term's next 64 blocks on its own (a block of
 * the long tail holds a dozen postings), a wave prefix sum of the blocks' counts gives every block its place */
__global__ __launch_bounds__(256) void k_flat_fill(xgm_seg_dev seg, const uint32_t* __restrict__ flat_terms, uint32_t n_flat, const uint64_t* __restrict__ flat_off,
                                                   uint32_t* __restrict__ out_did, unsigned char* __restrict__ out_wdf, uint32_t* __restrict__ out_pos) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= n_flat) return;                                       /* (whole waves leave: no barrier below) */
    const uint32_t t = flat_terms[w];
    const uint64_t b0 = seg.term_blk[t], b1 = seg.term_blk[t + 1], tw = seg.term_word[t];
    uint64_t place = flat_off[t];
    for (uint64_t c = b0; c < b1; c += 64u) {
        const uint64_t b = c + lane;
        const uint32_t meta = b < b1 ? seg.blk_meta[b] : 0u;
        const uint32_t n = b < b1 ? XGM_META_COUNT(meta) : 0u;
        const uint32_t incl = dn_scan(n);
        if (n) {
            const uint32_t bwg = XGM_META_BWG(meta), bww = XGM_META_BWW(meta), ngw = (n * bwg + 31u) >> 5;
            const uint32_t* words = seg.words + tw + seg.blk_word[b];
            uint32_t did = seg.blk_first[b];
            uint32_t pos = out_pos ? seg.blk_pos[b] : 0u;              /* a posting's positions start at entry blk_pos + the wdf of the block's earlier postings */
            const uint64_t o = place + incl - n;
            for (uint32_t i = 0; i < n; ++i) {
                if (i) did += dn_extract(words, i, bwg) + 1u;
                const uint32_t wdf = dn_extract(words + ngw, i, bww);
                out_did[o + i] = did;
                out_wdf[o + i] = (unsigned char)wdf;
                if (out_pos) { out_pos[o + i] = pos; pos += wdf; }
            }
        }
        place += (uint64_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
}

template <typename T>
__global__ void k_narrow_doclen(const uint32_t* __restrict__ doclen, uint32_t n, uint32_t base, T* __restrict__ out, uint32_t* __restrict__ misfit) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint32_t v = doclen[i], off = v >= base ? v - base : 0u;                         /* (entry 0 and deleted documents hold 0) */
        if ((v != 0u && v < base) || off > (uint32_t)(T)~(T)0) atomicOr(misfit, 1u);
        out[i] = (T)off;
    }
}

}  // namespace

/* doclen[] as offsets from the shard's shortest document, one or two bytes each (xgm_seg_dev::doclen_narrow) */
static int build_narrow_doclen(xgm_index* idx) {
    idx->view.doclen_narrow = nullptr; idx->view.doclen_narrow_bits = 0; idx->view.doclen_base = 0;
    if (getenv("XGM_NO_NARROW_DOCLEN") || !idx->view.doclen) return XGM_OK;              /* A/B switch for measurements */
    const uint32_t lb = idx->hdr.doclen_lower_bound, ub = idx->hdr.doclen_upper_bound, n = idx->hdr.lastdocid + 1u;
    if (ub < lb || ub - lb >= 65536u) return XGM_OK;
    const uint32_t bits = ub - lb < 256u ? 8u : 16u;
    if (idx->d_doclen_narrow) {                              /* a second xgm_build_dense on the same index: the old array goes first */
        hipFree(idx->d_doclen_narrow);
        idx->d_doclen_narrow = nullptr;
    }
    void* d = nullptr;
    const size_t arr = ((size_t)n * (bits / 8u) + 64 + 15) & ~(size_t)15;
    if (hipMalloc(&d, arr + 16) != hipSuccess) return xgm_set_error(XGM_E_NOMEM, "narrow document lengths: out of device memory");
    uint32_t* d_misfit = reinterpret_cast<uint32_t*>((unsigned char*)d + arr);
    uint32_t misfit = 0;
    if (hipMemset(d_misfit, 0, 16) != hipSuccess) { hipFree(d); return xgm_set_error(XGM_E_DEVICE, "narrow document lengths: memset failed"); }
    if (bits == 8u) hipLaunchKernelGGL(k_narrow_doclen<uint8_t>, dim3((n + 255u) / 256u), dim3(256), 0, 0, idx->view.doclen, n, lb, (uint8_t*)d, d_misfit);
    else hipLaunchKernelGGL(k_narrow_doclen<uint16_t>, dim3((n + 255u) / 256u), dim3(256), 0, 0, idx->view.doclen, n, lb, (uint16_t*)d, d_misfit);
    if (hipGetLastError() != hipSuccess || hipMemcpy(&misfit, d_misfit, 4, hipMemcpyDeviceToHost) != hipSuccess) { hipFree(d); return xgm_set_error(XGM_E_DEVICE, "narrow document lengths: kernel failed"); }
    if (misfit) { hipFree(d); return XGM_OK; }               /* the header's bounds do not hold: every kernel reads the u32 lengths */
    idx->d_doclen_narrow = d;
    idx->view.doclen_narrow = (const unsigned char*)d;
    idx->view.doclen_narrow_bits = bits;
    idx->view.doclen_base = lb;
    idx->device_bytes += (size_t)n * (bits / 8u);
    return XGM_OK;
}

/* Flat posting arrays of the terms without containers (xgm_seg_dev::flat_*): decoded once, here, from the blocks already in HBM. */
static int build_flat(xgm_index* idx) {
    idx->view.flat_off = nullptr; idx->view.flat_did = nullptr; idx->view.flat_wdf = nullptr; idx->view.flat_pos = nullptr;
    for (void** p : {&idx->d_flat_off, &idx->d_flat_did, &idx->d_flat_wdf, &idx->d_flat_pos}) if (*p) { hipFree(*p); *p = nullptr; }
    idx->flat_bytes = 0; idx->flat_postings = 0;
    /* XGM_NO_DENSE = no acceleration structures at all (the variant tests run the block decode K1 that way); XGM_NO_FLAT: this one off */
    if (getenv("XGM_NO_DENSE") || getenv("XGM_NO_FLAT")) return XGM_OK;
    const uint32_t T = idx->hdr.n_terms;
    std::vector<uint64_t> off((size_t)T + 1, 0);
    std::vector<uint32_t> terms;
    uint64_t n = 0;
    for (uint32_t t = 0; t < T; ++t) {
        off[t] = n;
        const bool dense = idx->view.n_dense && (uint64_t)idx->term_df[t] >= idx->dense_min_df && idx->term_wdfub[t] <= 254u;
        if (!dense && idx->term_df[t] && idx->term_wdfub[t] <= 254u) { n += idx->term_df[t]; terms.push_back(t); }
    }
    off[T] = n;
    if (n == 0) return XGM_OK;
    int rc = XGM_OK;
    uint32_t* d_terms = nullptr;
    /* The flat arrays are an ACCELERATOR (every kernel still has the blocks): they never fail an open.  Budget: XGM_FLAT_MAX_BYTES
     * (default 16 GiB) and at most half of the device memory that is free right now — several shards share a GPU; beyond it the position
     * starts go first (positional queries led by a long-tail term take the queue path again), then the arrays altogether. */
    bool want_pos = idx->hdr.has_positions && !getenv("XGM_NO_FLAT_PHRASE");      /* (A/B switch: positional queries never take the flat body) */
    {
        size_t budget = getenv("XGM_FLAT_MAX_BYTES") ? (size_t)strtoull(getenv("XGM_FLAT_MAX_BYTES"), nullptr, 0) : (size_t)16 << 30;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min(budget, free_b / 2);
        const size_t base_b = off.size() * 8 + n * 5 + 512 + terms.size() * 4;
        if (want_pos && base_b + n * 4 + 256 > budget) want_pos = false;
        if (base_b > budget) return XGM_OK;
    }
#define FLAT_TRY(expr) do { if ((expr) != hipSuccess) { (void)hipGetLastError(); rc = XGM_OK; goto fail; } } while (0)       /* out of memory: no flat arrays, the index opens */
    FLAT_TRY(hipMalloc(&idx->d_flat_off, off.size() * 8));
    FLAT_TRY(hipMalloc(&idx->d_flat_did, n * 4 + 256));
    FLAT_TRY(hipMalloc(&idx->d_flat_wdf, n + 256));
    if (want_pos && hipMalloc(&idx->d_flat_pos, n * 4 + 256) != hipSuccess) { (void)hipGetLastError(); idx->d_flat_pos = nullptr; }
    FLAT_TRY(hipMalloc((void**)&d_terms, terms.size() * 4));
#undef FLAT_TRY
    DN_TRY(hipMemcpy(idx->d_flat_off, off.data(), off.size() * 8, hipMemcpyHostToDevice));
    DN_TRY(hipMemcpy(d_terms, terms.data(), terms.size() * 4, hipMemcpyHostToDevice));
    DN_TRY(hipMemset((unsigned char*)idx->d_flat_did + n * 4, 0xFF, 256));         /* (a round reads up to 63 entries past a slice's end: sentinels, never used) */
    DN_TRY(hipMemset((unsigned char*)idx->d_flat_wdf + n, 0, 256));
    hipLaunchKernelGGL(k_flat_fill, dim3((unsigned)((terms.size() + 3u) / 4u)), dim3(256), 0, 0, idx->view, d_terms, (uint32_t)terms.size(),
                       (const uint64_t*)idx->d_flat_off, (uint32_t*)idx->d_flat_did, (unsigned char*)idx->d_flat_wdf, (uint32_t*)idx->d_flat_pos);
    DN_TRY(hipGetLastError());
    DN_TRY(hipDeviceSynchronize());
    hipFree(d_terms);
    idx->view.flat_off = (const uint64_t*)idx->d_flat_off;
    idx->view.flat_did = (const uint32_t*)idx->d_flat_did;
    idx->view.flat_wdf = (const unsigned char*)idx->d_flat_wdf;
    idx->view.flat_pos = (const uint32_t*)idx->d_flat_pos;
    idx->flat_postings = n;
    idx->flat_bytes = off.size() * 8 + n * (idx->d_flat_pos ? 9 : 5) + 768;
    idx->device_bytes += idx->flat_bytes;
    return XGM_OK;
fail:
    if (d_terms) hipFree(d_terms);
    for (void** p : {&idx->d_flat_off, &idx->d_flat_did, &idx->d_flat_wdf, &idx->d_flat_pos}) if (*p) { hipFree(*p); *p = nullptr; }
    return rc;
}

static int build_containers(xgm_index* idx);

int xgm_build_dense(xgm_index* idx) {
    if (int rc_ = build_narrow_doclen(idx)) return rc_;
    if (int rc_ = build_containers(idx)) return rc_;
    return build_flat(idx);
}

static int build_containers(xgm_index* idx) {
    idx->view.dense_id = nullptr; idx->view.dense_dir = nullptr; idx->view.dense_data = nullptr;
    idx->view.n_dense = 0; idx->view.dense_plane = 0;
    idx->term_wdfmax.clear();
    const uint32_t SB = idx->hdr.stripe_bits;
    const uint32_t n_stripes = (idx->hdr.lastdocid >> SB) + 1u;
    idx->view.n_stripes = n_stripes;
    if (getenv("XGM_NO_DENSE")) return XGM_OK;                          /* A/B switch for measurements */
    const uint32_t NW = (1u << SB) / 32u;
    /* indexes with positions: the containers also carry, per 64-slot bucket, where the bucket's first posting keeps its
     * positions, so that the positional filter can use the probe path (the wdf bytes of the bucket give the rest) */
    const int with_pos = idx->hdr.has_positions ? 1 : 0;
    const uint32_t min_avg = getenv("XGM_DENSE_MIN_AVG") ? (uint32_t)atoi(getenv("XGM_DENSE_MIN_AVG")) : XGM_DENSE_MIN_AVG;   /* tuning knob */
    std::vector<uint32_t> dense_terms;
    std::vector<uint32_t> dense_id(idx->hdr.n_terms, 0xFFFFFFFFu);
    uint32_t max_blocks = 1;
    for (uint32_t t = 0; t < idx->hdr.n_terms; ++t) {
        if ((uint64_t)idx->term_df[t] >= (uint64_t)min_avg * n_stripes && idx->term_wdfub[t] <= 254u) {
            dense_id[t] = (uint32_t)dense_terms.size();
            dense_terms.push_back(t);
            max_blocks = std::max<uint32_t>(max_blocks, (uint32_t)(idx->term_blk[t + 1] - idx->term_blk[t]));
        }
    }
    const uint32_t n_dense = (uint32_t)dense_terms.size();
    if (n_dense == 0) return XGM_OK;
    int rc = XGM_OK;
    uint32_t *d_terms = nullptr, *d_cnt = nullptr, *d_wmax = nullptr;
    std::vector<uint32_t> wmax;
    const uint32_t plane_off = NW * 4u + NW * 32u + (with_pos ? NW * 2u : 0u);
    std::vector<uint32_t> cnt((size_t)n_dense * n_stripes), dir((size_t)n_dense * n_stripes, 0u);
    uint64_t units = 1;                                                  /* 16-byte units; offset 0 means "no container" */
    DN_TRY(hipMalloc((void**)&d_terms, (size_t)n_dense * 4));
    DN_TRY(hipMalloc((void**)&d_cnt, cnt.size() * 4));
    DN_TRY(hipMemcpy(d_terms, dense_terms.data(), (size_t)n_dense * 4, hipMemcpyHostToDevice));
    DN_TRY(hipMemset(d_cnt, 0, cnt.size() * 4));
    {
        dim3 grid(std::min<uint32_t>((max_blocks + 255u) / 256u, 4096u), n_dense);
        hipLaunchKernelGGL(k_dense_count, grid, dim3(256), 0, 0, idx->view, d_terms, n_stripes, d_cnt);
        DN_TRY(hipGetLastError());
    }
    DN_TRY(hipMemcpy(cnt.data(), d_cnt, cnt.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < cnt.size(); ++i) {
        if (!cnt[i]) continue;
        if (units > 0xFFFFFFFFull) { rc = xgm_set_error(XGM_E_INVALID, "dense containers exceed the 32-bit directory"); goto fail; }
        dir[i] = (uint32_t)units;
        units += ((uint64_t)plane_off + (uint64_t)NW * 4) / 16;   /* bitmap + one byte per slot (+ a position base per 64 slots) + the wdf >= 2 bitmap */
    }
    DN_TRY(hipMalloc(&idx->d_dense_id, dense_id.size() * 4));
    DN_TRY(hipMalloc(&idx->d_dense_dir, dir.size() * 4));
    DN_TRY(hipMalloc(&idx->d_dense_data, units * 16 + 64));
    DN_TRY(hipMemcpy(idx->d_dense_id, dense_id.data(), dense_id.size() * 4, hipMemcpyHostToDevice));
    DN_TRY(hipMemcpy(idx->d_dense_dir, dir.data(), dir.size() * 4, hipMemcpyHostToDevice));
    DN_TRY(hipMalloc((void**)&d_wmax, (size_t)n_dense * 4));
    DN_TRY(hipMemset(d_wmax, 0, (size_t)n_dense * 4));
    hipLaunchKernelGGL(k_dense_fill, dim3(n_stripes, n_dense), dim3(256), 0, 0, idx->view, d_terms, n_stripes, (const uint32_t*)idx->d_dense_dir,
                       (unsigned char*)idx->d_dense_data, with_pos, d_wmax);
    DN_TRY(hipGetLastError());
    DN_TRY(hipDeviceSynchronize());
    wmax.resize(n_dense);
    DN_TRY(hipMemcpy(wmax.data(), d_wmax, (size_t)n_dense * 4, hipMemcpyDeviceToHost));
    idx->term_wdfmax.assign(idx->term_wdfub.begin(), idx->term_wdfub.end());
    for (uint32_t d = 0; d < n_dense; ++d)
        if (wmax[d] && wmax[d] < idx->term_wdfmax[dense_terms[d]]) idx->term_wdfmax[dense_terms[d]] = wmax[d];
    idx->dense_bytes = dense_id.size() * 4 + dir.size() * 4 + units * 16;
    idx->device_bytes += idx->dense_bytes;
    idx->view.dense_id = (const uint32_t*)idx->d_dense_id;
    idx->view.dense_dir = (const uint32_t*)idx->d_dense_dir;
    idx->view.dense_data = (const unsigned char*)idx->d_dense_data;
    idx->view.n_dense = n_dense;
    idx->view.dense_pos = (uint32_t)with_pos;
    idx->view.dense_plane = plane_off;
    idx->dense_min_df = (uint64_t)min_avg * n_stripes;
    hipFree(d_terms); hipFree(d_cnt); hipFree(d_wmax);
    return XGM_OK;
fail:
    if (d_terms) hipFree(d_terms);
    if (d_cnt) hipFree(d_cnt);
    if (d_wmax) hipFree(d_wmax);
    if (idx->d_dense_id) { hipFree(idx->d_dense_id); idx->d_dense_id = nullptr; }
    if (idx->d_dense_dir) { hipFree(idx->d_dense_dir); idx->d_dense_dir = nullptr; }
    if (idx->d_dense_data) { hipFree(idx->d_dense_data); idx->d_dense_data = nullptr; }
    return rc;
}
