/* C-ABI entry points: index lifetime, dictionary lookup, search drivers.  Host code only; the
 * kernels live in xgm_kernels.hip and are reached through xgm_launch.h.  There is deliberately no
 * CPU implementation of the search path here: without a HIP device the search calls fail. */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <thread>
#include <cstdarg>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <string>

#include "xgm_internal.h"
#include "xgm_launch.h"

int xgm_read_segment_file(const char* path, XgmSegmentBlob* blob);

/* ------------------------------------------------------------------ errors ------------------- */

static thread_local char g_err[512] = "";

int xgm_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int xgm_launch_error(const char* what, int code, const char* msg) {
    return xgm_set_error(XGM_E_DEVICE, "%s failed (%d): %s", what, code, msg ? msg : "?");
}

extern "C" const char* xgm_last_error(void) { return g_err; }
extern "C" const char* xgm_version(void) { return "xgm 0.1 (gfx950)"; }

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return xgm_launch_error(#expr, (int)e_, hipGetErrorString(e_)); \
    } while (0)

static int use_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return xgm_set_error(XGM_E_NO_DEVICE, "no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n) return xgm_set_error(XGM_E_NO_DEVICE, "HIP device %d out of range (have %d)", device, n);
    HIP_TRY(hipSetDevice(device));
    return XGM_OK;
}

/* ------------------------------------------------------------------ scratch ------------------ */

struct XgmScratch {
    hipStream_t stream = nullptr;
    /* device */
    xgm_dev_query* d_queries = nullptr;                /* view into d_in */
    xgm_cand* d_cand = nullptr; size_t cap_cand = 0;
    xgm_group_hdr* d_ghdr = nullptr; size_t cap_ghdr = 0;
    uint32_t* d_kq = nullptr; double* d_maxposs = nullptr;   /* views into d_in */
    uint32_t* d_mkq = nullptr; size_t cap_mkq = 0;     /* k[] of xgm_merge_shards_device */
    uint32_t* d_hist = nullptr; size_t cap_hist = 0;   /* [nq][XGM_OR_HIST] weight histograms of xgm_orw_kernel */
    xgm_work* d_work = nullptr;                        /* views into d_in */
    uint32_t* d_goff = nullptr;
    void* d_in = nullptr; size_t cap_in = 0;           /* all per-call inputs, one upload */
    xgm_hit* d_hits = nullptr; size_t cap_hits = 0;
    xgm_result_hdr* d_hdrs = nullptr; size_t cap_hdrs = 0;
    unsigned char* d_sorted = nullptr; size_t cap_sorted = 0;   /* xgm_search_sorted*: query, work list, headers, counters, candidates */
    void* h_sorted = nullptr; size_t cap_hsorted = 0;
    xgm_hit* d_part_hits = nullptr; size_t cap_part_hits = 0;   /* results of a heavy batch's parts (run_class_batch, bp.parts > 1) */
    xgm_result_hdr* d_part_hdrs = nullptr; size_t cap_part_hdrs = 0;
    xgm_hit* d_arena = nullptr; size_t cap_arena = 0;           /* XGM_REPLAY_BATCH_COUNT on conjunctions: every match of the batch (xgm_all_out) */
    unsigned char* d_count = nullptr; size_t cap_count = 0;     /* ... the arena's cursor, the units' chunk tables, arrival states and document counts */
    unsigned char* d_all = nullptr; size_t cap_all = 0;         /* xgm_search_all: counter, unordered + ordered match lists, the sort's temporary storage */
    bool inorder = false;                                       /* this batch's upload and download go on the batch's stream (the dispatcher's small batches: fewer HIP calls) */
    bool arrive_dirty = false;                                  /* a fused launch on this scratch was not enqueued completely: zero the counters before the next */
    uint32_t* d_arrive = nullptr; size_t cap_arrive = 0;        /* per-query arrival counters of a launch that finishes its queries itself (zero between launches) */
    /* pinned host */
    void* h_up = nullptr; size_t cap_up = 0;
    void* h_down = nullptr; size_t cap_down = 0;
    void* h_work = nullptr; size_t cap_hwork = 0;     /* pinned copy of the work list + group offsets */
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t last_stop = nullptr; /* the completion event the LAST launch of the current batch carries itself (run_class_batch: fused wave kernel), or null */
    hipEvent_t ev_done = nullptr;   /* recorded after an asynchronous call that still uses this scratch */
    bool pending = false;
};

template <class T>
static int grow(T** p, size_t* cap, size_t need) {
    if (need <= *cap) return XGM_OK;
    if (*p) HIP_TRY(hipFree(*p));
    *p = nullptr;
    size_t n = std::max(need + need / 2, *cap * 2);        /* headroom: hipFree synchronises the device, a regrow per batch-size wobble costs ms */
    HIP_TRY(hipMalloc((void**)p, n * sizeof(T)));
    *cap = n;
    return XGM_OK;
}

static int grow_pinned(void** p, size_t* cap, size_t need) {
    if (need <= *cap) return XGM_OK;
    if (*p) HIP_TRY(hipHostFree(*p));
    *p = nullptr;
    size_t n = std::max(need + need / 2, *cap * 2);
    HIP_TRY(hipHostMalloc(p, n, hipHostMallocDefault));
    *cap = n;
    return XGM_OK;
}

/* Scratch (per-call device/pinned buffers + a stream) comes from a small pool.  A scratch handed to
 * an asynchronous call (xgm_search_batch_device) is still in use by the enqueued kernels when it
 * returns to the pool ("pending"); the next call takes a FREE one, or creates another (up to
 * XGM_MAX_SCRATCH), so the host plans batch i+1 while the GPU runs batch i.  Only when every scratch is
 * busy does a call wait — for the oldest. */
#define XGM_MAX_SCRATCH 8u

static int scratch_acquire(xgm_index* idx, XgmScratch** out) {
    *out = nullptr;
    XgmScratch* wait_for = nullptr;
    {
        std::lock_guard<std::mutex> lk(idx->scratch_mu);
        std::vector<XgmScratch*>& pool = idx->scratch_pool;
        for (size_t i = pool.size(); i-- > 0;) {
            XgmScratch* c = pool[i];
            if (!c->pending || hipEventQuery(c->ev_done) == hipSuccess) {
                c->pending = false;
                c->inorder = false;
                pool.erase(pool.begin() + (ptrdiff_t)i);
                *out = c;
                break;
            }
        }
        /* an index bound to a caller's stream has ONE asynchronous producer: three scratches in flight keep the host
         * a batch ahead of the GPU; more would only be created (and allocated) to sit in the queue */
        const uint32_t depth = idx->stream ? 3u : XGM_MAX_SCRATCH;
        if (!*out && !pool.empty() && idx->scratch_total >= depth) {
            wait_for = pool.front();
            pool.erase(pool.begin());
        }
        if (!*out && !wait_for) ++idx->scratch_total;
    }
    if (*out) return XGM_OK;
    if (wait_for) {
        hipEventSynchronize(wait_for->ev_done);
        wait_for->pending = false;
        wait_for->inorder = false;
        *out = wait_for;
        return XGM_OK;
    }
    XgmScratch* s = new XgmScratch();
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete s;
        std::lock_guard<std::mutex> lk(idx->scratch_mu);
        --idx->scratch_total;
        return xgm_launch_error("hipStreamCreate", (int)e, hipGetErrorString(e));
    }
    hipEventCreate(&s->ev0);
    hipEventCreate(&s->ev1);
    hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming);
    *out = s;
    return XGM_OK;
}

static void scratch_release(xgm_index* idx, XgmScratch* s) {
    std::lock_guard<std::mutex> lk(idx->scratch_mu);
    idx->scratch_pool.push_back(s);
}

static void scratch_destroy(XgmScratch* s) {
    if (!s) return;
    hipFree(s->d_cand); hipFree(s->d_ghdr); hipFree(s->d_in); hipFree(s->d_mkq); hipFree(s->d_hist);
    hipFree(s->d_arena); hipFree(s->d_count);
    hipFree(s->d_hits); hipFree(s->d_hdrs); hipFree(s->d_part_hits); hipFree(s->d_part_hdrs); hipFree(s->d_arrive); hipFree(s->d_sorted); hipFree(s->d_all);
    if (s->h_sorted) hipHostFree(s->h_sorted);
    if (s->h_up) hipHostFree(s->h_up);
    if (s->h_down) hipHostFree(s->h_down);
    if (s->h_work) hipHostFree(s->h_work);
    if (s->ev0) hipEventDestroy(s->ev0);
    if (s->ev1) hipEventDestroy(s->ev1);
    if (s->ev_done) hipEventDestroy(s->ev_done);
    if (s->stream) hipStreamDestroy(s->stream);
    delete s;
}

/* ------------------------------------------------------------------ index lifetime ----------- */

void xgm_shard_ctx_destroy(XgmShardCtx* c);
void xgm_batcher_destroy(xgm_index* idx);

static void fill_view(xgm_index* idx) {
    xgm_seg_dev& v = idx->view;
    v.doclen = (const uint32_t*)idx->d_sections[XGM_S_DOCLEN];
    v.term_blk = (const uint64_t*)idx->d_sections[XGM_S_TERM_BLK];
    v.term_word = (const uint64_t*)idx->d_sections[XGM_S_TERM_WORD];
    v.term_pos = (const uint64_t*)idx->d_sections[XGM_S_TERM_POS];
    v.blk_first = (const uint32_t*)idx->d_sections[XGM_S_BLK_FIRST];
    v.blk_meta = (const uint32_t*)idx->d_sections[XGM_S_BLK_META];
    v.blk_word = (const uint32_t*)idx->d_sections[XGM_S_BLK_WORD];
    v.blk_pos = (const uint32_t*)idx->d_sections[XGM_S_BLK_POS];
    v.words = (const uint32_t*)idx->d_sections[XGM_S_WORDS];
    v.positions = (const unsigned char*)idx->d_sections[XGM_S_POSITIONS];
    v.term_flags = (const uint32_t*)idx->d_sections[XGM_S_TERM_FLAGS];
    v.stripe_bits = idx->hdr.stripe_bits;
    v.lastdocid = idx->hdr.lastdocid;
    v.dense_id = nullptr; v.dense_dir = nullptr; v.dense_data = nullptr; v.n_dense = 0; v.dense_pos = 0; v.dense_plane = 0;
    v.doclen_narrow = nullptr; v.doclen_narrow_bits = 0; v.doclen_base = 0;
    v.flat_off = nullptr; v.flat_did = nullptr; v.flat_wdf = nullptr; v.flat_pos = nullptr;
    v.n_stripes = (idx->hdr.lastdocid >> idx->hdr.stripe_bits) + 1u;
}

/* Host copies of the dictionary arrays from a host blob. */
static void adopt_host_dictionary(xgm_index* idx, const XgmSegmentBlob& blob) {
    const xgm_seg_header* h = blob.header();
    const uint32_t T = h->n_terms;
    idx->term_df.assign(blob.section<uint32_t>(XGM_S_TERM_DF), blob.section<uint32_t>(XGM_S_TERM_DF) + T);
    idx->term_cf.assign(blob.section<uint32_t>(XGM_S_TERM_CF), blob.section<uint32_t>(XGM_S_TERM_CF) + T);
    idx->term_wdfub.assign(blob.section<uint32_t>(XGM_S_TERM_WDFUB), blob.section<uint32_t>(XGM_S_TERM_WDFUB) + T);
    idx->term_flags.assign(blob.section<uint32_t>(XGM_S_TERM_FLAGS), blob.section<uint32_t>(XGM_S_TERM_FLAGS) + T);
    idx->term_blk.assign(blob.section<uint64_t>(XGM_S_TERM_BLK), blob.section<uint64_t>(XGM_S_TERM_BLK) + T + 1);
    idx->term_word.assign(blob.section<uint64_t>(XGM_S_TERM_WORD), blob.section<uint64_t>(XGM_S_TERM_WORD) + T + 1);
    idx->str_off.assign(blob.section<uint64_t>(XGM_S_STR_OFF), blob.section<uint64_t>(XGM_S_STR_OFF) + T + 1);
    const char* sb = blob.section<char>(XGM_S_STR_BYTES);
    idx->str_bytes.assign(sb, sb + h->sec_bytes[XGM_S_STR_BYTES]);
}

extern "C" int xgm_index_open(const char* segment_path, int device, uint64_t revision, xgm_index** out) {
    if (!segment_path || !out) return xgm_set_error(XGM_E_INVALID, "null argument");
    *out = nullptr;
    XgmSegmentBlob blob;
    int rc = xgm_read_segment_file(segment_path, &blob);
    if (rc) return rc;
    const xgm_seg_header* h = blob.header();
    if (revision != UINT64_MAX && revision != h->revision)
        return xgm_set_error(XGM_E_REVISION, "segment revision %llu != requested %llu", (unsigned long long)h->revision,
                             (unsigned long long)revision);
    if (h->n_blocks >= 0xFFFFFFFFull) return xgm_set_error(XGM_E_INVALID, "segment has too many blocks");
    if (device == XGM_DEVICE_NONE) {
        /* dictionary and statistics only: lookups and query planning work, every search fails with
         * XGM_E_NO_DEVICE (there is no CPU search path) */
        xgm_index* hidx = new xgm_index();
        hidx->device = XGM_DEVICE_NONE;
        hidx->hdr = *h;
        adopt_host_dictionary(hidx, blob);
        *out = hidx;
        return XGM_OK;
    }
    rc = use_device(device);
    if (rc) return rc;
    xgm_index* idx = new xgm_index();
    idx->device = device;
    idx->hdr = *h;
    adopt_host_dictionary(idx, blob);
    hipError_t e = hipMalloc(&idx->d_blob, h->file_bytes);
    if (e != hipSuccess) { delete idx; return xgm_launch_error("hipMalloc(segment)", (int)e, hipGetErrorString(e)); }
    e = hipMemcpy(idx->d_blob, blob.bytes.data(), h->file_bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(idx->d_blob); delete idx; return xgm_launch_error("hipMemcpy(segment)", (int)e, hipGetErrorString(e)); }
    idx->device_bytes = h->file_bytes;
    for (int s = 0; s < XGM_S_COUNT; ++s) idx->d_sections[s] = (char*)idx->d_blob + h->sec_off[s];
    fill_view(idx);
    if ((rc = xgm_build_dense(idx))) { xgm_index_close(idx); return rc; }
    *out = idx;
    return XGM_OK;
}

extern "C" void xgm_index_close(xgm_index* idx) {
    if (!idx) return;
    if (idx->device == XGM_DEVICE_NONE) { delete idx; return; }
    hipSetDevice(idx->device);
    hipDeviceSynchronize();
    xgm_batcher_destroy(idx);
    if (idx->shard_ctx) { xgm_shard_ctx_destroy(idx->shard_ctx); idx->shard_ctx = nullptr; hipSetDevice(idx->device); }
    for (XgmShardCtx* rc_ : idx->retired_shard_ctx) xgm_shard_ctx_destroy(rc_);
    idx->retired_shard_ctx.clear();
    for (XgmScratch* s : idx->scratch_pool) scratch_destroy(s);
    for (auto& pr : idx->prof_events) { hipEventDestroy((hipEvent_t)pr.first); hipEventDestroy((hipEvent_t)pr.second); }
    if (idx->sections_owned) {
        for (int s = 0; s < XGM_S_COUNT; ++s) if (idx->d_sections[s]) hipFree(idx->d_sections[s]);
    }
    if (idx->d_blob) hipFree(idx->d_blob);
    if (idx->d_dense_id) hipFree(idx->d_dense_id);
    if (idx->d_doclen_narrow) hipFree(idx->d_doclen_narrow);
    if (idx->d_flat_off) hipFree(idx->d_flat_off);
    if (idx->d_flat_did) hipFree(idx->d_flat_did);
    if (idx->d_flat_wdf) hipFree(idx->d_flat_wdf);
    if (idx->d_flat_pos) hipFree(idx->d_flat_pos);
    if (idx->d_dense_dir) hipFree(idx->d_dense_dir);
    if (idx->d_dense_data) hipFree(idx->d_dense_data);
    for (auto& c : idx->columns) if (c.second.first) hipFree(c.second.first);
    for (void* c : idx->retired_columns) hipFree(c);
    delete idx;
}

extern "C" int xgm_index_save(const xgm_index* idx, const char* path) {
    if (!idx || !path) return xgm_set_error(XGM_E_INVALID, "null argument");
    int rc = use_device(idx->device);
    if (rc) return rc;
    /* recompute the section table for a contiguous file */
    xgm_seg_header h = idx->hdr;
    uint64_t off = (sizeof h + 255) / 256 * 256;
    for (int s = 0; s < XGM_S_COUNT; ++s) { h.sec_off[s] = off; off = (off + h.sec_bytes[s] + 255) / 256 * 256; }
    h.file_bytes = off;
    std::vector<uint8_t> bytes(off, 0);
    memcpy(bytes.data(), &h, sizeof h);
    for (int s = 0; s < XGM_S_COUNT; ++s) {
        if (!h.sec_bytes[s]) continue;
        if (s == XGM_S_STR_OFF) memcpy(bytes.data() + h.sec_off[s], idx->str_off.data(), h.sec_bytes[s]);
        else if (s == XGM_S_STR_BYTES) memcpy(bytes.data() + h.sec_off[s], idx->str_bytes.data(), h.sec_bytes[s]);
        else HIP_TRY(hipMemcpy(bytes.data() + h.sec_off[s], idx->d_sections[s], h.sec_bytes[s], hipMemcpyDeviceToHost));
    }
    FILE* f = fopen(path, "wb");
    if (!f) return xgm_set_error(XGM_E_IO, "cannot create %s", path);
    size_t w = fwrite(bytes.data(), 1, bytes.size(), f);
    if (fclose(f) != 0 || w != bytes.size()) return xgm_set_error(XGM_E_IO, "short write on %s", path);
    return XGM_OK;
}

extern "C" int xgm_index_get_info(const xgm_index* idx, xgm_index_info* out) {
    if (!idx || !out) return xgm_set_error(XGM_E_INVALID, "null argument");
    const xgm_seg_header& h = idx->hdr;
    out->n_terms = h.n_terms; out->lastdocid = h.lastdocid; out->doccount = h.doccount; out->has_positions = h.has_positions;
    out->total_length = h.total_length; out->revision = h.revision; out->n_postings = h.n_postings;
    out->n_positions = h.n_positions; out->n_blocks = h.n_blocks;
    out->device_bytes = idx->device_bytes;
    out->payload_bytes = h.n_words * 4;
    out->stripe_bits = h.stripe_bits; out->block_size = h.block_size;
    out->doclen_lower_bound = h.doclen_lower_bound; out->wdf_upper_bound = h.wdf_upper_bound;
    return XGM_OK;
}

extern "C" int xgm_index_set_stream(xgm_index* idx, void* hip_stream) {
    if (!idx) return xgm_set_error(XGM_E_INVALID, "null argument");
    idx->stream = hip_stream;
    idx->own_stream = false;
    return XGM_OK;
}

/* reference src/xapian/matcher/nearpostlist.cc:106-140: the duplicate-position step of NearPostList::test_doc */
extern "C" int xgm_index_set_near_colocated(xgm_index* idx, int may_exist) {
    if (!idx) return xgm_set_error(XGM_E_INVALID, "null argument");
    idx->near_colocated.store(may_exist != 0);
    return XGM_OK;
}

extern "C" int xgm_index_set_profiling(xgm_index* idx, int on) {
    if (!idx) return xgm_set_error(XGM_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(idx->scratch_mu);
    idx->profiling = (on & 1) != 0;
    idx->tally = (on & 2) != 0;
    idx->prof_used = 0;
    return XGM_OK;
}

/* Mean duration of the match kernel over the launches recorded since profiling was switched on (or
 * since the previous collection); waits for the last of them, then starts a new window. */
extern "C" double xgm_last_kernel_ms(const xgm_index* cidx) {
    xgm_index* idx = const_cast<xgm_index*>(cidx);
    if (!idx) return -1.0;
    std::lock_guard<std::mutex> lk(idx->scratch_mu);
    if (idx->prof_used == 0) return -1.0;
    double sum = 0.0;
    size_t n = 0;
    for (size_t i = 0; i < idx->prof_used; ++i) {
        hipEvent_t a = (hipEvent_t)idx->prof_events[i].first, b = (hipEvent_t)idx->prof_events[i].second;
        float ms = 0.f;
        if (hipEventSynchronize(b) == hipSuccess && hipEventElapsedTime(&ms, a, b) == hipSuccess) { sum += ms; ++n; }
    }
    idx->prof_used = 0;
    return n ? sum / (double)n : -1.0;
}

extern "C" const char* xgm_last_kernel_name(const xgm_index* idx) { return idx ? idx->last_kernel : ""; }

/* ------------------------------------------------------------------ dictionary --------------- */

/* Dictionary lookup: an open-addressing hash over the sorted term table (built once, lazily and thread-safely),
 * ~2 cache misses per term instead of the ~40 of a binary search over a million strings — the lookups are part of
 * every get_mset (reference: one B-tree descent per term, glass_postlist.cc:151-192). */
static inline uint64_t term_hash(const char* p, size_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xFF51AFD7ED558CCDull);
    while (n >= 8) { uint64_t v; memcpy(&v, p, 8); h = (h ^ v) * 0xD6E8FEB86659FD93ull; h ^= h >> 32; p += 8; n -= 8; }
    uint64_t v = 0;
    memcpy(&v, p, n);
    h = (h ^ v) * 0xD6E8FEB86659FD93ull;
    return h ^ (h >> 29);
}

static void build_term_hash(xgm_index* idx) {
    const uint32_t T = idx->hdr.n_terms;
    uint32_t cap = 16;
    while (cap < 2u * T + 2u) cap <<= 1;
    idx->term_hash.assign(cap, UINT32_MAX);
    const uint64_t* so = idx->str_off.data();
    const char* sb = idx->str_bytes.data();
    for (uint32_t t = 0; t < T; ++t) {
        uint32_t i = (uint32_t)term_hash(sb + so[t], (size_t)(so[t + 1] - so[t])) & (cap - 1u);
        while (idx->term_hash[i] != UINT32_MAX) i = (i + 1u) & (cap - 1u);
        idx->term_hash[i] = t;
    }
}

int xgm_lookup_term_id(const xgm_index* cidx, const char* term, size_t len, uint32_t* id) {
    xgm_index* idx = const_cast<xgm_index*>(cidx);
    std::call_once(idx->term_hash_once, build_term_hash, idx);
    const uint64_t* so = idx->str_off.data();
    const char* sb = idx->str_bytes.data();
    const uint32_t mask = (uint32_t)idx->term_hash.size() - 1u;
    for (uint32_t i = (uint32_t)term_hash(term, len) & mask;; i = (i + 1u) & mask) {
        const uint32_t t = idx->term_hash[i];
        if (t == UINT32_MAX) break;
        if ((size_t)(so[t + 1] - so[t]) == len && memcmp(sb + so[t], term, len) == 0) { *id = t; return 1; }
    }
    *id = UINT32_MAX;
    return 0;
}

extern "C" int xgm_lookup_term(const xgm_index* idx, const char* term, size_t len, uint32_t* term_id, uint32_t* termfreq,
                               uint32_t* collfreq, uint32_t* wdf_ub) {
    if (!idx || !term) return xgm_set_error(XGM_E_INVALID, "null argument");
    uint32_t id;
    bool found = xgm_lookup_term_id(idx, term, len, &id) != 0;
    if (term_id) *term_id = id;
    if (termfreq) *termfreq = found ? idx->term_df[id] : 0;
    if (collfreq) *collfreq = found ? idx->term_cf[id] : 0;
    if (wdf_ub) *wdf_ub = found ? idx->term_wdfub[id] : 0;
    return XGM_OK;
}

/* ---- OP_WILDCARD: prefix expansion over the host dictionary (reference Context<T>::expand_wildcard, api/queryinternal.cc:246-315,
 * walks db.open_allterms(prefix) in term order).  Term ids need not follow byte order (the synthetic builder numbers terms by
 * rank): a byte-ordered permutation of the ids is made on first use. */
static void build_term_order(xgm_index* idx) {
    const uint32_t T = idx->hdr.n_terms;
    const uint64_t* so = idx->str_off.data();
    const char* sb = idx->str_bytes.data();
    idx->term_order.resize(T);
    for (uint32_t i = 0; i < T; ++i) idx->term_order[i] = i;
    auto less = [&](uint32_t a, uint32_t b) {
        const size_t la = (size_t)(so[a + 1] - so[a]), lb = (size_t)(so[b + 1] - so[b]);
        const int c = memcmp(sb + so[a], sb + so[b], std::min(la, lb));
        return c != 0 ? c < 0 : la < lb;
    };
    if (!std::is_sorted(idx->term_order.begin(), idx->term_order.end(), less)) std::sort(idx->term_order.begin(), idx->term_order.end(), less);
}

extern "C" int xgm_expand_prefix(const xgm_index* cidx, const char* prefix, size_t len, uint32_t cap, uint32_t* term_ids, uint32_t* n_total) {
    xgm_index* idx = const_cast<xgm_index*>(cidx);
    if (!idx || (!prefix && len) || !n_total || (cap && !term_ids)) return xgm_set_error(XGM_E_INVALID, "null argument");
    std::call_once(idx->term_order_once, build_term_order, idx);
    const uint64_t* so = idx->str_off.data();
    const char* sb = idx->str_bytes.data();
    /* first term >= prefix in byte order; the expansion is the run of terms that start with it */
    auto below = [&](uint32_t t, int) {
        const size_t lt = (size_t)(so[t + 1] - so[t]);
        const int c = memcmp(sb + so[t], prefix, std::min(lt, len));
        return c != 0 ? c < 0 : lt < len;
    };
    auto it = std::lower_bound(idx->term_order.begin(), idx->term_order.end(), 0, below);
    uint32_t n = 0;
    for (; it != idx->term_order.end(); ++it) {
        const uint32_t t = *it;
        if ((size_t)(so[t + 1] - so[t]) < len || memcmp(sb + so[t], prefix, len) != 0) break;
        if (n < cap) term_ids[n] = t;
        ++n;
    }
    *n_total = n;
    return XGM_OK;
}

extern "C" int xgm_term_info(const xgm_index* idx, uint32_t term_id, const char** bytes, size_t* len, uint32_t* termfreq, uint32_t* collfreq) {
    if (!idx || term_id >= idx->hdr.n_terms) return xgm_set_error(XGM_E_INVALID, "no such term id");
    if (bytes) *bytes = idx->str_bytes.data() + idx->str_off[term_id];
    if (len) *len = (size_t)(idx->str_off[term_id + 1] - idx->str_off[term_id]);
    if (termfreq) *termfreq = idx->term_df[term_id];
    if (collfreq) *collfreq = idx->term_cf[term_id];
    return XGM_OK;
}

extern "C" int xgm_index_termfreqs(const xgm_index* idx, uint32_t* termfreq, uint32_t cap) {
    if (!idx || !termfreq) return xgm_set_error(XGM_E_INVALID, "null argument");
    if (cap < idx->hdr.n_terms) return xgm_set_error(XGM_E_INVALID, "buffer too small");
    memcpy(termfreq, idx->term_df.data(), (size_t)idx->hdr.n_terms * 4);
    return XGM_OK;
}

extern "C" uint64_t xgm_query_postings_bytes(const xgm_index* idx, const xgm_query* q) {
    uint64_t n = 0;
    for (uint32_t t = 0; t < q->n_terms; ++t)
        if (q->terms[t].term_id != UINT32_MAX) n += idx->term_df[q->terms[t].term_id];
    return n * 8;
}

/* ------------------------------------------------------------------ search ------------------- */

static uint32_t next_pow2(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

/* A first guess of a disjunction's final k-th weight (xgm_dev_query::theta_seed).  A document that indexes exactly the terms S weighs at
 * least wlow(S) = Σ_{t in S} weight(wdf = 1, the longest document) — BM25 is monotone in both.  With independent terms N · Π p_t · Π (1 - p_t)
 * documents match exactly S; the guess is the largest wlow(S) such that the subsets at or above it are expected to hold a few times k
 * documents.  It only steers which documents are weighed first: xgm_orw_kernel weighs everything whose BOUND reaches the guess and, should
 * fewer than k of those reach it (correlated terms, tiny shards), goes round again below it. */
static double or_theta_seed(const xgm_index* idx, const xgm_query* q, const xgm_dev_query* d) {
    struct Tm { double p, wlow, ub; };
    Tm tm[XGM_MAX_TERMS];
    uint32_t n = 0;
    const double N = (double)idx->hdr.doccount;
    const double nl_ub = std::max((double)idx->hdr.doclen_upper_bound * q->len_factor, q->min_normlen);
    const double denom_max = q->k1 * (nl_ub * q->b + (1.0 - q->b));
    for (uint32_t t = 0; t < q->n_terms; ++t) {
        const uint32_t id = q->terms[t].term_id;
        if (id == UINT32_MAX || !(q->terms[t].termweight > 0.0)) continue;
        tm[n++] = Tm{std::min(1.0, (double)idx->term_df[id] / N), q->terms[t].termweight * (1.0 / (denom_max + 1.0)), d->ub[t]};
    }
    if (n == 0) return 0.0;
    std::sort(tm, tm + n, [](const Tm& a, const Tm& b) { return a.ub > b.ub; });
    if (n > 8u) n = 8u;                     /* the eight terms with the largest bounds; whatever else a document indexes only adds weight */
    std::pair<double, double> sub[256];     /* (wlow, expected documents) */
    const uint32_t ns = 1u << n;
    for (uint32_t S = 1; S < ns; ++S) {
        double p = N, w = 0.0;
        for (uint32_t i = 0; i < n; ++i) { if ((S >> i) & 1u) { p *= tm[i].p; w += tm[i].wlow; } else p *= 1.0 - tm[i].p; }
        sub[S - 1] = std::make_pair(w, p);
    }
    std::sort(sub, sub + (ns - 1u), [](const std::pair<double, double>& a, const std::pair<double, double>& b) { return a.first > b.first; });
    const double need = 3.0 * (double)d->k;
    double acc = 0.0;
    for (uint32_t i = 0; i + 1u < ns; ++i) { acc += sub[i].second; if (acc >= need) return sub[i].first; }
    return 0.0;
}

/* xgm_query → device form; returns the wdf table width the query needs (1 or 2 bytes), 0 if too big */
static int to_dev_query(const xgm_index* idx, const xgm_query* q, xgm_dev_query* d) {
    memset(d, 0, sizeof *d);
    d->op = q->op == XGM_OP_FILTER ? XGM_OP_AND : q->op == XGM_OP_NEAR ? XGM_OP_PHRASE : q->op;   /* a FILTER is a conjunction some of whose leaves weigh
                                                                                                      nothing; NEAR = PHRASE with another predicate */
    d->n_terms = q->n_terms;
    d->k = q->first + q->maxitems;
    d->window = q->window;
    d->len_factor = q->len_factor; d->k1 = q->k1; d->b = q->b; d->min_normlen = q->min_normlen;
    int width = 1;
    bool any_absent = false, all_absent = true;
    for (uint32_t t = 0; t < q->n_terms; ++t) {
        d->term_id[t] = q->terms[t].term_id;
        d->termweight[t] = q->terms[t].termweight;
        d->phrase_index[t] = (uint8_t)q->terms[t].phrase_index;
        if (q->terms[t].term_id == UINT32_MAX) { if (q->op == XGM_OP_OR || ((q->req_mask >> t) & 1u)) any_absent = true; continue; }
        all_absent = false;
        if (q->terms[t].term_id >= idx->hdr.n_terms) return -1;
        uint32_t ub = idx->term_wdfub[q->terms[t].term_id];
        if (ub > 65534u) return 0;
        if (ub > 254u) width = 2;
        /* leaf weight <= termweight * wdf_ub / (k1 * (min_normlen * b + 1 - b) + wdf_ub) <= termweight:
         * every factor of BM25Weight::get_sumpart is monotone in wdf and in the normalised length */
        double bound = q->terms[t].termweight, bound1 = bound;
        const double nl_lb = std::max((double)idx->hdr.doclen_lower_bound * q->len_factor, q->min_normlen);     /* normlen of any document is at least this */
        const double denom_min = q->k1 * (nl_lb * q->b + (1.0 - q->b));
        const uint32_t wmax = idx->term_wdfmax.empty() ? ub : std::min(ub, idx->term_wdfmax[q->terms[t].term_id]);   /* the true largest wdf where known */
        if (wmax > 0 && denom_min > 0) { bound = q->terms[t].termweight * ((double)wmax / (denom_min + (double)wmax)); bound1 = q->terms[t].termweight * (1.0 / (denom_min + 1.0)); }
        d->ub[t] = bound * 1.000000001;
        d->ub1[t] = std::min(bound1, bound) * 1.000000001;
    }
    if (q->op == XGM_OP_OR && d->k > 0 && idx->hdr.doccount > 0) {
        /* A/B switch (variant tests): scale the guess — far too high forces the kernel's second pass everywhere, 0 switches it off */
        static const double seed_scale = getenv("XGM_OR_SEED_SCALE") ? atof(getenv("XGM_OR_SEED_SCALE")) : 1.0;
        d->theta_seed = or_theta_seed(idx, q, d) * seed_scale;
    }
    if ((q->op == XGM_OP_PHRASE || q->op == XGM_OP_NEAR) && q->phrase_active) {
        d->flags |= XGM_QF_PHRASE;
        /* Enquire::get_mset's check_at_least (enquire.cc:419-426): beyond the page it asks for more documents to be LOOKED AT so that
         * the match count is accurate; within it (Xapiand passes 0) the matcher may stop caring about documents that cannot rank */
        static const bool no_pos_prune = getenv("XGM_NO_POS_PRUNE") != nullptr;         /* A/B switch for measurements */
        if (!no_pos_prune && q->check_at_least <= q->first + q->maxitems) d->flags |= XGM_QF_POSPRUNE;
        if (q->replay & XGM_REPLAY_BATCH_COUNT) d->flags |= XGM_QF_COUNT_ALL;       /* (read by xgm_andw_list_kernel's units only) */
        if (q->op == XGM_OP_NEAR) d->flags |= XGM_QF_NEAR | (idx->near_colocated.load(std::memory_order_relaxed) ? XGM_QF_NEAR_COLOC : 0u);
        else if (q->window == q->n_terms) d->flags |= XGM_QF_EXACT;
    }
    if (q->op == XGM_OP_TREE) {
        /* a nested query: groups + binary nodes straight from the plan; the match kernel evaluates them per document */
        if (q->n_groups == 0 || q->n_groups > XGM_MAX_TERMS || q->tree_len > XGM_MAX_TREE || q->tree_root >= q->n_groups + q->tree_len) return -1;
        d->flags |= XGM_QF_TREE;
        d->tree_len = q->tree_len; d->n_groups = q->n_groups; d->tree_root = q->tree_root; d->group_scored = q->group_scored;
        for (uint32_t t = 0; t < q->n_terms; ++t) { if (q->group_of[t] >= q->n_groups) return -1; d->group_of[t] = q->group_of[t]; }
        for (uint32_t g = 0; g < q->n_groups; ++g) d->termweight[g] = q->group_weight[g];
        for (uint32_t j = 0; j < q->tree_len; ++j) {
            if (q->tree_a[j] >= q->n_groups + j || q->tree_b[j] >= q->n_groups + j || q->tree_op[j] < XGM_N_AND || q->tree_op[j] > XGM_N_MAYBE) return -1;
            d->tnode_op[j] = q->tree_op[j]; d->tnode_a[j] = q->tree_a[j]; d->tnode_b[j] = q->tree_b[j];
        }
        d->score_mask = 0; d->req_mask = 0; d->neg_mask = 0; d->n_req = 0;
        return width;
    }
    if (q->op == XGM_OP_OR ? all_absent : any_absent) d->flags |= XGM_QF_EMPTY;
    /* post-order program → node list */
    int stack[2 * XGM_MAX_TERMS];
    int sp = 0, n_nodes = 0;
    if (q->sum_len == 0 || q->sum_len > 2 * q->n_terms - 1 || (q->sum_len & 1u) == 0) return -1;
    for (uint32_t i = 0; i < q->sum_len; ++i) {
        int8_t op = q->sum_prog[i];
        if (op >= 0) {
            if ((uint32_t)op >= q->n_terms) return -1;
            if (q->terms[op].termweight != 0.0) d->score_mask |= 1u << op;   /* a real term's BM25 weight is > 0; 0 marks an unweighted (FILTER) leaf */
            stack[sp++] = op;
        } else {
            if (sp < 2) return -1;
            int r = stack[--sp], l = stack[--sp];
            d->node_a[n_nodes] = (uint8_t)l;
            d->node_b[n_nodes] = (uint8_t)r;
            stack[sp++] = (int)q->n_terms + n_nodes;
            ++n_nodes;
        }
    }
    if (sp != 1 || n_nodes != ((int)q->sum_len - 1) / 2) return -1;
    d->n_nodes = (uint32_t)n_nodes;
    d->sum_root = (uint32_t)stack[0];
    d->req_mask = q->req_mask;
    d->neg_mask = q->neg_mask;
    d->n_req = (uint32_t)__builtin_popcount(q->req_mask);
    if (q->op != XGM_OP_OR && (q->req_mask == 0 || (q->req_mask >> q->n_terms) != 0 || (q->req_mask & q->neg_mask))) return -1;
    {
        int slot_of[2 * XGM_MAX_TERMS];
        for (uint32_t t = 0; t < q->n_terms; ++t) slot_of[t] = (int)t;
        for (int j = 0; j < n_nodes; ++j) {
            d->ip_a[j] = (uint8_t)slot_of[d->node_a[j]];
            d->ip_b[j] = (uint8_t)slot_of[d->node_b[j]];
            slot_of[q->n_terms + j] = slot_of[d->node_a[j]];
        }
        d->ip_root = (uint32_t)slot_of[stack[0]];
    }
    return width;
}

/* Diagnostics: nanoseconds the host spent per section of a batch call since the last fetch
 * (xgm_debug_host_ns): [0] xgm_plan_query of the descriptions, [1] plan_batch (device queries, cost model, work
 * list), [2] staging + enqueue (copies, events, launches), [3] batches. */
static std::atomic<uint64_t> g_host_ns[8];           /* [4..7]: staging memcpys, H2D enqueue + events, match launch, merge launch */
static inline uint64_t now_ns() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static std::vector<xgm_work> g_last_work;          /* diagnostics only */
static xgm_group_hdr* g_last_ghdr = nullptr;

struct BatchPlan {
    uint32_t nq, k_max, tab_terms, n_work, stripes_per_group, cap, k_stride_c, merge_cap;
    std::vector<xgm_work> work;        /* heaviest first */
    std::vector<uint32_t> goff;        /* [nq+1] first output slot of each query */
    bool phrase, wide;
    bool and_only;      /* every query is a plain conjunction of >= 2 terms → xgm_and_kernel */
    bool andw;          /* ... and k is small: the wave-autonomous variant (one wave per unit) */
    int sided;          /* andw batch with right-hand terms: 1 = AND_NOT only, 2 = AND_MAYBE too */
    bool orw;           /* every query is a plain disjunction → xgm_orw_kernel (one wave per unit) */
    bool fused = false; /* the conjunction kernel finishes its queries itself (xgm_unit_finish.h): no merge launch, no parts */
    uint32_t parts = 1; /* > 1: a query's units are merged in `parts` groups (pseudo-query p * nq + q of goff) and the groups' lists once more */
    uint32_t sub_bits = 0; /* workgroup kernels: every stripe in 2^sub_bits passes over narrower tables (positional queries of > 3 terms) */
    int orw_planes = 6;    /* orw batch: planes of xgm_orw_kernel's bound sum (4 where every query has 4-8 terms) */
    int orw2 = 0;          /* orw batch whose every query xgm_orw2_kernel takes: 1 = containers only, 2 = with flat-array terms */
    bool or_flat = false;  /* orw batch: every term without a container has a flat posting array (xgm_orw_kernel's FLAT instantiation) */
    /* (a plan that is kept between calls — run_class_batch's, per thread: a work list of tens of thousands of units is a megabyte that would otherwise be
     *  allocated, faulted in and released per batch) */
    void reset() { work.clear(); goff.clear(); fused = false; parts = 1; sub_bits = 0; orw2 = 0; orw_planes = 6; or_flat = false; }
};

static int dense_kind(const xgm_index* idx, const xgm_query& q, bool fused = false);
static bool flat_kind(const xgm_index* idx, const xgm_query& q);
static int or2_kind(const xgm_index* idx, const xgm_query& q);

static int plan_batch(const xgm_index* idx, const xgm_query* qs, uint32_t nq, xgm_dev_query* dq, uint32_t* kq, double* maxposs,
                      BatchPlan* bp, bool force_general = false, int mode = 0) {
    const bool list = mode == 1;                     /* xgm_andw_list_kernel's launch; mode 2: xgm_andw_all_kernel's (both: one part, no last-unit merge) */
    bp->nq = nq; bp->k_max = 1; bp->tab_terms = 1; bp->phrase = false; bp->wide = false; bp->and_only = true;
    bool or_only = getenv("XGM_NO_ORW") == nullptr;                                  /* A/B switch for measurements */
    bool conj_only = true;       /* every query: AND / PHRASE of >= 2 terms (positional filter or not) */
    bool andnot_ok = true;       /* every query: AND, AND_NOT or AND_MAYBE whose required terms are plan positions [0, n_req) */
    static const bool no_and_kernel = getenv("XGM_NO_AND_KERNEL") != nullptr;      /* A/B switch for measurements */
    if (no_and_kernel) bp->and_only = false;
    if (force_general) { bp->and_only = false; or_only = false; conj_only = false; andnot_ok = false; }   /* the workgroup kernel's decomposition (xgm_search_sorted) */
    for (uint32_t i = 0; i < nq; ++i) {
        if (qs[i].n_terms == 0 || qs[i].n_terms > XGM_MAX_TERMS) return xgm_set_error(XGM_E_INVALID, "query %u: bad n_terms", i);
        int width = to_dev_query(idx, &qs[i], &dq[i]);
        if (width < 0) return xgm_set_error(XGM_E_INVALID, "query %u: malformed plan", i);
        if (width == 0) return XGM_UNSUPPORTED;
        if (dq[i].k > XGM_MAX_K) return XGM_UNSUPPORTED;
        if (width == 2) bp->wide = true;
        if (dq[i].op != XGM_OP_AND || dq[i].n_terms < 2 || (dq[i].flags & XGM_QF_PHRASE)) bp->and_only = false;
        if (dq[i].op != XGM_OP_OR) or_only = false;
        if ((dq[i].op != XGM_OP_AND && dq[i].op != XGM_OP_AND_NOT && dq[i].op != XGM_OP_AND_MAYBE) || dq[i].n_terms < 2 ||
            (dq[i].flags & XGM_QF_PHRASE) || dq[i].req_mask != (dq[i].n_req >= 32u ? 0xFFFFFFFFu : (1u << dq[i].n_req) - 1u) ||
            (dq[i].op != XGM_OP_AND && dq[i].n_terms > 8u) ||                       /* the wave kernel sums <= 8 leaves in registers */
            (dq[i].op == XGM_OP_AND_NOT && (dq[i].req_mask | dq[i].neg_mask) != (1u << dq[i].n_terms) - 1u))
            andnot_ok = false;
        if ((dq[i].op != XGM_OP_AND && dq[i].op != XGM_OP_PHRASE) || dq[i].n_terms < 2) conj_only = false;
        if (dq[i].flags & XGM_QF_PHRASE) {
            if (dq[i].n_terms > XGM_PHRASE_MAX_TERMS) return XGM_UNSUPPORTED;
            bp->phrase = true;
        }
        bp->k_max = std::max(bp->k_max, dq[i].k);
        bp->tab_terms = std::max(bp->tab_terms, dq[i].n_terms);
        kq[i] = dq[i].k;
        maxposs[i] = qs[i].max_possible;
    }
    if (bp->phrase && bp->tab_terms > XGM_PHRASE_MAX_TERMS) {
        /* a batch mixing long AND/OR queries with phrases would blow the LDS budget: caller splits */
        return XGM_UNSUPPORTED;
    }
    static const bool no_andw = getenv("XGM_NO_ANDW") != nullptr;                    /* A/B switch for measurements */
    const uint32_t n_stripes = (idx->hdr.lastdocid >> idx->hdr.stripe_bits) + 1u;
    const uint32_t k_pad = next_pow2(bp->k_max);
    static const bool no_phrase_w = getenv("XGM_NO_PHRASEW") != nullptr;           /* A/B switch for measurements */
    const bool phrase_conj = conj_only && bp->phrase && !no_phrase_w && !no_and_kernel;
    if (no_and_kernel) andnot_ok = false;
    if (!bp->and_only && bp->tab_terms > 8u) andnot_ok = false;      /* a SIDED launch sums every query's leaves in 8 registers */
    bool any_maybe = false;
    for (uint32_t i = 0; i < nq; ++i) any_maybe = any_maybe || dq[i].op == XGM_OP_AND_MAYBE;
    bp->sided = (andnot_ok && !bp->and_only) ? (any_maybe ? 2 : 1) : 0;
    bp->andw = (bp->and_only || andnot_ok || phrase_conj) && !no_andw && bp->k_max <= 192u && (uint64_t)((n_stripes + 31u) / 32u) * k_pad <= XGM_MERGE_CAP &&
               xgm_andw_smem_bytes(idx->hdr.stripe_bits, bp->tab_terms, std::max(128u, next_pow2(bp->k_max + 64u)), bp->wide, 32u, bp->phrase, bp->sided == 2) <= 160u * 1024u;
    /* disjunctions: one wave per unit as well; a unit spans as many stripes as the merge capacity
     * (units x k candidates per query) requires */
    uint32_t orw_spg = 32u;
    while ((uint64_t)((n_stripes + orw_spg - 1u) / orw_spg) * k_pad > XGM_MERGE_CAP && orw_spg < 4096u) orw_spg *= 2u;
    bp->orw = or_only && (uint64_t)((n_stripes + orw_spg - 1u) / orw_spg) * k_pad <= XGM_MERGE_CAP &&
              xgm_orw_smem_bytes(idx->hdr.stripe_bits, bp->tab_terms, std::max(128u, next_pow2(bp->k_max + 64u)), bp->wide, orw_spg) <= 160u * 1024u;
    static const bool no_or_flat = getenv("XGM_NO_OR_FLAT") != nullptr;                 /* A/B switch (the variant tests): block decode for every term without a container */
    if (bp->orw && !bp->wide && !no_or_flat && idx->view.flat_off && idx->view.n_dense && idx->flat_postings < 0xFFFFFFFFull) {
        bool all = true;
        for (uint32_t i = 0; i < nq && all; ++i)
            for (uint32_t t = 0; t < qs[i].n_terms && all; ++t) {
                const uint32_t id = qs[i].terms[t].term_id;
                if (id == UINT32_MAX || (uint64_t)idx->term_df[id] >= idx->dense_min_df) continue;       /* absent, or it has a container */
                all = idx->term_wdfub[id] <= 254u && idx->term_df[id] != 0;                             /* (what build_flat gives an array) */
            }
        bp->or_flat = all;
    }
    if (bp->orw) {
        /* planes of the bound sum (xgm_or.hip, PL): 4 where every query of the batch has 4-8 terms, else 6 (XGM_ORW_PLANES=4 | 6 forces one: A/B, the variant tests) */
        static const int forced = getenv("XGM_ORW_PLANES") ? atoi(getenv("XGM_ORW_PLANES")) : 0;
        bool four = true;
        for (uint32_t i = 0; i < nq && four; ++i) four = qs[i].n_terms >= 4u && qs[i].n_terms <= 8u;
        bp->orw_planes = forced == 4 || forced == 6 ? forced : (four ? 4 : 6);
    }
    if (bp->orw && !bp->wide && bp->tab_terms <= 8u) {
        int kind = 1;
        for (uint32_t i = 0; i < nq && kind; ++i) { const int kq_ = or2_kind(idx, qs[i]); kind = kq_ == 0 ? 0 : std::max(kind, kq_); }
        if (kind && xgm_orw2_smem_bytes(idx->hdr.stripe_bits, bp->tab_terms, std::max(128u, next_pow2(bp->k_max + 64u)), orw_spg, kind == 2) <= 160u * 1024u) bp->orw2 = kind;
    }
    const bool wave_units = bp->andw || bp->orw;
    /* plain conjunctions over containers only: their units take xgm_dense_unit inside xgm_andw_kernel<uint8_t, false, 0> */
    /* ... and positional queries over containers only that prune by weight: xgm_dense_unit<PHRASE> inside the positional instantiation
     * (XGM_NO_DENSE_PHRASE_BODY: A/B switch, the variant tests) */
    static const bool no_dense_phrase = getenv("XGM_NO_DENSE_PHRASE_BODY") != nullptr;
    if (bp->andw && bp->sided == 0 && !bp->wide)
        for (uint32_t i = 0; i < nq; ++i) {
            const int dk = dense_kind(idx, qs[i], true);
            if ((dk == 1 && !bp->phrase) || (dk == 2 && bp->phrase && !no_dense_phrase && (dq[i].flags & XGM_QF_POSPRUNE))) dq[i].flags |= XGM_QF_DENSE;
            /* ... or led by a long-tail term with a flat posting array: xgm_flat_unit (positional queries: those that prune by weight) */
            else if (!(dq[i].flags & XGM_QF_EMPTY) && flat_kind(idx, qs[i]) && (!bp->phrase || (dq[i].flags & XGM_QF_POSPRUNE)) &&
                     ((qs[i].op == XGM_OP_PHRASE || qs[i].op == XGM_OP_NEAR) ? bp->phrase : !bp->phrase))
                dq[i].flags |= XGM_QF_FLAT;
        }
    /* the bodies live in the first bytes of xgm_andw_kernel's per-wave LDS slice: flagged only where they fit (they do for any batch the
     * wave kernel takes: its positional slice holds its own staging area of tab_terms x 2 KiB) */
    if (bp->andw && bp->sided == 0 && !bp->wide) {
        const size_t slice = xgm_andw_smem_bytes(idx->hdr.stripe_bits, bp->tab_terms, std::max(128u, next_pow2(bp->k_max + 64u)), bp->wide, 32u, bp->phrase, false) / XGM_WAVES;
        for (uint32_t i = 0; i < nq; ++i) {
            if ((dq[i].flags & XGM_QF_FLAT) && xgm_body_wave_bytes(true, bp->phrase, dq[i].n_terms) > slice) dq[i].flags &= ~XGM_QF_FLAT;
            if ((dq[i].flags & XGM_QF_DENSE) && xgm_body_wave_bytes(false, bp->phrase, dq[i].n_terms) > slice) dq[i].flags &= ~XGM_QF_DENSE;
        }
    }
    bp->cap = wave_units ? std::max(128u, next_pow2(bp->k_max + 64u)) : std::max(512u, next_pow2(bp->k_max + XGM_WG));
    /* Work decomposition.  Cost model of a query: the posting blocks its terms own (df/128 full blocks
     * plus about one partial block per stripe a term touches).  Every query is cut into units of
     * about total/(units the chip holds) cost — heavy queries into many — bounded by the LDS run table
     * (8 B per term and stripe → at most spg_max stripes per unit) and by the merge kernel's sort
     * capacity (units × k candidates). */
    uint32_t spg_max = bp->andw ? 32u : bp->orw ? orw_spg : std::max(1u, (16u * 1024u) / (8u * bp->tab_terms));
    if (!wave_units) {
        /* the run table shares the 160 KiB with the tables (PHRASE position tables are large) */
        /* (the sorted instantiation — force_general — keeps a second key, a collapse ordinal and its control words per top-k entry) */
        const size_t extra = force_general ? (size_t)bp->cap * 12 + 128 : 0;
        size_t base = xgm_match_smem_bytes(idx->hdr.stripe_bits, bp->tab_terms, bp->phrase, bp->cap, bp->wide, 0) + extra;
        /* a positional query of more than XGM_PHRASE_MAX_TERMS_WG terms: 4-byte position starts for every slot of a stripe and term do not
         * fit; the kernel then takes a stripe in 2^sub_bits passes over tables of W >> sub_bits slots (xgm_match_body.inc), leaving room
         * for a run table of at least 16 stripes */
        while (bp->phrase && base + 8u * bp->tab_terms * 16u + 64u > 160u * 1024u && bp->sub_bits < 3u && idx->hdr.stripe_bits - bp->sub_bits > 8u) {
            ++bp->sub_bits;
            base = xgm_match_smem_bytes(idx->hdr.stripe_bits - bp->sub_bits, bp->tab_terms, bp->phrase, bp->cap, bp->wide, 0) + extra;
        }
        if (base + 8u * bp->tab_terms + 64u > 160u * 1024u) return XGM_UNSUPPORTED;
        spg_max = std::min<uint32_t>(spg_max, (uint32_t)((160u * 1024u - 64u - base) / (8u * bp->tab_terms)));
    }
    const uint32_t g_min = (n_stripes + spg_max - 1) / spg_max;
    /* few queries in flight (latency mode): a smaller merge (sort of <= 4096) beats more units */
    const uint32_t merge_budget = (nq <= 4u && !bp->orw) ? XGM_MERGE_CAP / 2u : XGM_MERGE_CAP;   /* (a disjunction's units are long: more of them wins) */
    static const uint32_t units_cap = getenv("XGM_MAX_UNITS_PER_QUERY") ? (uint32_t)atoi(getenv("XGM_MAX_UNITS_PER_QUERY")) : 0u;   /* A/B switch for measurements */
    /* units x k candidates must fit the merge kernel's LDS sort; a unit's window there is k_max entries (not the next power of two) */
    static const bool units_by_kpad = getenv("XGM_UNITS_BY_KPAD") != nullptr;                      /* A/B switch: the round-2 bound */
    /* ... per PART: a query whose units exceed one merge's capacity is merged in up to kMaxParts groups and the groups' lists once more
     * (xgm_merge_parts), so that the heaviest conjunctions — frequent-term phrases whose every document is a candidate — can be cut
     * down to single stripes: the launch ends with its longest unit (measured: C5's kernel ran at 20 % mean occupancy behind them) */
    static const uint32_t kMaxParts = getenv("XGM_MAX_PARTS") ? (uint32_t)std::min(8, std::max(1, atoi(getenv("XGM_MAX_PARTS")))) : 4u;      /* A/B switch (<= 8: the parts' merge sorts parts x k in LDS) */
    const uint32_t units_per_part = std::max(1u, merge_budget / (units_by_kpad ? k_pad : std::max(1u, bp->k_max)));
    const uint32_t parts_ok = (bp->andw && nq > 4u) ? kMaxParts : 1u;
    uint32_t g_max = std::max(g_min, std::min(n_stripes, units_per_part * parts_ok));
    if (units_cap) g_max = std::max(g_min, std::min(g_max, units_cap));
    if ((uint64_t)g_min * k_pad > XGM_MERGE_CAP) return XGM_UNSUPPORTED;
    /* Cost model of a query (unit: ~1k cycles of one wave, measured on MI355X): every
     * active stripe pays a fixed latency chain; each posting block that still has to be decoded (terms
     * without probe containers) adds ~1; each candidate costs a probe + a share of the scoring. */
    std::vector<double> cost(nq), conj_per_stripe(nq, 0.0);
    double total_cost = 0;
    const double n_docs = std::max<double>(1.0, idx->hdr.doccount);
    const double Wd = (double)(1u << idx->hdr.stripe_bits);
    for (uint32_t i = 0; i < nq; ++i) {
        double sparse_blocks = 0, min_df = 1e30, dens = 1.0;
        bool all_dense = bp->andw;
        uint32_t sparse_req = 0;                     /* required terms without containers */
        bool sparse_rhs = false;                     /* ... right-hand terms without them */
        for (uint32_t t = 0; t < qs[i].n_terms; ++t) {
            uint32_t id = qs[i].terms[t].term_id;
            /* the candidates are the conjunction of the REQUIRED terms; right-hand terms (excluded / optional) are only probed */
            const bool required = qs[i].op == XGM_OP_OR || ((qs[i].req_mask >> t) & 1u);
            if (id == UINT32_MAX) { if (required) { min_df = 0; all_dense = false; } continue; }
            const double df = idx->term_df[id];
            const bool dense = bp->andw && !bp->wide && (!bp->phrase || idx->view.dense_pos) && idx->dense_min_df && (uint64_t)idx->term_df[id] >= idx->dense_min_df;
            if (!dense) { sparse_blocks += (required ? 1.0 : 0.25) * (df / XGM_BLOCK + std::min<double>(df, n_stripes)); if (required) all_dense = false; }
            if (!dense) { if (required) ++sparse_req; else sparse_rhs = true; }
            if (!required) continue;
            min_df = std::min(min_df, df);
            dens *= df / n_docs;
        }
        /* AND visits only stripes where the rarest term has postings */
        const double stripes = qs[i].op == XGM_OP_OR ? n_stripes : std::min<double>(n_stripes, min_df);
        const double cand_per_stripe = all_dense ? Wd * dens : (stripes > 0 ? min_df / stripes : 0.0);
        conj_per_stripe[i] = cand_per_stripe;
        /* a candidate of a positional query costs a 64-byte sector per term, its positions and the predicate on top of the probe */
        static const double phrase_cand = getenv("XGM_PHRASE_CAND_COST") ? atof(getenv("XGM_PHRASE_CAND_COST")) : 0.25;
        /* ... and with three or more terms the matches are rarer, the query-wide threshold comes later and more of the candidates are tested
         * (C5's `t70 t11 t5`: 38 matches, 120 k tests: its 41 units ran as long as the whole launch) */
        static const double phrase_t3 = getenv("XGM_PHRASE_T3_COST") ? atof(getenv("XGM_PHRASE_T3_COST")) : 2.0;      /* A/B switch; measured C5 231 / 253 / 248 k queries/s at 1 / 2 / 4 */
        const double per_cand = bp->phrase ? phrase_cand * (qs[i].n_terms >= 3 ? phrase_t3 : 1.0) : 0.03;
        cost[i] = stripes * (bp->andw ? 15.0 : 8.0) + 0.9 * sparse_blocks + per_cand * cand_per_stripe * qs[i].n_terms * stripes + 1.0;
        if (bp->andw && !bp->phrase && min_df > 0 && sparse_req <= 1 && !sparse_rhs && qs[i].n_terms <= 8) {
            /* the conjunction kernel's queue path (measured per unit on MI355X, tools/units.py, in ~1k cycles of a wave
             * at 4 waves per SIMD): the producer costs ~8 per stripe when the candidates are the AND of the bitmaps and
             * ~0.05 per posting when they are the rarest term's postings; a probed candidate ~0.11, a weighed match
             * ~0.15 (more with optional terms to sum) */
            const double cands = sparse_req ? min_df : Wd * dens * n_stripes;
            const double matches = sparse_req ? min_df * dens / std::max(1e-12, min_df / n_docs) : cands;
            cost[i] = (sparse_req ? 4.0 + 0.3 * stripes : 8.0 * stripes) + 0.16 * cands + (bp->sided == 2 ? 0.22 : 0.15) * matches + 1.0;
        } else if (bp->andw && !bp->phrase) {
            cost[i] = stripes * 28.0 + 0.9 * sparse_blocks + per_cand * cand_per_stripe * qs[i].n_terms * stripes + 1.0;
        }
        if (dq[i].flags & XGM_QF_FLAT) {
            /* xgm_flat_unit: rounds of 64 postings of the lead term, whatever stripes they fall in (~3 k cycles a round: two dependent gathers;
             * a binary search per other term that has no containers) + the unit's prologue */
            double flat_others = 0;
            for (uint32_t t = 1; t < qs[i].n_terms; ++t) {
                const uint32_t id = qs[i].terms[t].term_id;
                if (id != UINT32_MAX && !(idx->view.n_dense && (uint64_t)idx->term_df[id] >= idx->dense_min_df)) flat_others += 1.0;
            }
            cost[i] = 2.0 + (double)idx->term_df[qs[i].terms[0].term_id] / 64.0 * (3.0 + 2.0 * flat_others);
        }
        if (bp->orw) {
            /* every stripe: the terms' bitmaps / block decodes (twice where candidates remain) and a
             * share of the union that survives the MaxScore pruning */
            double sum_df = 0, blocks = 0;
            for (uint32_t t = 0; t < qs[i].n_terms; ++t) {
                uint32_t id = qs[i].terms[t].term_id;
                if (id == UINT32_MAX) continue;
                const double df = idx->term_df[id];
                sum_df += df;
                if (bp->wide || (uint64_t)idx->term_df[id] < idx->dense_min_df || idx->dense_min_df == 0) blocks += df / XGM_BLOCK + std::min<double>(df, n_stripes);
            }
            /* (the documents weighed are a small share of the union since the threshold starts at the planner's guess) */
            cost[i] = n_stripes * (10.0 + 3.0 * qs[i].n_terms) + 1.8 * blocks + 0.001 * sum_df + 1.0;
        }
        total_cost += cost[i];
    }
    /* units per launch: ~3 per wave slot of the chip (4 waves per SIMD for the conjunction kernel; the disjunction kernel runs 2 and
     * pays a longer prologue per unit) */
    static const double orw_units = getenv("XGM_ORW_UNITS") ? atof(getenv("XGM_ORW_UNITS")) : 8192.0;      /* A/B switch for measurements */
    static const double and_units = getenv("XGM_TARGET_UNITS") ? atof(getenv("XGM_TARGET_UNITS")) : 12288.0;      /* A/B switch for measurements */
    /* a SMALL batch (a server's natural batches are a few dozen queries, xgm_index_set_batching) keeps the units-per-query ratio of a
     * full one instead of the full unit count: a 24-query batch cut into 12 288 units spends more on its work list (196 KB built, staged,
     * uploaded) and on scheduling 3 072 tiny workgroups than on matching.  XGM_UNITS_PER_QUERY: A/B switch (0 = rounds 1-3). */
    static const double units_per_query = getenv("XGM_UNITS_PER_QUERY") ? atof(getenv("XGM_UNITS_PER_QUERY")) : 48.0;
    static const double units_floor = getenv("XGM_UNITS_FLOOR") ? atof(getenv("XGM_UNITS_FLOOR")) : 3072.0;                     /* A/B switch */
    const double scaled_units = (units_per_query > 0.0 && nq > 4u) ? std::min(and_units, std::max(units_floor, units_per_query * (double)nq)) : and_units;
    /* ... and a disjunction launch of a FEW queries (the class split leaves xgm_orw_kernel the odd query xgm_orw2_kernel does not take) is not cut
     * into 8 192 units of one stripe each: 32 units per query, at least 1 024 */
    const double orw_scaled = nq >= 192u ? orw_units : std::min(orw_units, std::max(1024.0, 32.0 * (double)nq));
    const double target_units = bp->orw ? orw_scaled : scaled_units;
    double unit_cost = std::max(1.0, total_cost / (wave_units ? target_units : 3072.0));
    if (wave_units) {
        /* the floor of g_min units per query (the LDS table bounds a unit's stripes) eats part of the budget: raise the
         * unit cost until the batch fits ~12288 units again (3 per wave slot of the chip) */
        for (int it = 0; it < 8; ++it) {
            double units = 0;
            for (uint32_t i = 0; i < nq; ++i) units += std::min<double>(g_max, std::max<double>(g_min, std::ceil(cost[i] / unit_cost)));
            if (units <= target_units * 1.03) break;
            unit_cost *= std::max(1.02, units / target_units);
        }
    }
    bp->work.clear();
    uint32_t spg_used = 1;
    /* units in descending cost order (longest first); the units of a query share one cost, so ordering the QUERIES
     * (stable) orders the units exactly as a stable sort of all of them would */
    std::vector<uint32_t> gqv(nq), spgv(nq), order(nq), subv(nq, 1u);       /* subv: LIST — units per stripe (1, 2 or 4: parts of a stripe, see xgm_dense_unit) */
    uint32_t g_most_q = 0;
    /* positional queries: what a unit costs depends on how soon it holds k matches (until then every candidate's positions are tested),
     * which the cost model cannot know — a 3-term phrase of frequent terms with few matches ran 2 ms in ONE 30-stripe unit while the
     * rest of the launch took 0.6 ms (tools/qcost.py, round 4, BEFORE the dense body tested survivors from LDS-staged positions).  A/B switch:
     * XGM_PHRASE_UNIT_STRIPES = n bounds a unit of such a query to n stripes (0 = off, the default). */
    static const uint32_t phrase_unit_stripes = getenv("XGM_PHRASE_UNIT_STRIPES") ? (uint32_t)atoi(getenv("XGM_PHRASE_UNIT_STRIPES")) : 0u;   /* (measured with the LDS-staged positional test in place: C5 113.3 k queries/s unbounded, 109.5 k at 8 stripes, 100.9 k at 4: off) */
    for (uint32_t i = 0; i < nq; ++i) {
        uint32_t gq = (uint32_t)std::min<double>(g_max, std::max<double>(g_min, std::ceil(cost[i] / unit_cost)));
        if (bp->phrase && bp->andw && phrase_unit_stripes && (dq[i].flags & XGM_QF_DENSE) && nq > 4u)
            gq = std::max(gq, std::min(g_max, (n_stripes + phrase_unit_stripes - 1u) / phrase_unit_stripes));
        if (list && bp->phrase && (dq[i].flags & XGM_QF_DENSE)) {
            /* a LIST unit tests the positions of EVERY document of the conjunction in its range, in docid order, until the query's first matches are
             * found: the launch ends with the longest such walk (measured, round 6: a 3-stripe unit of `t3 t8 t9` — 7 000 documents of the conjunction,
             * 104 matches in the shard — ran 2.5 M cycles, the whole launch 1.27 ms).  Units of about XGM_LIST_UNIT_DOCS documents: the later ones
             * cost nothing once the earlier ones hold the matches that decide the page (PrefixList::look_back) */
            static const double list_unit_docs = getenv("XGM_LIST_UNIT_DOCS") ? atof(getenv("XGM_LIST_UNIT_DOCS")) : 1024.0;      /* A/B switch */
            const uint32_t spg_l = (uint32_t)std::max(1.0, std::min(32.0, std::floor(list_unit_docs / std::max(1.0, conj_per_stripe[i]))));
            gq = std::max(gq, std::min(n_stripes, (n_stripes + spg_l - 1u) / spg_l));
            /* ... and below one stripe: halves or quarters of it (word-major bitmaps: a component of the lanes' words is a quarter of the stripe) */
            static const bool no_parts = getenv("XGM_LIST_NO_STRIPE_PARTS") != nullptr;      /* A/B switch */
            if (!no_parts && xgm_dense_word_major() && idx->hdr.stripe_bits >= 9u && conj_per_stripe[i] > 1.5 * list_unit_docs && (uint64_t)n_stripes * 4u < (1u << 24))
                subv[i] = conj_per_stripe[i] > 3.0 * list_unit_docs ? 4u : 2u;
        }
        uint32_t spg = (n_stripes + gq - 1) / gq;
        gq = (n_stripes + spg - 1) / spg;
        if (subv[i] > 1u && spg == 1u) gq = n_stripes * subv[i]; else subv[i] = 1u;
        spg_used = std::max(spg_used, spg);
        gqv[i] = gq; spgv[i] = spg; order[i] = i;
        g_most_q = std::max(g_most_q, gq);
    }
    /* parts: units [p * upp, (p + 1) * upp) of query q are pseudo-query p * nq + q of goff */
    /* xgm_andw_kernel merges a query's lists in its last unit, whatever their number (XGM_NO_FUSED_MERGE: A/B switch, the variant tests) */
    static const bool no_fused = getenv("XGM_NO_FUSED_MERGE") != nullptr;
    /* ... xgm_orw_kernel can as well (round 6) — OPT-IN, XGM_OR_FUSED_MERGE=1: measured on the MI355X it LOSES (C3: 135.1 k queries/s, kernel 1.848 ms, against
     * 139.1 k / 1.774 ms with the merge launch): one wave merging 32 lists of 100 candidates lengthens the launch's tail by more than the merge launch and the
     * gap before it cost; the conjunction's k = 10 lists are another matter */
    static const bool or_fused = getenv("XGM_OR_FUSED_MERGE") != nullptr;
    bp->fused = ((bp->andw && !no_fused) || (bp->orw && !bp->orw2 && !no_fused && or_fused)) && mode == 0;                                                    /* (the stand-alone dense kernel, XGM_DENSE_KERNEL=1, finishes its queries the same way) */
    /* (list: xgm_andw_list_kernel's units are walked in stripe order by xgm_frozen_finish_kernel, whatever their number: one part) */
    const uint32_t P = (bp->fused || mode != 0) ? 1u : (g_most_q + units_per_part - 1) / units_per_part;
    bp->parts = std::max(1u, P);
    const uint32_t upp = bp->parts > 1 ? units_per_part : g_most_q + 1u;
    bp->goff.assign((size_t)nq * bp->parts + 1, 0);
    for (uint32_t p = 0; p < bp->parts; ++p)
        for (uint32_t i = 0; i < nq; ++i) {
            const uint32_t lo = std::min(gqv[i], p * upp), hi = std::min(gqv[i], (p + 1) * upp);
            bp->goff[(size_t)p * nq + i + 1] = bp->goff[(size_t)p * nq + i] + (bp->parts > 1 ? hi - lo : gqv[i]);
        }
    /* (round 6, measured and dropped: the units of queries that take the same body of xgm_andw_kernel — dense / flat / queue — next to each other in the
     *  grid, for the instruction cache: 0.3404 vs 0.3396 ms per launch of C2, no difference) */
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] / gqv[a] > cost[b] / gqv[b]; });
    bp->work.reserve(bp->goff.back());
    for (uint32_t oi = 0; oi < nq; ++oi) {
        const uint32_t i = order[oi], gq = gqv[i], spg = spgv[i];
        for (uint32_t g = 0; g < gq; ++g) {
            xgm_work w;
            const uint32_t p = bp->parts > 1 ? g / upp : 0u;
            w.qi = i; w.s_begin = g * spg; w.s_end = std::min(n_stripes, (g + 1) * spg); w.slot = bp->goff[(size_t)p * nq + i] + (g - p * (bp->parts > 1 ? upp : 0u));
            if (subv[i] > 1u) {                                    /* part g % sub of stripe g / sub: its component mask rides in s_end's top byte */
                const uint32_t sub = subv[i], st = g / sub, part = g % sub;
                w.s_begin = st; w.s_end = (st + 1u) | ((sub == 4u ? (1u << part) : (3u << (2u * part))) << 24);
            }
            bp->work.push_back(w);
        }
    }
    /* a LIST launch: stripe order — the units of lower stripes are done (and have published their match counts) when those of higher stripes
     * start, which then have nothing to do for every query whose page is decided (PrefixList::look_back) */
    static const bool list_lpt = getenv("XGM_LIST_LPT_ORDER") != nullptr;            /* A/B switch: heaviest first, as the other launches */
    if (list && !list_lpt) {
        /* (a counting sort on the first stripe, stable: tens of thousands of units per batch — a comparison sort of them was the host's largest item) */
        static thread_local std::vector<uint32_t> at;
        static thread_local std::vector<xgm_work> sorted;
        at.assign((size_t)n_stripes + 1u, 0u);
        for (const xgm_work& w : bp->work) ++at[w.s_begin + 1u];
        for (uint32_t st = 0; st < n_stripes; ++st) at[st + 1u] += at[st];
        sorted.resize(bp->work.size());
        for (const xgm_work& w : bp->work) sorted[at[w.s_begin]++] = w;
        bp->work.swap(sorted);
    }
    bp->n_work = (uint32_t)bp->work.size();
    bp->stripes_per_group = spg_used;
    uint32_t g_most = 0;
    for (size_t i = 0; i + 1 < bp->goff.size(); ++i) g_most = std::max(g_most, bp->goff[i + 1] - bp->goff[i]);
    const size_t smem = bp->andw ? xgm_andw_smem_bytes(idx->hdr.stripe_bits, bp->tab_terms, bp->cap, bp->wide, bp->stripes_per_group, bp->phrase, bp->sided == 2)
                        : bp->orw2 ? xgm_orw2_smem_bytes(idx->hdr.stripe_bits, bp->tab_terms, bp->cap, bp->stripes_per_group, bp->orw2 == 2)
                        : bp->orw ? xgm_orw_smem_bytes(idx->hdr.stripe_bits, bp->tab_terms, bp->cap, bp->wide, bp->stripes_per_group)
                                 : xgm_match_smem_bytes(idx->hdr.stripe_bits - bp->sub_bits, bp->tab_terms, bp->phrase, bp->cap, bp->wide, bp->stripes_per_group);
    if (smem > 160u * 1024u) return XGM_UNSUPPORTED;
    bp->merge_cap = std::max(512u, next_pow2(g_most * bp->k_max));     /* fixed window of k_max per unit */
    bp->k_stride_c = bp->k_max;
    return XGM_OK;
}


/* Runs one batch of ONE kernel class (or a batch whose mix plan_batch resolves by itself); results land in rows
 * rows[i] (i when rows == NULL) of d_hits / d_hdrs (device).  Asynchronous on `stream`. */
constexpr int XGM_LIST_DECLINED = 2;       /* run_class_batch(list): the batch is not one xgm_andw_list_kernel takes — nothing was enqueued */
static int run_class_batch(xgm_index* idx, XgmScratch* s, hipStream_t stream, const xgm_query* qs, uint32_t nq, uint32_t k_stride,
                           xgm_hit* d_hits, xgm_result_hdr* d_hdrs, const uint32_t* rows, int mode = 0, unsigned long long* d_extra = nullptr) {
    int rc;
    const bool list = mode == 1, all = mode == 2;      /* the reference-identical modes: positional queries listed (xgm_andw_list_kernel), conjunctions counted (xgm_andw_all_kernel) */
    /* planned in ordinary (cached) memory and copied to the pinned staging buffer in one go below: the CPU reads pinned
     * host memory at a few GB/s (measured: 250 us per batch for reading 180 KB of device queries back out of it) */
    size_t up_bytes = (size_t)nq * (sizeof(xgm_dev_query) + sizeof(uint32_t) + sizeof(double));
    static thread_local std::vector<uint64_t> plan_buf;
    if (plan_buf.size() * 8 < up_bytes) plan_buf.resize((up_bytes + 7) / 8);
    xgm_dev_query* h_dq = (xgm_dev_query*)plan_buf.data();
    double* h_mp = (double*)(h_dq + nq);
    uint32_t* h_kq = (uint32_t*)(h_mp + nq);
    static thread_local BatchPlan bp_tls;
    BatchPlan& bp = bp_tls;
    bp.reset();
    const uint64_t t_pb = now_ns();
    if ((rc = plan_batch(idx, qs, nq, h_dq, h_kq, h_mp, &bp, false, mode))) return rc;
    const uint64_t t_st = now_ns();
    g_host_ns[1] += t_st - t_pb;
    if (k_stride < bp.k_max) return xgm_set_error(XGM_E_INVALID, "k_stride %u < first+maxitems %u", k_stride, bp.k_max);
    if (list) {
        /* the reference-identical mode of positional queries: the units list their first matches (xgm_prefix_entry, 24 bytes) where their candidates would be */
        if (!bp.andw || !bp.phrase || bp.wide || bp.sided || !d_extra) return XGM_LIST_DECLINED;
        bp.k_stride_c = XGM_PREFIX_CAND_STRIDE(bp.k_max);
    }
    size_t o_cur = 0, o_tab = 0, o_state = 0, o_before = 0, o_ver = 0;
    xgm_all_out all_out;
    memset(&all_out, 0, sizeof all_out);
    if (all) {
        /* every match of the batch goes to one arena (XGM_COUNT_ARENA_ENTRIES, default 16 M entries = 256 MB per batch in flight; a unit whose chunk
         * does not fit flags its query, which xgm_batch_end then counts by itself) */
        if (!bp.andw || bp.phrase || bp.wide || bp.sided || !d_extra || bp.k_max > 64u) return XGM_LIST_DECLINED;
        static const size_t arena_entries = getenv("XGM_COUNT_ARENA_ENTRIES") ? (size_t)std::max(4096ll, atoll(getenv("XGM_COUNT_ARENA_ENTRIES"))) : ((size_t)16 << 20);
        if ((rc = grow(&s->d_arena, &s->cap_arena, arena_entries))) return rc;
        o_cur = 0; o_tab = 64;
        o_state = o_tab + (((size_t)bp.n_work * XGM_ALL_CHUNKS * 4 + 63) & ~(size_t)63);
        o_before = o_state + ((size_t)bp.n_work + nq) * bp.k_stride_c * sizeof(xgm_cand);
        o_ver = o_before + (size_t)bp.n_work * 8;
        if ((rc = grow(&s->d_count, &s->cap_count, o_ver + (size_t)bp.n_work * 4))) return rc;
        all_out.arena = s->d_arena; all_out.cursor = (unsigned long long*)(s->d_count + o_cur); all_out.cap = std::min<unsigned long long>(arena_entries, 0xFFFFFFFFull);
        all_out.chunk_tab = (uint32_t*)(s->d_count + o_tab);
        HIP_TRY(hipMemsetAsync(s->d_count, 0, 64, stream));
    }
    if ((rc = grow(&s->d_cand, &s->cap_cand, (size_t)bp.n_work * bp.k_stride_c))) return rc;
    if ((rc = grow(&s->d_ghdr, &s->cap_ghdr, (size_t)bp.n_work))) return rc;
    /* every per-call input goes up in ONE copy: [dev queries | max_possible | work list | k | group offsets] */
    const uint32_t P = bp.parts, npq = nq * P;              /* pseudo-queries of the units' merge (parts of queries, plan_batch) */
    const size_t o_dq = 0, b_dq = (size_t)nq * sizeof(xgm_dev_query);
    const size_t o_mp = o_dq + b_dq, b_mp = (size_t)npq * 8;
    const size_t o_wk = o_mp + b_mp, b_wk = (size_t)bp.n_work * sizeof(xgm_work);
    const size_t o_kq = o_wk + b_wk, b_kq = (size_t)npq * 4;
    const size_t o_go = o_kq + b_kq, b_go = ((size_t)npq + 1) * 4;
    const size_t o_ro = o_go + b_go, b_ro = rows ? (size_t)nq * 4 : 0;
    const size_t o_fu = (o_ro + b_ro + 15) & ~(size_t)15, b_fu = sizeof(xgm_fuse);      /* (filled below, once the scratch buffers are known) */
    const size_t in_bytes = (o_fu + b_fu + 15) & ~(size_t)15;
    if ((rc = grow_pinned(&s->h_work, &s->cap_hwork, in_bytes))) return rc;
    if ((rc = grow(reinterpret_cast<unsigned char**>(&s->d_in), &s->cap_in, in_bytes))) return rc;
    unsigned char* hin = (unsigned char*)s->h_work;
    memcpy(hin + o_dq, h_dq, b_dq);
    for (uint32_t p = 0; p < P; ++p) { memcpy(hin + o_mp + (size_t)p * nq * 8, h_mp, (size_t)nq * 8); memcpy(hin + o_kq + (size_t)p * nq * 4, h_kq, (size_t)nq * 4); }
    memcpy(hin + o_wk, bp.work.data(), b_wk);
    memcpy(hin + o_go, bp.goff.data(), b_go);
    if (rows) memcpy(hin + o_ro, rows, b_ro);
    /* every query of the batch a conjunction over probe containers only (dense_kind, all of one kind): the kernel written for that */
    bool dense = mode == 0 && bp.andw && !bp.wide && bp.sided == 0 && bp.stripes_per_group <= xgm_dense_max_stripes();
    for (uint32_t i = 0; i < nq && dense; ++i) dense = dense_kind(idx, qs[i]) == (bp.phrase ? 2 : 1);
    static const bool dense_class_old_kernel = getenv("XGM_DENSE_CLASS_OLD_KERNEL") != nullptr;      /* A/B: the same class split, xgm_andw_kernel for both */
    if (dense_class_old_kernel) dense = false;
    const bool fused = bp.fused;
    {
        unsigned char* din_ = (unsigned char*)s->d_in;
        xgm_fuse fu;
        memset(&fu, 0, sizeof fu);
        if (fused) {
            /* every launch leaves its counters at zero — unless it failed part-way: the scratch is then marked and zeroed
             * here before its next use.  (Zeroing before EVERY launch was measured: a fill command between two match kernels adds ~8 us
             * of stream time per batch, gpurun_out/r04b_*.)  A kernel that FAULTS poisons the HIP context: every later call fails. */
            const size_t cap_before = s->cap_arrive;
            if ((rc = grow(&s->d_arrive, &s->cap_arrive, (size_t)std::max<uint32_t>(nq, 1024u)))) return rc;
            if (s->cap_arrive != cap_before || s->arrive_dirty) HIP_TRY(hipMemsetAsync(s->d_arrive, 0, s->cap_arrive * sizeof(uint32_t), stream));
            s->arrive_dirty = true;                          /* until this call has enqueued everything (cleared at its end) */
            /* stress-test switch (tests/test_gpu_stress.py): wipe the units' lists and headers of the previous batch, so that a unit
             * read before it landed shows as missing hits instead of passing for plausible stale ones */
            static const bool poison = getenv("XGM_DEBUG_POISON_SCRATCH") != nullptr;
            if (poison) {
                HIP_TRY(hipMemsetAsync(s->d_cand, 0, (size_t)bp.n_work * bp.k_stride_c * sizeof(xgm_cand), stream));
                HIP_TRY(hipMemsetAsync(s->d_ghdr, 0, (size_t)bp.n_work * sizeof(xgm_group_hdr), stream));
            }
            fu.arrive = s->d_arrive; fu.goff = (const uint32_t*)(din_ + o_go); fu.max_possible = (const double*)(din_ + o_mp);
            fu.row_of = rows ? (const uint32_t*)(din_ + o_ro) : nullptr;
            fu.hits = d_hits; fu.hdrs = d_hdrs; fu.k_stride_out = k_stride;
        }
        if (list) fu.goff = (const uint32_t*)(din_ + o_go);      /* (xgm_andw_list_kernel: the first unit slot of every query) */
        memcpy(hin + o_fu, &fu, sizeof fu);
    }
    const uint64_t t_cp = now_ns();
    g_host_ns[4] += t_cp - t_st;
    bool hist_zeroed = false;
    if (stream != s->stream && !s->inorder) {
        /* asynchronous caller: the inputs go up on the scratch's own stream, so the copy overlaps the
         * kernels of the previous batch still running on the caller's stream */
        if (bp.orw || (bp.andw && bp.phrase)) {
            /* the query-wide histogram is zeroed here, ahead of the upload on the scratch's stream, not between two match kernels on the
             * caller's (the scratch is this batch's alone: whatever used it before has completed — scratch_acquire) */
            /* (list: the same buffer holds the units' match counters, one per unit) */
            const size_t n_hist = list ? (size_t)bp.n_work : (size_t)nq * XGM_OR_HIST;
            if ((rc = grow(&s->d_hist, &s->cap_hist, n_hist))) return rc;
            HIP_TRY(hipMemsetAsync(s->d_hist, 0, n_hist * 4, s->stream));
            hist_zeroed = true;
        }
        HIP_TRY(hipMemcpyAsync(s->d_in, hin, in_bytes, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipEventRecord(s->ev0, s->stream));
        HIP_TRY(hipStreamWaitEvent(stream, s->ev0, 0));
    } else {
        HIP_TRY(hipMemcpyAsync(s->d_in, hin, in_bytes, hipMemcpyHostToDevice, stream));
    }
    const uint64_t t_up = now_ns();
    g_host_ns[5] += t_up - t_cp;
    unsigned char* din = (unsigned char*)s->d_in;
    s->d_queries = (xgm_dev_query*)(din + o_dq);
    s->d_maxposs = (double*)(din + o_mp);
    s->d_work = (xgm_work*)(din + o_wk);
    s->d_kq = (uint32_t*)(din + o_kq);
    s->d_goff = (uint32_t*)(din + o_go);

    xgm_match_launch L;
    L.seg = idx->view;
    L.queries = s->d_queries;
    L.nq = nq; L.n_work = bp.n_work; L.work = s->d_work; L.stripes_per_group = bp.stripes_per_group; L.sub_bits = bp.sub_bits;
    static const bool debug_units = getenv("XGM_DEBUG_UNITS") != nullptr;       /* tools/units.py; single-threaded use only */
    if (debug_units && nq > 1) { g_last_work = bp.work; g_last_ghdr = s->d_ghdr; }
    L.tab_terms = bp.tab_terms; L.cap = bp.cap; L.k_stride = bp.k_stride_c;
    L.phrase = bp.phrase; L.wide = bp.wide; L.sided = bp.andw ? bp.sided : 0; L.orw2 = bp.orw2; L.orw_planes = bp.orw_planes; L.or_flat = bp.or_flat;
    L.tally = idx->tally;
    L.cand = s->d_cand; L.ghdr = s->d_ghdr;
    hipEvent_t pe0 = nullptr, pe1 = nullptr;
    if (idx->profiling) {
        std::lock_guard<std::mutex> lk(idx->scratch_mu);
        if (idx->prof_used == idx->prof_events.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            idx->prof_events.push_back({a, b});
        }
        pe0 = (hipEvent_t)idx->prof_events[idx->prof_used].first;
        pe1 = (hipEvent_t)idx->prof_events[idx->prof_used].second;
        ++idx->prof_used;
    }
    idx->last_kernel = all ? "xgm_andw_all_kernel" : list ? "xgm_andw_list_kernel" : dense ? "xgm_dense_kernel" : bp.andw ? "xgm_andw_kernel" : bp.orw2 ? "xgm_orw2_kernel" : bp.orw ? "xgm_orw_kernel" : bp.and_only ? "xgm_and_kernel" : "xgm_match_kernel";
    idx->last_ghdr = s->d_ghdr; idx->last_n_work = bp.n_work;        /* xgm_last_batch_traffic */
    if (bp.orw || (bp.andw && bp.phrase)) {
        if (!hist_zeroed) {
            const size_t n_hist = list ? (size_t)bp.n_work : (size_t)nq * XGM_OR_HIST;
            if ((rc = grow(&s->d_hist, &s->cap_hist, n_hist))) return rc;
            HIP_TRY(hipMemsetAsync(s->d_hist, 0, n_hist * 4, stream));
        }
        L.hist = s->d_hist;
    }
    if (fused || list) L.fuse = (const xgm_fuse*)(din + o_fu);
    /* the wave kernels' dispatch packets carry their events themselves (xgm_launch.h): the profiling pair, and — when the kernel is the
     * batch's last launch (the fused merge) — the event the download waits for.  XGM_NO_EXT_LAUNCH=1: A/B switch (events recorded around) */
    static const bool ext_launch = getenv("XGM_NO_EXT_LAUNCH") == nullptr;
    const bool ext = ext_launch && !dense && (bp.andw || (bp.orw && !bp.orw2));
    s->last_stop = nullptr;
    if (ext) {
        L.ev_start = pe0;
        L.ev_stop = pe1 ? pe1 : (fused ? s->ev1 : nullptr);
        if (fused) s->last_stop = L.ev_stop;
        pe0 = pe1 = nullptr;
    }
    if (pe0) HIP_TRY(hipEventRecord(pe0, stream));
    if ((rc = all ? xgm_launch_andw_all(L, all_out, stream) : list ? xgm_launch_andw_list(L, stream) : dense ? xgm_launch_dense(L, stream) : bp.andw ? xgm_launch_andw(L, stream)
              : bp.orw ? xgm_launch_orw(L, s->d_hist, stream)
              : bp.and_only ? xgm_launch_and(L, stream) : xgm_launch_match(L, stream)))
        return rc;
    if (pe1) HIP_TRY(hipEventRecord(pe1, stream));
    const uint64_t t_mk = now_ns();
    g_host_ns[6] += t_mk - t_up;
    const uint32_t* d_rows = rows ? (const uint32_t*)(din + o_ro) : nullptr;
    if (list) {
        /* one wave per query walks its units' lists as ProtoMSet + SelectPostList would and writes the row (xgm_frozen.hip) */
        if ((rc = xgm_launch_frozen_finish(s->d_queries, nq, s->d_goff, s->d_cand, s->d_ghdr, bp.k_stride_c, s->d_maxposs, d_rows, d_hits, d_hdrs, d_extra, k_stride, stream)))
            return rc;
    } else if (all) {
        /* the scan of the units' top-k lists (= the merge: pages and headers) and ProtoMSet's count per unit, added onto the rows' extra words (xgm_count.hip) */
        if ((rc = xgm_launch_count_finish(s->d_queries, nq, s->d_work, bp.n_work, s->d_goff, s->d_cand, s->d_ghdr, bp.k_stride_c, all_out, (xgm_cand*)(s->d_count + o_state),
                                          (uint32_t*)(s->d_count + o_ver), (unsigned long long*)(s->d_count + o_before), s->d_maxposs, d_rows, d_hits, d_hdrs, d_extra, k_stride, stream)))
            return rc;
    } else if (fused) {
        /* (the kernel wrote d_hits / d_hdrs) */
    } else if (P == 1u) {
        if ((rc = xgm_launch_merge(s->d_cand, s->d_ghdr, s->d_goff, bp.k_stride_c, s->d_kq, nq, bp.merge_cap, k_stride, d_hits,
                                   d_hdrs, s->d_maxposs, d_rows, stream)))
            return rc;
    } else {
        /* the units of a heavy query exceed one merge's capacity: its parts are merged like queries of their own (row p * nq + q of a
         * scratch result), then the parts' lists per query */
        if ((rc = grow(&s->d_part_hits, &s->cap_part_hits, (size_t)npq * k_stride))) return rc;
        if ((rc = grow(&s->d_part_hdrs, &s->cap_part_hdrs, (size_t)npq))) return rc;
        if ((rc = xgm_launch_merge(s->d_cand, s->d_ghdr, s->d_goff, bp.k_stride_c, s->d_kq, npq, bp.merge_cap, k_stride, s->d_part_hits,
                                   s->d_part_hdrs, s->d_maxposs, nullptr, stream)))
            return rc;
        if ((rc = xgm_launch_merge_parts(s->d_part_hits, s->d_part_hdrs, P, nq, k_stride, s->d_kq, std::max(256u, next_pow2(P * k_stride)), d_hits, d_hdrs,
                                         d_rows, stream)))
            return rc;
    }
    g_host_ns[7] += now_ns() - t_mk;
    g_host_ns[2] += now_ns() - t_st;
    g_host_ns[3] += 1;
    s->arrive_dirty = false;
    return XGM_OK;
}

extern "C" int xgm_debug_host_ns(uint64_t* out4) {      /* (u64[8]) */
    if (!out4) return xgm_set_error(XGM_E_INVALID, "null argument");
    for (int i = 0; i < 8; ++i) out4[i] = g_host_ns[i].exchange(0);
    return XGM_OK;
}

/* Kernel class of a planned query: which match kernel serves it best.  A server's natural batch is heterogeneous
 * (Xapiand's HTTP threads issue whatever the clients send): run_batch cuts it into one launch per class present instead
 * of sending the whole batch to the slowest common denominator. */
enum { XGM_CLS_AND = 0, XGM_CLS_SIDED1, XGM_CLS_SIDED2, XGM_CLS_PHRASE, XGM_CLS_OR, XGM_CLS_OTHER, XGM_CLS_BIGK, XGM_CLS_DENSE_AND, XGM_CLS_DENSE_PHRASE,
       XGM_CLS_OR2,                              /* disjunctions for xgm_orw2_kernel (or2_kind): ONE launch — its flat-array instantiation takes the queries whose
                                                    every term has a container too (measured, round 5: a launch per kind cost 3.7 ms per 256-query batch against
                                                    2.0 for the old kernel alone: three tails, three sets of unit prologues) */
       XGM_CLS_COUNTED,                          /* plain conjunctions with XGM_REPLAY_BATCH_COUNT that xgm_andw_all_kernel takes: ProtoMSet's count on the device */
       XGM_CLS_FROZEN,                           /* positional queries answered as the reference answers them (XGM_REPLAY_BATCH_FROZEN) that xgm_andw_list_kernel takes */
       XGM_CLS_COUNT };

/* xgm_dense_kernel's queries (xgm_dense_and.hip): a conjunction / FILTER — or a positional query that prunes by weight — of 2 to 4
 * terms that ALL have probe containers (xgm_build_dense's own criterion), a page of at most 64.  1 = plain, 2 = positional, 0 = no. */
static int dense_kind(const xgm_index* idx, const xgm_query& q, bool fused) {
    /* (fused: the question xgm_andw_kernel's own dense body asks — the same shapes, plain only, decided per query inside one launch)
     * OFF by default (round 3 measurements): as a launch of its own the kernel costs more than it gains — the class split
     * means two match + merge launches per batch.  XGM_DENSE_KERNEL=1 switches it on (A/B runs, tests/test_gpu_variants.py). */
    static const bool off_alone = getenv("XGM_DENSE_KERNEL") == nullptr || getenv("XGM_NO_ANDW") != nullptr || getenv("XGM_NO_AND_KERNEL") != nullptr;
    static const bool off_fused = getenv("XGM_NO_DENSE_BODY") != nullptr;                           /* A/B switch for measurements */
    const bool off = fused ? off_fused : off_alone;
    static const bool no_pos_prune = getenv("XGM_NO_POS_PRUNE") != nullptr;
    static const bool no_phrase_w = getenv("XGM_NO_PHRASEW") != nullptr;
    if (off || idx->view.n_dense == 0 || idx->dense_min_df == 0 || idx->hdr.stripe_bits > 13u) return 0;
    const uint32_t T = q.n_terms, k = q.first + q.maxitems;
    if (T < 2u || T > xgm_dense_max_terms() || k == 0u || k > xgm_dense_max_k()) return 0;
    const bool positional = (q.op == XGM_OP_PHRASE || q.op == XGM_OP_NEAR) && q.phrase_active;
    if (!(q.op == XGM_OP_AND || q.op == XGM_OP_FILTER || positional)) return 0;
    if (positional && (no_pos_prune || no_phrase_w || !idx->view.dense_pos || q.check_at_least > k)) return 0;
    if ((q.op == XGM_OP_PHRASE || q.op == XGM_OP_NEAR) && !positional) return 0;
    for (uint32_t t = 0; t < T; ++t) {
        const uint32_t id = q.terms[t].term_id;
        if (id == UINT32_MAX || (uint64_t)idx->term_df[id] < idx->dense_min_df || idx->term_wdfub[id] > 254u) return 0;
    }
    return positional ? 2 : 1;
}

/* xgm_flat_unit's queries (xgm_flat_body.inc): a plain conjunction / FILTER of 2..4 terms, a page of at most 64, whose FIRST plan term — the
 * rarest: MultiAndPostList order — has no probe containers but a flat posting array, every other term containers or a flat array.  The
 * criteria are xgm_build_dense's own (xgm_dense.hip: build_containers / build_flat). */
static bool flat_kind(const xgm_index* idx, const xgm_query& q) {
    static const bool off = getenv("XGM_NO_FLAT") != nullptr || getenv("XGM_NO_DENSE") != nullptr || getenv("XGM_NO_ANDW") != nullptr ||
                            getenv("XGM_NO_AND_KERNEL") != nullptr;
    if (off || !idx->view.flat_off || idx->hdr.stripe_bits > 13u) return false;
    const uint32_t T = q.n_terms, k = q.first + q.maxitems;
    if (T < 2u || T > xgm_dense_max_terms() || k == 0u || k > xgm_dense_max_k()) return false;
    const bool positional = (q.op == XGM_OP_PHRASE || q.op == XGM_OP_NEAR) && q.phrase_active;
    if (!(q.op == XGM_OP_AND || q.op == XGM_OP_FILTER || positional)) return false;
    /* positional: the flat arrays carry position offsets, the containers position bases; only queries that prune by weight (XGM_QF_POSPRUNE) */
    static const bool no_pos_prune = getenv("XGM_NO_POS_PRUNE") != nullptr, no_phrase_w = getenv("XGM_NO_PHRASEW") != nullptr,
                      no_flat_phrase = getenv("XGM_NO_FLAT_PHRASE") != nullptr;
    if (positional && (no_pos_prune || no_phrase_w || no_flat_phrase || !idx->view.flat_pos || q.check_at_least > k)) return false;
    for (uint32_t t = 0; t < T; ++t) {
        const uint32_t id = q.terms[t].term_id;
        if (id == UINT32_MAX || idx->term_wdfub[id] > 254u || idx->term_df[id] == 0u) return false;
        const bool dense = idx->view.n_dense && (uint64_t)idx->term_df[id] >= idx->dense_min_df;
        if (t == 0u && dense) return false;
        if (positional && dense && !idx->view.dense_pos) return false;
    }
    return true;
}

/* A disjunction xgm_orw2_kernel takes (xgm_or.hip): <= 8 terms, first + maxitems <= 192, every term of the shard with a one-byte wdf and either a
 * probe container (→ 1 when all have one) or a flat posting array (→ 2; at most two such terms per query).  0: xgm_orw_kernel. */
static int or2_kind(const xgm_index* idx, const xgm_query& q) {
    if (q.op != XGM_OP_OR || q.n_terms == 0 || q.n_terms > 8u || (uint64_t)q.first + q.maxitems > 192u || !xgm_orw2_enabled()) return 0;
    if (!idx->view.n_dense || !idx->dense_min_df) return 0;
    uint32_t flat = 0;
    for (uint32_t t = 0; t < q.n_terms; ++t) {
        const uint32_t id = q.terms[t].term_id;
        if (id == UINT32_MAX) continue;
        if (idx->term_wdfub[id] > 254u) return 0;
        if ((uint64_t)idx->term_df[id] >= idx->dense_min_df) continue;
        if (!idx->view.flat_off || idx->flat_postings >= 0xFFFFFFFFull) return 0;
        if (++flat > 2u) return 0;
    }
    return flat ? 2 : 1;
}

static int classify_query(const xgm_index* idx, const xgm_query& q) {
    const uint32_t T = q.n_terms;
    if (const int dk = dense_kind(idx, q)) return dk == 2 ? XGM_CLS_DENSE_PHRASE : XGM_CLS_DENSE_AND;
    /* the wave kernels keep first + maxitems <= 192 candidates per unit: deeper pages go together to the workgroup kernels
     * instead of dragging a whole class there */
    if (q.op != XGM_OP_OR && q.first + q.maxitems > 192u) return XGM_CLS_BIGK;
    const uint32_t prefix = q.req_mask && !(q.req_mask & (q.req_mask + 1u));            /* required terms = plan positions [0, n_req) */
    switch (q.op) {
    case XGM_OP_OR: return or2_kind(idx, q) ? XGM_CLS_OR2 : XGM_CLS_OR;
    case XGM_OP_PHRASE:
    case XGM_OP_NEAR: return (q.phrase_active && T >= 2u) ? XGM_CLS_PHRASE : XGM_CLS_OTHER;
    case XGM_OP_AND:
    case XGM_OP_FILTER: return T >= 2u ? XGM_CLS_AND : XGM_CLS_OTHER;
    case XGM_OP_AND_NOT: return (T >= 2u && T <= 8u && prefix && (q.req_mask | q.neg_mask) == (1u << T) - 1u) ? XGM_CLS_SIDED1 : XGM_CLS_OTHER;
    case XGM_OP_AND_MAYBE: return (T >= 2u && T <= 8u && prefix) ? XGM_CLS_SIDED2 : XGM_CLS_OTHER;
    default: return XGM_CLS_OTHER;
    }
}

static std::atomic<uint64_t> g_batch_replays[3];    /* diagnostics: rows listed on the device / declined there / answered by xgm_search_replay when collected */
extern "C" int xgm_debug_batch_replay_info(uint64_t* out3) { if (!out3) return -1; for (int i = 0; i < 3; ++i) out3[i] = g_batch_replays[i].load(); return 0; }

/* xgm_andw_list_kernel's queries: positional, a page of at most 64 with check_at_least inside it, 2..4 terms that all have probe containers — or are
 * led by a long-tail term with a flat posting array (the criteria of the two bodies that have a LIST form) */
static bool list_kind(const xgm_index* idx, const xgm_query& q) {
    const bool positional = (q.op == XGM_OP_PHRASE || q.op == XGM_OP_NEAR) && q.phrase_active;
    return positional && (dense_kind(idx, q, true) == 2 || flat_kind(idx, q));
}

/* xgm_andw_all_kernel's queries: a plain conjunction / FILTER the dense or the flat body takes, a page of at most 64 with check_at_least inside it */
static bool all_kind(const xgm_index* idx, const xgm_query& q) {
    if (q.op != XGM_OP_AND && q.op != XGM_OP_FILTER) return false;
    const uint32_t k = q.first + q.maxitems;
    if (k == 0u || k > 64u || q.check_at_least > k) return false;
    return dense_kind(idx, q, true) == 1 || flat_kind(idx, q);
}

/* Runs the batch; results land in d_hits / d_hdrs (device).  Asynchronous on `stream`.
 * d_extra != NULL (device, [nq] zeroed): the queries' XGM_REPLAY_BATCH_* bits are honoured — rows the device answers get their extra word
 * (known_matching_docs | XGM_EXTRA_*), rows it does not take are appended to *host_replay (answered by xgm_batch_end). */
static int run_batch(xgm_index* idx, XgmScratch* s, hipStream_t stream, const xgm_query* qs, uint32_t nq, uint32_t k_stride,
                     xgm_hit* d_hits, xgm_result_hdr* d_hdrs, unsigned long long* d_extra = nullptr, std::vector<uint32_t>* host_replay = nullptr) {
    static const bool no_split = getenv("XGM_NO_CLASS_SPLIT") != nullptr;          /* A/B switch for measurements */
    static const bool no_list = getenv("XGM_NO_LIST_KERNEL") != nullptr;           /* A/B switch (the variant tests): every replay on the host side of the batch */
    uint32_t count[XGM_CLS_COUNT] = {};
    static thread_local std::vector<uint8_t> cls;
    cls.resize(nq);
    uint32_t present = 0;
    s->last_stop = nullptr;
    for (uint32_t i = 0; i < nq; ++i) {
        cls[i] = (uint8_t)classify_query(idx, qs[i]);
        const uint32_t rp = (d_extra && host_replay) ? qs[i].replay : 0u;
        if ((rp & XGM_REPLAY_BATCH_FROZEN) && (qs[i].op == XGM_OP_PHRASE || qs[i].op == XGM_OP_NEAR) && qs[i].phrase_active) {
            if (!no_list && !no_split && list_kind(idx, qs[i])) { cls[i] = XGM_CLS_FROZEN; ++g_batch_replays[0]; }
            else host_replay->push_back(i);
        } else if (rp & XGM_REPLAY_BATCH_COUNT) {
            /* ProtoMSet's count: conjunctions on the device (xgm_andw_all_kernel + xgm_count.hip), any other operator by xgm_search_replay when the batch is collected */
            if (!no_list && !no_split && all_kind(idx, qs[i])) { cls[i] = XGM_CLS_COUNTED; ++g_batch_replays[0]; }
            else host_replay->push_back(i);
        }
        if (count[cls[i]]++ == 0) ++present;
    }
    if ((present <= 1u && !count[XGM_CLS_FROZEN] && !count[XGM_CLS_COUNTED]) || no_split) return run_class_batch(idx, s, stream, qs, nq, k_stride, d_hits, d_hdrs, nullptr);
    /* one launch per class present, all on `stream`; the first uses the caller's scratch, the others take their own
     * from the pool (each is marked pending behind its launch) */
    static thread_local std::vector<xgm_query> sub;
    static thread_local std::vector<uint32_t> rows;
    bool first = true;
    int rc = XGM_OK;
    for (int c = 0; c < XGM_CLS_COUNT && rc == XGM_OK; ++c) {
        if (!count[c]) continue;
        sub.clear(); rows.clear();
        for (uint32_t i = 0; i < nq; ++i) if (cls[i] == c) { sub.push_back(qs[i]); rows.push_back(i); }
        XgmScratch* sc = s;
        if (!first && (rc = scratch_acquire(idx, &sc))) break;
        const uint32_t* rows_arg = (present == 1u) ? nullptr : rows.data();
        rc = run_class_batch(idx, sc, stream, sub.data(), (uint32_t)sub.size(), k_stride, d_hits, d_hdrs, rows_arg, c == XGM_CLS_FROZEN ? 1 : c == XGM_CLS_COUNTED ? 2 : 0, d_extra);
        if (rc == XGM_LIST_DECLINED) {
            /* (the batch as a whole is not the listing kernel's: the intended-semantics launch now, the replays when the batch is collected) */
            for (uint32_t r : rows) host_replay->push_back(r);
            g_batch_replays[0] -= rows.size();
            rc = run_class_batch(idx, sc, stream, sub.data(), (uint32_t)sub.size(), k_stride, d_hits, d_hdrs, rows_arg);
        }
        if (sc != s) {
            if (hipEventRecord(sc->ev_done, stream) == hipSuccess) sc->pending = true; else hipStreamSynchronize(stream);
            scratch_release(idx, sc);
        }
        first = false;
    }
    s->last_stop = nullptr;              /* several launches: the download waits behind the last one (an event recorded in stream order) */
    return rc;
}

static hipStream_t pick_stream(xgm_index* idx, XgmScratch* s) { return idx->stream ? (hipStream_t)idx->stream : s->stream; }

static int batcher_submit(xgm_index* idx, const xgm_query* q, uint32_t k_stride, xgm_hit* hits, xgm_result_hdr* hdr, uint64_t* known = nullptr);

static int search_batch_now(xgm_index* idx, const xgm_query* qs, uint32_t nq, uint32_t k_stride, xgm_hit* hits, xgm_result_hdr* hdrs, uint64_t* known = nullptr);

extern "C" int xgm_search_batch(xgm_index* idx, const xgm_query* qs, uint32_t nq, uint32_t k_stride, xgm_hit* hits,
                                xgm_result_hdr* hdrs) {
    if (!idx || !qs || !hits || !hdrs) return xgm_set_error(XGM_E_INVALID, "null argument");
    if (nq == 0) return XGM_OK;
    /* server mode: single-query calls of many host threads ride in shared launches (xgm_index_set_batching) */
    if (nq == 1 && idx->batcher) return batcher_submit(idx, qs, k_stride, hits, hdrs);
    return search_batch_now(idx, qs, nq, k_stride, hits, hdrs);
}

extern "C" int xgm_search_batch_known(xgm_index* idx, const xgm_query* qs, uint32_t nq, uint32_t k_stride, xgm_hit* hits, xgm_result_hdr* hdrs, uint64_t* known) {
    if (!idx || !qs || !hits || !hdrs) return xgm_set_error(XGM_E_INVALID, "null argument");
    if (nq == 0) return XGM_OK;
    if (known) memset(known, 0, (size_t)nq * 8);
    if (nq == 1 && idx->batcher) return batcher_submit(idx, qs, k_stride, hits, hdrs, known);
    return search_batch_now(idx, qs, nq, k_stride, hits, hdrs, known);
}

/* ---- a batch in flight ---------------------------------------------------------------------------------------------------
 * begin: plan the work units, upload, launch the match (+ merge) kernels and the copy of the results into PINNED host memory the
 * library owns — all asynchronous on the batch's own stream (or the index's, xgm_index_set_stream) — and return at once; end: wait
 * for that batch alone.  A server (the dispatcher of xgm_index_set_batching, bench.py's timed loop) keeps two or three batches in
 * flight: the host plans batch i + 1 while the GPU runs batch i, consecutive batches on different streams overlap on the chip (the
 * tail of one — the last queries' merges — under the head of the next), and every batch's hits reach the HOST.  The scratch (device
 * + pinned buffers, stream) belongs to the batch until xgm_batch_release. */
struct xgm_inflight {
    xgm_index* idx = nullptr;
    XgmScratch* s = nullptr;
    uint32_t nq = 0, k_stride = 0;
    bool ended = false;
    int rc_end = XGM_OK;
    /* queries with XGM_REPLAY_BATCH_* bits: the downloaded rows end with [nq] extra words (known_matching_docs | XGM_EXTRA_*); the plans are
     * kept so that xgm_batch_end can answer the rows the device declined (host_replay: known at launch; XGM_EXTRA_FALLBACK: found on the device) */
    bool has_extra = false;
    std::vector<xgm_query> plans;
    std::vector<uint32_t> host_replay;
};

static int batch_begin(xgm_index* idx, const xgm_query* qs, uint32_t nq, uint32_t k_stride, xgm_inflight** out, hipStream_t on = nullptr, bool inorder = false) {
    *out = nullptr;
    int rc = use_device(idx->device);
    if (rc) return rc;
    XgmScratch* s;
    if ((rc = scratch_acquire(idx, &s))) return rc;
    hipStream_t stream = on ? on : pick_stream(idx, s);
    s->inorder = inorder;
    bool has_extra = false;
    for (uint32_t i = 0; i < nq && !has_extra; ++i) has_extra = qs[i].replay != 0u;
    std::vector<uint32_t> host_replay;
    do {
        /* hits and headers share one device buffer → one download (+ one extra word per query when any carries replay bits) */
        const size_t n_hit = (size_t)nq * k_stride;
        if ((rc = grow(&s->d_hits, &s->cap_hits, n_hit + (size_t)nq * 2 + (has_extra ? (nq + 1u) / 2u : 0u)))) break;
        xgm_result_hdr* d_hdrs = reinterpret_cast<xgm_result_hdr*>(s->d_hits + n_hit);
        unsigned long long* d_extra = has_extra ? reinterpret_cast<unsigned long long*>(s->d_hits + n_hit + (size_t)nq * 2) : nullptr;
        const size_t down = n_hit * sizeof(xgm_hit) + (size_t)nq * sizeof(xgm_result_hdr) + (has_extra ? (size_t)nq * 8 : 0);
        if ((rc = grow_pinned(&s->h_down, &s->cap_down, down))) break;
        if (has_extra) {
            const hipError_t ez = hipMemsetAsync(d_extra, 0, (size_t)nq * 8, stream);
            if (ez != hipSuccess) { rc = xgm_launch_error("hipMemsetAsync", (int)ez, hipGetErrorString(ez)); break; }
        }
        if ((rc = run_batch(idx, s, stream, qs, nq, k_stride, s->d_hits, d_hdrs, d_extra, has_extra ? &host_replay : nullptr))) break;
        /* an index bound to ONE stream (xgm_index_set_stream) runs its batches' kernels back to back there; the download of batch i then
         * goes on the scratch's own stream behind an event, so that it does not sit between the match kernels of batches i and i + 1
         * (XGM_COPY_ON_BATCH_STREAM=1: A/B switch, the copy in stream order) */
        static const bool copy_in_order = getenv("XGM_COPY_ON_BATCH_STREAM") != nullptr;
        hipError_t e = hipSuccess;
        if (stream != s->stream && !copy_in_order && !s->inorder) {
            hipEvent_t after = s->last_stop;                       /* the batch's one launch carries its completion event itself */
            if (!after) { after = s->ev1; e = hipEventRecord(after, stream); }
            if (e == hipSuccess) e = hipStreamWaitEvent(s->stream, after, 0);
            if (e == hipSuccess) e = hipMemcpyAsync(s->h_down, s->d_hits, down, hipMemcpyDeviceToHost, s->stream);
            if (e == hipSuccess) e = hipEventRecord(s->ev_done, s->stream);
        } else {
            e = hipMemcpyAsync(s->h_down, s->d_hits, down, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipEventRecord(s->ev_done, stream);
        }
        if (e != hipSuccess) { rc = xgm_launch_error("result copy", (int)e, hipGetErrorString(e)); break; }
    } while (0);
    if (rc) {
        hipStreamSynchronize(stream);              /* whatever was enqueued no longer uses the scratch */
        hipStreamSynchronize(s->stream);
        scratch_release(idx, s);
        return rc;
    }
    xgm_inflight* f = new xgm_inflight();
    f->idx = idx; f->s = s; f->nq = nq; f->k_stride = k_stride;
    if (has_extra) { f->has_extra = true; f->plans.assign(qs, qs + nq); f->host_replay.swap(host_replay); }
    *out = f;
    return XGM_OK;
}

extern "C" int xgm_search_batch_begin(xgm_index* idx, const xgm_query* qs, uint32_t nq, uint32_t k_stride, xgm_inflight** out) {
    if (!idx || !qs || !out || nq == 0) return xgm_set_error(XGM_E_INVALID, "null argument");
    if (idx->device == XGM_DEVICE_NONE) return xgm_set_error(XGM_E_NO_DEVICE, "index opened without a device");
    return batch_begin(idx, qs, nq, k_stride, out);
}

extern "C" int xgm_get_mset_batch_begin(xgm_index* idx, const xgm_query_desc* descs, const xgm_global_stats* gs, uint32_t nq, uint32_t k_stride,
                                        xgm_inflight** out) {
    if (!idx || !descs || !out || nq == 0) return xgm_set_error(XGM_E_INVALID, "null argument");
    if (idx->device == XGM_DEVICE_NONE) return xgm_set_error(XGM_E_NO_DEVICE, "index opened without a device");
    static thread_local std::vector<xgm_query> plans;
    plans.resize(nq);
    const uint64_t t0 = now_ns();
    for (uint32_t i = 0; i < nq; ++i) {
        int rc = xgm_plan_query(idx, &descs[i], gs ? &gs[i] : nullptr, &plans[i]);
        if (rc) return rc;
    }
    g_host_ns[0] += now_ns() - t0;
    return batch_begin(idx, plans.data(), nq, k_stride, out);
}

/* The rows of a collected batch that carry replay bits and were not answered on the device: each by xgm_search_replay, into the batch's own
 * pinned rows.  A row needs it only when its page is full (fewer matches than the page: the reference's answer IS the intended one). */
static int batch_host_replays(xgm_inflight* f) {
    xgm_hit* h_hits = (xgm_hit*)f->s->h_down;
    xgm_result_hdr* h_hdrs = (xgm_result_hdr*)(h_hits + (size_t)f->nq * f->k_stride);
    unsigned long long* h_extra = (unsigned long long*)(h_hdrs + f->nq);
    for (uint32_t i = 0; i < f->nq; ++i) if (h_extra[i] & XGM_EXTRA_FALLBACK) { f->host_replay.push_back(i); ++g_batch_replays[1]; }
    if (f->host_replay.empty()) return XGM_OK;
    std::vector<xgm_hit> page(XGM_MAX_K);
    for (uint32_t i : f->host_replay) {
        const xgm_query& q = f->plans[i];
        const uint32_t k = q.first + q.maxitems;
        const bool device_declined = (h_extra[i] & XGM_EXTRA_FALLBACK) != 0ull;
        const bool positional = (q.op == XGM_OP_PHRASE || q.op == XGM_OP_NEAR) && q.phrase_active;
        const bool frozen = (q.replay & XGM_REPLAY_BATCH_FROZEN) && positional;
        h_extra[i] = 0;
        if (!frozen && !(q.replay & XGM_REPLAY_BATCH_COUNT)) continue;
        /* a page that did not fill: every match was shown to ProtoMSet — the intended row is the reference's, the count its length */
        if (!device_declined && h_hdrs[i].n_hits < k && !(h_hdrs[i].matches_exact & XGM_MATCHES_LOWER_BOUND)) { h_extra[i] = h_hdrs[i].n_hits; continue; }
        xgm_result_hdr hdr;
        uint64_t known = 0;
        const int rc = xgm_search_replay(f->idx, &q, frozen ? XGM_REPLAY_FROZEN_WEIGHT : XGM_REPLAY_COUNT, page.data(), &hdr, &known);
        if (rc) return rc;
        ++g_batch_replays[2];
        h_extra[i] = known;
        if (!frozen && !device_declined) { h_hdrs[i].matches_exact = hdr.matches_exact; continue; }      /* (the counting mode keeps the batch's own page: ProtoMSet keeps the same documents) */
        hdr.max_possible = q.max_possible;
        memcpy(h_hits + (size_t)i * f->k_stride, page.data(), (size_t)hdr.n_hits * sizeof(xgm_hit));
        h_hdrs[i] = hdr;
    }
    return XGM_OK;
}

extern "C" int xgm_batch_end(xgm_inflight* f, const xgm_hit** hits, const xgm_result_hdr** hdrs) {
    if (!f) return xgm_set_error(XGM_E_INVALID, "null argument");
    if (!f->ended) {
        f->ended = true;
        hipError_t e = hipEventSynchronize(f->s->ev_done);
        if (e != hipSuccess) f->rc_end = xgm_launch_error("batch completion", (int)e, hipGetErrorString(e));
        else if (f->has_extra) f->rc_end = batch_host_replays(f);
    }
    if (hits) *hits = (const xgm_hit*)f->s->h_down;
    if (hdrs) *hdrs = (const xgm_result_hdr*)((const xgm_hit*)f->s->h_down + (size_t)f->nq * f->k_stride);
    return f->rc_end;
}

extern "C" int xgm_batch_known(xgm_inflight* f, const uint64_t** known) {
    if (!f || !known) return xgm_set_error(XGM_E_INVALID, "null argument");
    *known = nullptr;
    if (!f->ended) return xgm_set_error(XGM_E_INVALID, "xgm_batch_known before xgm_batch_end");
    if (f->has_extra) *known = (const uint64_t*)((const xgm_result_hdr*)((const xgm_hit*)f->s->h_down + (size_t)f->nq * f->k_stride) + f->nq);
    return f->rc_end;
}

extern "C" int xgm_batch_poll(xgm_inflight* f) {
    if (!f) return xgm_set_error(XGM_E_INVALID, "null argument");
    if (f->ended) return 1;
    return hipEventQuery(f->s->ev_done) == hipSuccess ? 1 : 0;
}

extern "C" void xgm_batch_release(xgm_inflight* f) {
    if (!f) return;
    if (!f->ended) hipEventSynchronize(f->s->ev_done);
    scratch_release(f->idx, f->s);
    delete f;
}

static int search_batch_now(xgm_index* idx, const xgm_query* qs, uint32_t nq, uint32_t k_stride, xgm_hit* hits, xgm_result_hdr* hdrs, uint64_t* known) {
    xgm_inflight* f = nullptr;
    int rc = batch_begin(idx, qs, nq, k_stride, &f);
    if (rc) return rc;
    const xgm_hit* h_hits = nullptr;
    const xgm_result_hdr* h_hdrs = nullptr;
    rc = xgm_batch_end(f, &h_hits, &h_hdrs);
    if (rc == XGM_OK) {
        memcpy(hdrs, h_hdrs, (size_t)nq * sizeof(xgm_result_hdr));
        /* only the valid prefix of each row is defined on the device */
        for (uint32_t i = 0; i < nq; ++i)
            memcpy(hits + (size_t)i * k_stride, h_hits + (size_t)i * k_stride, (size_t)h_hdrs[i].n_hits * sizeof(xgm_hit));
        const uint64_t* kn = nullptr;
        if (known && xgm_batch_known(f, &kn) == XGM_OK && kn) memcpy(known, kn, (size_t)nq * 8);
    }
    xgm_batch_release(f);
    return rc;
}

/* ---- searches under a value sort (first version) -------------------------------------------------------------
 * One query at a time, synchronous, its own device buffers: correctness first.  Every query shape the workgroup kernel handles
 * (plain operators, PHRASE / NEAR, nested trees).  The unit decomposition is that kernel's (plan_batch with the wave kernels
 * switched off); the units' candidates come back to the host, which merges them
 * under the same comparison the kernel ranks by. */

static int attach_ordinals(xgm_index* idx, uint32_t slot, const uint32_t* ord, size_t n_ord, uint32_t n_distinct);

extern "C" int xgm_index_attach_column(xgm_index* idx, const char* column_path) {
    if (!idx || !column_path) return xgm_set_error(XGM_E_INVALID, "null argument");
    if (idx->device == XGM_DEVICE_NONE) return xgm_set_error(XGM_E_NO_DEVICE, "index opened without a device");
    FILE* f = fopen(column_path, "rb");
    if (!f) return xgm_set_error(XGM_E_IO, "cannot open %s: %s", column_path, strerror(errno));
    char magic[8];
    uint32_t h32[4];
    std::vector<uint32_t> ord;
    bool ok = fread(magic, 1, 8, f) == 8 && fread(h32, 4, 4, f) == 4 && memcmp(magic, "XGMCOL1", 8) == 0;
    if (ok && h32[1] != idx->hdr.lastdocid) { fclose(f); return xgm_set_error(XGM_E_INVALID, "%s: column of %u documents, index of %u (another revision?)", column_path, h32[1], idx->hdr.lastdocid); }
    if (ok) { ord.resize((size_t)h32[1] + 1); ok = fread(ord.data(), 4, ord.size(), f) == ord.size(); }
    fclose(f);
    if (!ok) return xgm_set_error(XGM_E_INVALID, "%s is not a column file", column_path);
    for (uint32_t o : ord) if (o > h32[2]) return xgm_set_error(XGM_E_INVALID, "%s: ordinal beyond the distinct values", column_path);
    return attach_ordinals(idx, h32[0], ord.data(), ord.size(), h32[2]);
}

/* A replaced column's device array is not freed while searches that copied its pointer may still be launching with it: it
 * goes to the index's graveyard, emptied when the index closes. */
static int attach_ordinals(xgm_index* idx, uint32_t slot, const uint32_t* ord, size_t n_ord, uint32_t n_distinct) {
    int rc = use_device(idx->device);
    if (rc) return rc;
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, n_ord * 4));
    hipError_t e = hipMemcpy(d, ord, n_ord * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(d); return xgm_launch_error("column upload", (int)e, hipGetErrorString(e)); }
    std::lock_guard<std::mutex> lk(idx->columns_mu);
    std::pair<void*, uint32_t>& col = idx->columns[slot];
    if (col.first) idx->retired_columns.push_back(col.first);
    col = std::make_pair(d, n_distinct);
    return XGM_OK;
}

extern "C" int xgm_index_attach_column_ordinals(xgm_index* idx, uint32_t slot, const uint32_t* ord, uint32_t n_ord, uint32_t n_distinct) {
    if (!idx || !ord) return xgm_set_error(XGM_E_INVALID, "null argument");
    if (idx->device == XGM_DEVICE_NONE) return xgm_set_error(XGM_E_NO_DEVICE, "index opened without a device");
    if (n_ord != idx->hdr.lastdocid + 1u) return xgm_set_error(XGM_E_INVALID, "column of %u entries, index of %u documents (another revision?)", n_ord, idx->hdr.lastdocid);
    for (uint32_t i = 0; i < n_ord; ++i) if (ord[i] > n_distinct) return xgm_set_error(XGM_E_INVALID, "ordinal beyond the distinct values");
    return attach_ordinals(idx, slot, ord, n_ord, n_distinct);
}

/* a scratch goes back to the pool only when nothing enqueued on its stream still uses it (the early returns) */
/* The whole-match list of a replay / xgm_search_all (d_all) is sized by the match's upper bound — hundreds of MB for a frequent-term query on a
 * 10 M-document shard — and the pool keeps a scratch per concurrent caller: ONE scratch of the pool keeps such a list (a caller that replays one
 * query after another gets it back: no allocation per call), any further one is trimmed when it returns. */
constexpr size_t kAllKeepBytes = (size_t)64 << 20;
struct ScratchRelease {
    xgm_index* i; XgmScratch* s;
    ~ScratchRelease() {
        hipStreamSynchronize(s->stream);
        if (s->cap_all > kAllKeepBytes) {
            bool another = false;
            {
                std::lock_guard<std::mutex> lk(i->scratch_mu);
                for (const XgmScratch* o : i->scratch_pool) another = another || o->cap_all > kAllKeepBytes;
            }
            if (another) { hipFree(s->d_all); s->d_all = nullptr; s->cap_all = 0; }
        }
        scratch_release(i, s);
    }
};

namespace {
struct DeviceBuffers {                       /* freed on every way out */
    std::vector<void*> p;
    ~DeviceBuffers() { for (void* q : p) if (q) hipFree(q); }
    int alloc(void** out, size_t bytes) {
        *out = nullptr;
        hipError_t e = hipMalloc(out, bytes ? bytes : 16);
        if (e != hipSuccess) return xgm_launch_error("hipMalloc", (int)e, hipGetErrorString(e));
        p.push_back(*out);
        return XGM_OK;
    }
};
}  // namespace

/* spy_slot >= 0: also count the matching documents by their value in that slot (counts[0 .. n_counts), n_counts = the column's
 * distinct values + 1).  collapse_slot >= 0: Enquire::set_collapse_key(collapse_slot, cmax) — the kernel collapses inside every
 * unit, the merge below once more; the per-key match counts (the spy mechanism on the collapse column) give the items' collapse
 * counts and the collapsed lower bound.  sort == NULL: by relevance. */
static int sorted_core(xgm_index* idx, const xgm_query* q, const xgm_sort_spec* sort, xgm_hit* hits, uint32_t* hit_ord, xgm_result_hdr* hdr,
                       int spy_slot, uint32_t* counts, uint32_t n_counts,
                       int collapse_slot = -1, uint32_t cmax = 0, uint32_t* hit_cord = nullptr, uint32_t* hit_ccount = nullptr, uint64_t* collapsed_lb = nullptr) {
    if (!idx || !q || !hits || !hdr) return xgm_set_error(XGM_E_INVALID, "null argument");
    if (idx->device == XGM_DEVICE_NONE) return xgm_set_error(XGM_E_NO_DEVICE, "index opened without a device");
    if (sort && (sort->sort_by < XGM_SORT_VALUE || sort->sort_by > XGM_SORT_RELEVANCE_VALUE)) return xgm_set_error(XGM_E_INVALID, "sort_by %u", sort->sort_by);
    if (!sort && collapse_slot < 0 && spy_slot < 0) return xgm_set_error(XGM_E_INVALID, "neither a sort nor a collapse key nor a spy");
    if (collapse_slot >= 0 && (cmax == 0 || spy_slot >= 0)) return xgm_set_error(XGM_E_INVALID, "collapse_max 0, or a spy together with a collapse key");
    const uint32_t mode = sort ? sort->sort_by : 4u;
    const bool reverse = sort && sort->reverse;
    const uint32_t* d_ord = nullptr;
    const uint32_t* d_spy_ord = nullptr;
    const uint32_t* d_cord = nullptr;
    std::vector<uint32_t> key_counts;
    {
        std::lock_guard<std::mutex> lk(idx->columns_mu);
        if (sort) {
            auto it = idx->columns.find(sort->slot);
            if (it == idx->columns.end()) return XGM_UNSUPPORTED;
            d_ord = (const uint32_t*)it->second.first;
        }
        if (spy_slot >= 0) {
            auto it = idx->columns.find((uint32_t)spy_slot);
            if (it == idx->columns.end()) return XGM_UNSUPPORTED;
            if (!counts || n_counts != it->second.second + 1u) return xgm_set_error(XGM_E_INVALID, "spy: %u counters for a column of %u distinct values (+ 1 for no value)", n_counts, it->second.second);
            d_spy_ord = (const uint32_t*)it->second.first;
        }
        if (collapse_slot >= 0) {
            auto it = idx->columns.find((uint32_t)collapse_slot);
            if (it == idx->columns.end()) return XGM_UNSUPPORTED;
            d_cord = (const uint32_t*)it->second.first;
            d_spy_ord = d_cord;                       /* matches per collapse key */
            key_counts.assign((size_t)it->second.second + 1u, 0u);
            counts = key_counts.data(); n_counts = (uint32_t)key_counts.size();
        }
    }
    int rc = use_device(idx->device);
    if (rc) return rc;
    xgm_dev_query dq;
    uint32_t kq = 0;
    double mp = 0;
    BatchPlan bp;
    if ((rc = plan_batch(idx, q, 1, &dq, &kq, &mp, &bp, true))) return rc;
    if (bp.andw || bp.orw || bp.and_only || dq.k == 0 || bp.cap > 8u * XGM_WG) return XGM_UNSUPPORTED;
    if (xgm_match_sorted_smem_bytes(idx->hdr.stripe_bits - bp.sub_bits, bp.tab_terms, bp.phrase, bp.cap, bp.wide, bp.stripes_per_group) > 160u * 1024u) return XGM_UNSUPPORTED;
    const uint32_t k = dq.k, n_work = bp.n_work;
    /* per-call device and pinned buffers + a stream from the index's scratch pool (re-used across calls: no allocation, no null stream —
     * searches of concurrent threads do not serialise): [query | work list] go up in one copy, [headers | counters | candidates] come down in one */
    XgmScratch* sc;
    if ((rc = scratch_acquire(idx, &sc))) return rc;
    ScratchRelease release_{idx, sc};            /* (synchronises the stream before the scratch goes back to the pool: an error return leaves nothing in flight on it) */
    hipStream_t stream = sc->stream;
    const size_t o_q = 0, b_q = (sizeof dq + 15) & ~(size_t)15;
    const size_t o_wk = o_q + b_q, b_wk = ((size_t)n_work * sizeof(xgm_work) + 15) & ~(size_t)15;
    const size_t up_bytes = o_wk + b_wk;
    const size_t o_gh = up_bytes, b_gh = (size_t)n_work * sizeof(xgm_group_hdr);
    const size_t o_ct = o_gh + b_gh, b_ct = d_spy_ord ? (((size_t)n_counts * 4 + 15) & ~(size_t)15) : 0;
    const size_t o_cd = o_ct + b_ct, b_cd = (size_t)n_work * k * sizeof(xgm_cand_sorted);
    const size_t total = o_cd + b_cd;
    if ((rc = grow(&sc->d_sorted, &sc->cap_sorted, total))) return rc;
    if ((rc = grow_pinned(&sc->h_sorted, &sc->cap_hsorted, total))) return rc;
    unsigned char* hb = (unsigned char*)sc->h_sorted;
    memcpy(hb + o_q, &dq, sizeof dq);
    memcpy(hb + o_wk, bp.work.data(), (size_t)n_work * sizeof(xgm_work));
    HIP_TRY(hipMemcpyAsync(sc->d_sorted, hb, up_bytes, hipMemcpyHostToDevice, stream));
    xgm_dev_query* d_q = (xgm_dev_query*)(sc->d_sorted + o_q);
    xgm_work* d_work = (xgm_work*)(sc->d_sorted + o_wk);
    xgm_group_hdr* d_ghdr = (xgm_group_hdr*)(sc->d_sorted + o_gh);
    uint32_t* d_counts = d_spy_ord ? (uint32_t*)(sc->d_sorted + o_ct) : nullptr;
    xgm_cand_sorted* d_cand = (xgm_cand_sorted*)(sc->d_sorted + o_cd);
    if (d_counts) HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)n_counts * 4, stream));
    xgm_match_launch L;
    L.seg = idx->view;
    L.queries = d_q;
    L.nq = 1; L.n_work = n_work; L.work = d_work; L.stripes_per_group = bp.stripes_per_group; L.sub_bits = bp.sub_bits;
    L.tab_terms = bp.tab_terms; L.cap = bp.cap; L.k_stride = k;
    L.phrase = bp.phrase; L.wide = bp.wide; L.sided = 0;
    L.cand = nullptr; L.ghdr = d_ghdr;
    if ((rc = xgm_launch_match_sorted(L, d_ord, mode, reverse ? 1u : 0u, d_spy_ord, d_counts, d_cord, cmax, d_cand, stream))) return rc;
    HIP_TRY(hipMemcpyAsync(hb + o_gh, sc->d_sorted + o_gh, total - o_gh, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    const xgm_group_hdr* gh = (const xgm_group_hdr*)(hb + o_gh);
    const xgm_cand_sorted* cand = (const xgm_cand_sorted*)(hb + o_cd);
    if (d_counts) memcpy(counts, hb + o_ct, (size_t)n_counts * 4);
    /* merge the units: their best k each under the comparison the kernel used; the whole match's best weight and count */
    std::vector<xgm_cand_sorted> all;
    uint64_t matches = 0, max_w = 0;
    uint32_t max_d = UINT32_MAX, max_m = 0;
    for (uint32_t u = 0; u < n_work; ++u) {
        const xgm_group_hdr& g = gh[u];
        if (g.n_cand > k) return xgm_set_error(XGM_E_DEVICE, "sorted search: unit %u reports %u candidates for k = %u", u, g.n_cand, k);
        matches += g.matches;
        all.insert(all.end(), cand + (size_t)u * k, cand + (size_t)u * k + g.n_cand);
        if (g.c_pad[0] != UINT32_MAX && (max_d == UINT32_MAX || g.c_pos > max_w || (g.c_pos == max_w && g.c_pad[0] < max_d))) { max_w = g.c_pos; max_d = g.c_pad[0]; max_m = g.c_pad[1]; }
    }
    const bool use_x = mode == XGM_SORT_VALUE_RELEVANCE || mode == XGM_SORT_RELEVANCE_VALUE;
    std::sort(all.begin(), all.end(), [&](const xgm_cand_sorted& a, const xgm_cand_sorted& b) {
        if (a.kw != b.kw) return a.kw > b.kw;
        if (use_x && a.kx != b.kx) return a.kx > b.kx;
        return a.did < b.did;
    });
    if (collapse_slot >= 0) {
        /* the merged ranking collapsed once more: of every key the first cmax stay */
        std::vector<uint32_t> kept_of(key_counts.size(), 0u);
        size_t out = 0;
        for (const xgm_cand_sorted& c : all) {
            if (c.cord >= kept_of.size()) return xgm_set_error(XGM_E_DEVICE, "sorted search: collapse ordinal %u beyond the column", c.cord);
            if (c.cord == 0u || kept_of[c.cord]++ < cmax) all[out++] = c;
        }
        all.resize(out);
        if (collapsed_lb) {
            /* Collapser::get_matches_lower_bound: documents without a key + per key the entries that stay */
            uint64_t lb = key_counts[0];
            for (size_t o = 1; o < key_counts.size(); ++o) lb += std::min<uint32_t>(key_counts[o], cmax);
            *collapsed_lb = lb;
        }
    }
    const uint32_t n = (uint32_t)std::min<size_t>(k, all.size());
    const bool weight_first = mode >= XGM_SORT_RELEVANCE_VALUE;             /* relevance then value, relevance alone */
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t wbits = weight_first ? all[i].kw : all[i].kx;
        const uint32_t okey = (uint32_t)(weight_first ? all[i].kx : all[i].kw);
        hits[i].docid = all[i].did;
        hits[i].subqs_matched = all[i].subqs;
        memcpy(&hits[i].weight, &wbits, 8);
        if (hit_ord) hit_ord[i] = sort ? (reverse ? okey : ~okey) : 0u;
        if (hit_cord) hit_cord[i] = all[i].cord;
        if (hit_ccount) hit_ccount[i] = (collapse_slot >= 0 && all[i].cord && key_counts[all[i].cord] > cmax) ? key_counts[all[i].cord] - cmax : 0u;
    }
    memset(hdr, 0, sizeof *hdr);
    hdr->n_hits = n;
    hdr->matches_exact = matches;
    hdr->max_possible = q->max_possible;
    if (max_d != UINT32_MAX) { memcpy(&hdr->max_attained, &max_w, 8); hdr->max_weight_subqs_matched = max_m; }
    return XGM_OK;
}

extern "C" int xgm_search_sorted(xgm_index* idx, const xgm_query* q, const xgm_sort_spec* sort, xgm_hit* hits, uint32_t* hit_ord,
                                 xgm_result_hdr* hdr) {
    if (!sort) return xgm_set_error(XGM_E_INVALID, "null argument");
    return sorted_core(idx, q, sort, hits, hit_ord, hdr, -1, nullptr, 0);
}

/* nq searches under ONE sort in ONE launch (include/xgm.h: xgm_search_sorted_batch): the workgroup kernel's units of every query go up in one
 * work list (plan_batch), every unit leaves its best k under the sort, the host merges each query's units — what xgm_search_sorted does
 * for one query, without a launch, an upload, a download and a synchronisation per query. */
/* collapse_slot >= 0 (sorted_core's counterpart for a batch): the kernel collapses inside every unit, the merge below once more per query; the
 * matches per collapse key — the spy mechanism on the collapse column, one row per query — give the collapse counts and lower bounds */
static int sorted_batch_core(xgm_index* idx, const xgm_query* qs, uint32_t nq, const xgm_sort_spec* sort, uint32_t k_stride, xgm_hit* hits,
                             uint32_t* hit_ord, xgm_result_hdr* hdrs, int spy_slot, uint32_t* counts, uint32_t n_counts,
                             int collapse_slot = -1, uint32_t cmax = 0, uint32_t* hit_cord = nullptr, uint32_t* hit_ccount = nullptr, uint64_t* collapsed_lb = nullptr) {
    if (!idx || !qs || !hits || !hdrs || nq == 0) return xgm_set_error(XGM_E_INVALID, "null argument");
    if (!sort && collapse_slot < 0 && spy_slot < 0) return xgm_set_error(XGM_E_INVALID, "neither a sort nor a collapse key nor a spy");
    if (idx->device == XGM_DEVICE_NONE) return xgm_set_error(XGM_E_NO_DEVICE, "index opened without a device");
    if (sort && (sort->sort_by < XGM_SORT_VALUE || sort->sort_by > XGM_SORT_RELEVANCE_VALUE)) return xgm_set_error(XGM_E_INVALID, "sort_by %u", sort->sort_by);
    if (collapse_slot >= 0 && (cmax == 0 || spy_slot >= 0)) return xgm_set_error(XGM_E_INVALID, "collapse_max 0, or a spy together with a collapse key");
    const uint32_t mode = sort ? sort->sort_by : 4u;
    const bool reverse = sort && sort->reverse != 0;
    const uint32_t* d_ord = nullptr;
    const uint32_t* d_spy_ord = nullptr;
    const uint32_t* d_cord = nullptr;
    std::vector<uint32_t> key_counts;                         /* collapse: [nq][n_counts] matches per key */
    {
        std::lock_guard<std::mutex> lk(idx->columns_mu);
        if (sort) {
            auto it = idx->columns.find(sort->slot);
            if (it == idx->columns.end()) return XGM_UNSUPPORTED;
            d_ord = (const uint32_t*)it->second.first;
        }
        if (collapse_slot >= 0) {
            auto it = idx->columns.find((uint32_t)collapse_slot);
            if (it == idx->columns.end()) return XGM_UNSUPPORTED;
            /* a row of counters per query: a column of many distinct keys under a large batch is left to single searches */
            if (((size_t)it->second.second + 1u) * nq > ((size_t)64 << 20)) return XGM_UNSUPPORTED;
            d_cord = (const uint32_t*)it->second.first;
            d_spy_ord = d_cord;
            n_counts = it->second.second + 1u;
            key_counts.assign((size_t)n_counts * nq, 0u);
            counts = key_counts.data();
        }
        if (spy_slot >= 0) {
            auto sp = idx->columns.find((uint32_t)spy_slot);
            if (sp == idx->columns.end()) return XGM_UNSUPPORTED;
            if (!counts || n_counts != sp->second.second + 1u) return xgm_set_error(XGM_E_INVALID, "n_counts %u, the spy column has %u distinct values", n_counts, sp->second.second);
            d_spy_ord = (const uint32_t*)sp->second.first;
        }
    }
    int rc = use_device(idx->device);
    if (rc) return rc;
    std::vector<xgm_dev_query> dq(nq);
    std::vector<uint32_t> kq(nq);
    std::vector<double> mp(nq);
    BatchPlan bp;
    if ((rc = plan_batch(idx, qs, nq, dq.data(), kq.data(), mp.data(), &bp, true))) return rc;
    if (bp.andw || bp.orw || bp.and_only || bp.cap > 8u * XGM_WG || bp.parts != 1u) return XGM_UNSUPPORTED;
    for (uint32_t i = 0; i < nq; ++i) if (dq[i].k == 0 || dq[i].k > k_stride) return dq[i].k ? xgm_set_error(XGM_E_INVALID, "k_stride %u < first + maxitems %u", k_stride, dq[i].k) : XGM_UNSUPPORTED;
    if (xgm_match_sorted_smem_bytes(idx->hdr.stripe_bits - bp.sub_bits, bp.tab_terms, bp.phrase, bp.cap, bp.wide, bp.stripes_per_group) > 160u * 1024u) return XGM_UNSUPPORTED;
    const uint32_t n_work = bp.n_work, kc = bp.k_stride_c;
    XgmScratch* sc;
    if ((rc = scratch_acquire(idx, &sc))) return rc;
    ScratchRelease release_{idx, sc};
    hipStream_t stream = sc->stream;
    const size_t o_q = 0, b_q = ((size_t)nq * sizeof(xgm_dev_query) + 15) & ~(size_t)15;
    const size_t o_wk = o_q + b_q, b_wk = ((size_t)n_work * sizeof(xgm_work) + 15) & ~(size_t)15;
    const size_t up_bytes = o_wk + b_wk;
    const size_t o_gh = up_bytes, b_gh = (size_t)n_work * sizeof(xgm_group_hdr);
    const size_t o_cd = o_gh + b_gh, b_cd = ((size_t)n_work * kc * sizeof(xgm_cand_sorted) + 15) & ~(size_t)15;
    const size_t o_ct = o_cd + b_cd, b_ct = d_spy_ord ? (size_t)nq * n_counts * 4 : 0;          /* the spy: one row of counts per query */
    const size_t total = o_ct + b_ct;
    if ((rc = grow(&sc->d_sorted, &sc->cap_sorted, total))) return rc;
    if ((rc = grow_pinned(&sc->h_sorted, &sc->cap_hsorted, total))) return rc;
    unsigned char* hb = (unsigned char*)sc->h_sorted;
    memcpy(hb + o_q, dq.data(), (size_t)nq * sizeof(xgm_dev_query));
    memcpy(hb + o_wk, bp.work.data(), (size_t)n_work * sizeof(xgm_work));
    HIP_TRY(hipMemcpyAsync(sc->d_sorted, hb, up_bytes, hipMemcpyHostToDevice, stream));
    uint32_t* d_counts = d_spy_ord ? (uint32_t*)(sc->d_sorted + o_ct) : nullptr;
    if (d_counts) HIP_TRY(hipMemsetAsync(d_counts, 0, b_ct, stream));
    xgm_match_launch L;
    L.seg = idx->view;
    L.queries = (xgm_dev_query*)(sc->d_sorted + o_q);
    L.nq = nq; L.n_work = n_work; L.work = (xgm_work*)(sc->d_sorted + o_wk); L.stripes_per_group = bp.stripes_per_group; L.sub_bits = bp.sub_bits;
    L.tab_terms = bp.tab_terms; L.cap = bp.cap; L.k_stride = kc;
    L.phrase = bp.phrase; L.wide = bp.wide; L.sided = 0;
    L.cand = nullptr; L.ghdr = (xgm_group_hdr*)(sc->d_sorted + o_gh);
    idx->last_kernel = "xgm_match_sorted_kernel";
    L.spy_stride = d_counts ? n_counts : 0u;
    if ((rc = xgm_launch_match_sorted(L, d_ord, mode, reverse ? 1u : 0u, d_spy_ord, d_counts, d_cord, d_cord ? cmax : 0u, (xgm_cand_sorted*)(sc->d_sorted + o_cd), stream))) return rc;
    HIP_TRY(hipMemcpyAsync(hb + o_gh, sc->d_sorted + o_gh, total - o_gh, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (d_counts) memcpy(counts, hb + o_ct, b_ct);
    const xgm_group_hdr* gh = (const xgm_group_hdr*)(hb + o_gh);
    const xgm_cand_sorted* cand = (const xgm_cand_sorted*)(hb + o_cd);
    const bool use_x = mode == XGM_SORT_VALUE_RELEVANCE || mode == XGM_SORT_RELEVANCE_VALUE;
    const bool weight_first = mode >= XGM_SORT_RELEVANCE_VALUE;
    std::vector<xgm_cand_sorted> all;
    for (uint32_t qi = 0; qi < nq; ++qi) {
        const uint32_t k = dq[qi].k;
        all.clear();
        uint64_t matches = 0, max_w = 0;
        uint32_t max_d = UINT32_MAX, max_m = 0;
        for (uint32_t u = bp.goff[qi]; u < bp.goff[qi + 1]; ++u) {
            const xgm_group_hdr& g = gh[u];
            if (g.n_cand > k) return xgm_set_error(XGM_E_DEVICE, "sorted batch: unit %u reports %u candidates for k = %u", u, g.n_cand, k);
            matches += g.matches;
            all.insert(all.end(), cand + (size_t)u * kc, cand + (size_t)u * kc + g.n_cand);
            if (g.c_pad[0] != UINT32_MAX && (max_d == UINT32_MAX || g.c_pos > max_w || (g.c_pos == max_w && g.c_pad[0] < max_d))) { max_w = g.c_pos; max_d = g.c_pad[0]; max_m = g.c_pad[1]; }
        }
        std::sort(all.begin(), all.end(), [&](const xgm_cand_sorted& a, const xgm_cand_sorted& b) {
            if (a.kw != b.kw) return a.kw > b.kw;
            if (use_x && a.kx != b.kx) return a.kx > b.kx;
            return a.did < b.did;
        });
        const uint32_t* kc_q = d_cord ? counts + (size_t)qi * n_counts : nullptr;        /* this query's matches per collapse key */
        if (d_cord) {
            /* the merged ranking collapsed once more: of every key the first cmax stay */
            std::vector<uint32_t> kept_of(n_counts, 0u);
            size_t keep = 0;
            for (const xgm_cand_sorted& c : all) {
                if (c.cord >= n_counts) return xgm_set_error(XGM_E_DEVICE, "sorted batch: collapse ordinal %u beyond the column", c.cord);
                if (c.cord == 0u || kept_of[c.cord]++ < cmax) all[keep++] = c;
            }
            all.resize(keep);
            if (collapsed_lb) {
                uint64_t lb = kc_q[0];                                       /* Collapser::get_matches_lower_bound */
                for (uint32_t o = 1; o < n_counts; ++o) lb += std::min<uint32_t>(kc_q[o], cmax);
                collapsed_lb[qi] = lb;
            }
        }
        const uint32_t n = (uint32_t)std::min<size_t>(k, all.size());
        xgm_hit* out = hits + (size_t)qi * k_stride;
        for (uint32_t i = 0; i < n; ++i) {
            const uint64_t wbits = weight_first ? all[i].kw : all[i].kx;
            const uint32_t okey = (uint32_t)(weight_first ? all[i].kx : all[i].kw);
            out[i].docid = all[i].did;
            out[i].subqs_matched = all[i].subqs;
            memcpy(&out[i].weight, &wbits, 8);
            if (hit_ord) hit_ord[(size_t)qi * k_stride + i] = sort ? (reverse ? okey : ~okey) : 0u;
            if (hit_cord) hit_cord[(size_t)qi * k_stride + i] = all[i].cord;
            if (hit_ccount) hit_ccount[(size_t)qi * k_stride + i] = (d_cord && all[i].cord && kc_q[all[i].cord] > cmax) ? kc_q[all[i].cord] - cmax : 0u;
        }
        xgm_result_hdr* hdr = &hdrs[qi];
        memset(hdr, 0, sizeof *hdr);
        hdr->n_hits = n;
        hdr->matches_exact = matches;
        hdr->max_possible = qs[qi].max_possible;
        if (max_d != UINT32_MAX) { memcpy(&hdr->max_attained, &max_w, 8); hdr->max_weight_subqs_matched = max_m; }
    }
    return XGM_OK;
}

extern "C" int xgm_search_sorted_batch(xgm_index* idx, const xgm_query* qs, uint32_t nq, const xgm_sort_spec* sort, uint32_t k_stride, xgm_hit* hits,
                                       uint32_t* hit_ord, xgm_result_hdr* hdrs) {
    return sorted_batch_core(idx, qs, nq, sort, k_stride, hits, hit_ord, hdrs, -1, nullptr, 0);
}

/* ... every search under Enquire::set_collapse_key(collapse_slot, collapse_max) (include/xgm.h: xgm_search_collapsed_batch) */
extern "C" int xgm_search_collapsed_batch(xgm_index* idx, const xgm_query* qs, uint32_t nq, const xgm_sort_spec* sort, uint32_t collapse_slot,
                                          uint32_t collapse_max, uint32_t k_stride, xgm_hit* hits, uint32_t* hit_ord, uint32_t* hit_collapse_ord,
                                          uint32_t* hit_collapse_count, xgm_result_hdr* hdrs, uint64_t* collapsed_lower_bound) {
    return sorted_batch_core(idx, qs, nq, sort, k_stride, hits, hit_ord, hdrs, -1, nullptr, 0, (int)collapse_slot, collapse_max, hit_collapse_ord,
                             hit_collapse_count, collapsed_lower_bound);
}

/* ... every search with a ValueCountMatchSpy on spy_slot (include/xgm.h: xgm_search_sorted_spy_batch): counts [nq][n_counts] */
extern "C" int xgm_search_sorted_spy_batch(xgm_index* idx, const xgm_query* qs, uint32_t nq, const xgm_sort_spec* sort, uint32_t k_stride, xgm_hit* hits,
                                           uint32_t* hit_ord, xgm_result_hdr* hdrs, uint32_t spy_slot, uint32_t* counts, uint32_t n_counts) {
    if (!counts) return xgm_set_error(XGM_E_INVALID, "null argument");
    return sorted_batch_core(idx, qs, nq, sort, k_stride, hits, hit_ord, hdrs, (int)spy_slot, counts, n_counts);
}

extern "C" int xgm_search_sorted_spy(xgm_index* idx, const xgm_query* q, const xgm_sort_spec* sort, xgm_hit* hits, uint32_t* hit_ord,
                                     xgm_result_hdr* hdr, uint32_t spy_slot, uint32_t* counts, uint32_t n_counts) {
    return sorted_core(idx, q, sort, hits, hit_ord, hdr, (int)spy_slot, counts, n_counts);
}

extern "C" int xgm_search_collapsed(xgm_index* idx, const xgm_query* q, const xgm_sort_spec* sort, uint32_t collapse_slot, uint32_t collapse_max,
                                    xgm_hit* hits, uint32_t* hit_ord, uint32_t* hit_collapse_ord, uint32_t* hit_collapse_count, xgm_result_hdr* hdr,
                                    uint64_t* collapsed_lower_bound) {
    return sorted_core(idx, q, sort, hits, hit_ord, hdr, -1, nullptr, 0, (int)collapse_slot, collapse_max, hit_collapse_ord, hit_collapse_count, collapsed_lower_bound);
}

/* ---- every match of a query, in docid order (include/xgm.h: xgm_search_all) -------------------------------------------------
 * The workgroup kernel (every query shape; it decodes the posting blocks of each stripe: K1 at full size) weighs every matching
 * document anyway when it runs under a sort; here it also appends each to one list (a wave-aggregated atomic per round), which
 * xgm_all.hip then puts in docid order by RANK (a bitmap of the matches, prefix counts, one scatter).  One query per call, synchronous. */
/* The device half both entry points share: the whole match of `q` in ascending docid order, left in HBM (sc->d_all).  list_conj: a
 * positional query's list also carries the documents of the underlying conjunction that fail the positional test, flagged
 * XGM_ALL_NOT_A_MATCH (what the frozen-weight replay needs).  *n_list = entries of the list, *n_matches = matching documents among them,
 * hdr = the whole match's figures.  *d_list = NULL when the list does not fit cap_dev entries (nothing packed). */
static int search_all_device(xgm_index* idx, const xgm_query* q, XgmScratch* sc, bool list_conj, uint64_t cap_dev, xgm_hit** d_list, uint64_t* n_list,
                             uint64_t* n_matches, xgm_result_hdr* hdr, size_t extra_bytes, unsigned char** d_extra) {
    *d_list = nullptr; *n_list = 0; *n_matches = 0;
    int rc;
    xgm_query q1 = *q;
    q1.first = 0; q1.maxitems = 1; q1.check_at_least = 0xFFFFFFFFu;          /* the kernel's own top-k is not used: keep it smallest; positions of every candidate tested */
    xgm_dev_query dq;
    uint32_t kq = 0;
    double mp = 0;
    BatchPlan bp;
    if ((rc = plan_batch(idx, &q1, 1, &dq, &kq, &mp, &bp, true))) return rc;
    if (bp.andw || bp.orw || bp.and_only || bp.cap > 8u * XGM_WG) return XGM_UNSUPPORTED;
    if (xgm_match_sorted_smem_bytes(idx->hdr.stripe_bits - bp.sub_bits, bp.tab_terms, bp.phrase, bp.cap, bp.wide, bp.stripes_per_group) > 160u * 1024u) return XGM_UNSUPPORTED;
    dq.flags &= ~XGM_QF_POSPRUNE;
    if (list_conj && (dq.flags & XGM_QF_PHRASE)) dq.flags |= XGM_QF_LIST_CONJ;
    const uint32_t n_work = bp.n_work;
    hipStream_t stream = sc->stream;
    /* the sorted kernel's own buffers: [query | work list] up, [unit headers | unit candidates (k = 1)] */
    const size_t o_q = 0, b_q = (sizeof dq + 15) & ~(size_t)15;
    const size_t o_wk = o_q + b_q, b_wk = ((size_t)n_work * sizeof(xgm_work) + 15) & ~(size_t)15;
    const size_t up_bytes = o_wk + b_wk;
    const size_t o_gh = up_bytes, b_gh = (size_t)n_work * sizeof(xgm_group_hdr);
    const size_t o_cd = o_gh + b_gh, b_cd = (size_t)n_work * sizeof(xgm_cand_sorted);
    const size_t total = o_cd + b_cd;
    if ((rc = grow(&sc->d_sorted, &sc->cap_sorted, total))) return rc;
    if ((rc = grow_pinned(&sc->h_sorted, &sc->cap_hsorted, total + 16))) return rc;
    /* the list: [counter 16 B | keys | weights | hits | the ordering's bitmap and prefix counts | the caller's extra bytes] */
    const size_t tmp_bytes = xgm_all_order_bytes(idx->hdr.lastdocid);
    const size_t a_cnt = 0, a_k0 = 16, a_v0 = a_k0 + cap_dev * 8, a_hit = a_v0 + cap_dev * 8;
    const size_t a_tmp = (a_hit + cap_dev * sizeof(xgm_hit) + 255) & ~(size_t)255, a_ext = (a_tmp + tmp_bytes + 255) & ~(size_t)255, a_total = a_ext + extra_bytes;
    if ((rc = grow(&sc->d_all, &sc->cap_all, a_total))) return rc;
    if (d_extra) *d_extra = sc->d_all + a_ext;
    unsigned char* hb = (unsigned char*)sc->h_sorted;
    memcpy(hb + o_q, &dq, sizeof dq);
    memcpy(hb + o_wk, bp.work.data(), (size_t)n_work * sizeof(xgm_work));
    HIP_TRY(hipMemcpyAsync(sc->d_sorted, hb, up_bytes, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemsetAsync(sc->d_all + a_cnt, 0, 16, stream));
    xgm_match_launch L;
    L.seg = idx->view;
    L.queries = (xgm_dev_query*)(sc->d_sorted + o_q);
    L.nq = 1; L.n_work = n_work; L.work = (xgm_work*)(sc->d_sorted + o_wk); L.stripes_per_group = bp.stripes_per_group; L.sub_bits = bp.sub_bits;
    L.tab_terms = bp.tab_terms; L.cap = bp.cap; L.k_stride = 1;
    L.phrase = bp.phrase; L.wide = bp.wide; L.sided = 0;
    L.cand = nullptr; L.ghdr = (xgm_group_hdr*)(sc->d_sorted + o_gh);
    unsigned long long* d_cnt = (unsigned long long*)(sc->d_all + a_cnt);
    unsigned long long* k0 = (unsigned long long*)(sc->d_all + a_k0);
    unsigned long long* v0 = (unsigned long long*)(sc->d_all + a_v0);
    idx->last_kernel = "xgm_match_sorted_kernel";
    if ((rc = xgm_launch_match_sorted(L, nullptr, 4u, 0u, nullptr, nullptr, nullptr, 0u, (xgm_cand_sorted*)(sc->d_sorted + o_cd), stream, k0, v0, d_cnt, cap_dev))) return rc;
    unsigned long long* h_cnt = (unsigned long long*)(hb + total);
    HIP_TRY(hipMemcpyAsync(hb + o_gh, sc->d_sorted + o_gh, b_gh, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(h_cnt, d_cnt, 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    const uint64_t n = *h_cnt;
    /* the whole match's count and best weight from the unit headers (the same fields xgm_search_sorted reports) */
    const xgm_group_hdr* gh = (const xgm_group_hdr*)(hb + o_gh);
    uint64_t matches = 0, max_w = 0;
    uint32_t max_d = UINT32_MAX, max_m = 0;
    for (uint32_t u = 0; u < n_work; ++u) {
        const xgm_group_hdr& g = gh[u];
        matches += g.matches;
        if (g.c_pad[0] != UINT32_MAX && (max_d == UINT32_MAX || g.c_pos > max_w || (g.c_pos == max_w && g.c_pad[0] < max_d))) { max_w = g.c_pos; max_d = g.c_pad[0]; max_m = g.c_pad[1]; }
    }
    if ((dq.flags & XGM_QF_LIST_CONJ) ? matches > n : matches != n)
        return xgm_set_error(XGM_E_DEVICE, "xgm_search_all: %llu documents listed, %llu counted", (unsigned long long)n, (unsigned long long)matches);
    memset(hdr, 0, sizeof *hdr);
    hdr->matches_exact = matches;
    hdr->max_possible = q->max_possible;
    if (max_d != UINT32_MAX) { memcpy(&hdr->max_attained, &max_w, 8); hdr->max_weight_subqs_matched = max_m; }
    *n_list = n; *n_matches = matches;
    if (n > cap_dev || n == 0) return XGM_OK;                       /* does not fit (the caller decides what that means) / nothing to order */
    xgm_hit* d_out = (xgm_hit*)(sc->d_all + a_hit);
    if ((rc = xgm_all_order_pack(sc->d_all + a_tmp, idx->hdr.lastdocid, k0, v0, (size_t)n, d_out, stream))) return rc;
    *d_list = d_out;
    return XGM_OK;
}


extern "C" int xgm_search_all(xgm_index* idx, const xgm_query* q, xgm_hit* hits, uint64_t cap, uint64_t* n_matches, xgm_result_hdr* hdr) {
    if (!idx || !q || !n_matches || !hdr || (cap && !hits)) return xgm_set_error(XGM_E_INVALID, "null argument");
    if (idx->device == XGM_DEVICE_NONE) return xgm_set_error(XGM_E_NO_DEVICE, "index opened without a device");
    *n_matches = 0;
    int rc = use_device(idx->device);
    if (rc) return rc;
    /* no more documents can match than the tree's own upper bound (the reference's get_termfreq_max, which the planner restates) —
     * nor than the shard holds, nor than the caller has room for */
    uint64_t cap_dev = std::min<uint64_t>(cap, idx->hdr.doccount);
    if (q->est_max) cap_dev = std::min<uint64_t>(cap_dev, q->est_max);
    XgmScratch* sc;
    if ((rc = scratch_acquire(idx, &sc))) return rc;
    ScratchRelease release_{idx, sc};
    xgm_hit* d_out = nullptr;
    uint64_t n = 0, m = 0;
    if ((rc = search_all_device(idx, q, sc, false, cap_dev, &d_out, &n, &m, hdr, 0, nullptr))) return rc;
    *n_matches = n;
    if (n > cap_dev) {
        if (n > cap) return XGM_OK;                                /* the caller's buffer is too small: nothing written, *n_matches says how many there are */
        return xgm_set_error(XGM_E_DEVICE, "xgm_search_all: %llu matches exceed the plan's upper bound %u", (unsigned long long)n, q->est_max);
    }
    if (n == 0) return XGM_OK;
    HIP_TRY(hipMemcpyAsync(hits, d_out, (size_t)n * sizeof(xgm_hit), hipMemcpyDeviceToHost, sc->stream));
    HIP_TRY(hipStreamSynchronize(sc->stream));
    hdr->n_hits = (uint32_t)n;
    return XGM_OK;
}

/* ---- the reference's collation of a search by relevance, replayed on the device (include/xgm.h: xgm_search_replay) ----------
 * search_all_device leaves the match in docid order in HBM; xgm_replay_kernel (xgm_replay.hip) walks it as ProtoMSet would.  Only the
 * page (first + maxitems hits) and 48 bytes of figures cross PCIe. */
constexpr uint32_t kReplaySegments = 512u;           /* at most: one wave each; the states' exclusive scan is sequential over them */
static std::atomic<uint64_t> g_replays[2];          /* diagnostics: replays walked by one workgroup / by segments in parallel */
extern "C" int xgm_debug_replay_info(uint64_t* out2) { if (!out2) return -1; out2[0] = g_replays[0].load(); out2[1] = g_replays[1].load(); return 0; }

extern "C" int xgm_search_replay(xgm_index* idx, const xgm_query* q, uint32_t mode, xgm_hit* hits, xgm_result_hdr* hdr, uint64_t* known_matching_docs) {
    if (!idx || !q || !hdr || !known_matching_docs || mode > XGM_REPLAY_FROZEN_WEIGHT) return xgm_set_error(XGM_E_INVALID, "bad argument");
    if (idx->device == XGM_DEVICE_NONE) return xgm_set_error(XGM_E_NO_DEVICE, "index opened without a device");
    const uint64_t k64 = (uint64_t)q->first + q->maxitems;
    if (k64 > XGM_MAX_K) return XGM_UNSUPPORTED;
    const uint32_t k = (uint32_t)k64;
    if (k && !hits) return xgm_set_error(XGM_E_INVALID, "null hits");
    const bool positional = (q->op == XGM_OP_PHRASE || q->op == XGM_OP_NEAR) && q->phrase_active;
    if (mode == XGM_REPLAY_FROZEN_WEIGHT && !positional) return xgm_set_error(XGM_E_INVALID, "the frozen weight is a positional query's");
    *known_matching_docs = 0;
    int rc = use_device(idx->device);
    if (rc) return rc;
    /* room: the tree's own upper bound (for the frozen weight: the underlying conjunction's, which is what a positional tree reports) */
    uint64_t cap_dev = idx->hdr.doccount;
    if (q->est_max) cap_dev = std::min<uint64_t>(cap_dev, q->est_max);
    cap_dev = std::max<uint64_t>(cap_dev, 1);
    XgmScratch* sc;
    if ((rc = scratch_acquire(idx, &sc))) return rc;
    ScratchRelease release_{idx, sc};
    xgm_hit* d_list = nullptr;
    uint64_t n = 0, m = 0;
    unsigned char* d_ext = nullptr;
    /* the parallel formulation (xgm_replay.hip: segments replayed from the prefix's top k) takes the counting mode whenever ProtoMSet's state is a
     * function of the prefix: check_at_least within the page; scratch for up to 256 segment states rides behind the page (XGM_REPLAY_SERIAL=1: A/B) */
    static const bool serial_only = getenv("XGM_REPLAY_SERIAL") != nullptr;
    const bool may_parallel = mode == XGM_REPLAY_COUNT && k >= 1u && q->check_at_least <= k && !serial_only;
    const size_t b_hits = ((size_t)std::max(k, 1u) * sizeof(xgm_hit) + 255) & ~(size_t)255, b_down = b_hits + 256;
    const size_t ext = b_down + (may_parallel ? xgm_replay_parallel_bytes(kReplaySegments, k) : 0);
    if ((rc = search_all_device(idx, q, sc, mode == XGM_REPLAY_FROZEN_WEIGHT, cap_dev, &d_list, &n, &m, hdr, ext, &d_ext))) return rc;
    if (n > cap_dev) return xgm_set_error(XGM_E_DEVICE, "xgm_search_replay: %llu documents exceed the plan's upper bound %u", (unsigned long long)n, q->est_max);
    hdr->n_hits = 0;
    if (n == 0) return XGM_OK;
    xgm_hit* d_page = (xgm_hit*)d_ext;
    xgm_replay_out* d_out = (xgm_replay_out*)(d_ext + b_hits);
    unsigned long long* d_known = (unsigned long long*)(d_ext + b_hits + 128);
    /* segments of >= 4 096 entries (XGM_REPLAY_SEG_MIN: the tests run small lists through the parallel path), longer than the page, at least four */
    static const uint64_t seg_min_env = getenv("XGM_REPLAY_SEG_MIN") ? (uint64_t)std::max(64, atoi(getenv("XGM_REPLAY_SEG_MIN"))) : 4096u;
    const uint64_t seg_min = std::max<uint64_t>(seg_min_env, (uint64_t)k + 64u);
    const bool parallel = may_parallel && n >= 4u * seg_min;
    ++g_replays[parallel ? 1 : 0];
    if (parallel) {
        const uint32_t S = (uint32_t)std::min<uint64_t>(kReplaySegments, n / seg_min);
        const uint64_t seg_len = (((n + S - 1u) / S) + 63u) & ~(uint64_t)63;
        if ((rc = xgm_launch_replay_parallel(d_list, n, k, q->check_at_least, S, seg_len, d_ext + b_down, d_page, d_out, d_known, sc->stream))) return rc;
    } else if ((rc = xgm_launch_replay(d_list, n, k, q->check_at_least, mode == XGM_REPLAY_FROZEN_WEIGHT, m, d_page, d_out, sc->stream))) return rc;
    if ((rc = grow_pinned(&sc->h_down, &sc->cap_down, b_down))) return rc;
    HIP_TRY(hipMemcpyAsync(sc->h_down, d_ext, b_down, hipMemcpyDeviceToHost, sc->stream));
    HIP_TRY(hipStreamSynchronize(sc->stream));
    const xgm_replay_out* o = (const xgm_replay_out*)((unsigned char*)sc->h_down + b_hits);
    if (o->n_hits > k) return xgm_set_error(XGM_E_DEVICE, "xgm_search_replay: %u documents kept for a page of %u", o->n_hits, k);
    memcpy(hits, sc->h_down, (size_t)o->n_hits * sizeof(xgm_hit));
    hdr->n_hits = o->n_hits;
    if (parallel) {
        /* (the counting mode shows ProtoMSet the true weights: its max_weight is the match's, already in hdr from the unit headers) */
        *known_matching_docs = *(const unsigned long long*)((unsigned char*)sc->h_down + b_hits + 128);
        return XGM_OK;
    }
    *known_matching_docs = o->known_matching_docs;
    /* ProtoMSet's own max_weight (update_max_weight sees what add() is shown: the frozen weight where it was served) */
    hdr->max_attained = o->max_weight;
    hdr->max_weight_subqs_matched = o->max_weight_subqs;
    return XGM_OK;
}

/* ---- opt-in micro-batching (server mode) -----------------------------------------------------------------------------
 * Xapiand's HTTP worker threads issue ONE get_mset each (reference src/manager.cc:161, src/database/handler.cc:1338);
 * a GPU wants hundreds of queries per launch.  With xgm_index_set_batching(idx, max_batch) single-query calls are
 * queued, and a dispatcher thread owned by the index launches whatever has accumulated — up to max_batch — as one
 * heterogeneous batch, then hands every caller its rows.  "Natural" batching: nothing waits on a timer; while one batch
 * runs the next one fills, so batches grow with the offered load and an idle server adds no latency. */
struct XgmBatchReq {
    const xgm_query* q; uint32_t k_stride; xgm_hit* hits; xgm_result_hdr* hdr; uint64_t* known = nullptr;
    int rc = 0; bool done = false; char err[192];
    std::condition_variable cv;                 /* this request's own: a finished batch wakes exactly its callers */
};

struct XgmFlight {                              /* a launched batch and the requests it answers */
    xgm_inflight* f = nullptr;
    std::vector<XgmBatchReq*> reqs;
    uint32_t ks = 1;
};

struct XgmBatcher {
    std::mutex mu;
    std::condition_variable cv_work, cv_flight, cv_room;
    std::deque<XgmBatchReq*> queue;
    std::deque<XgmFlight*> flights;             /* launched, not yet handed back: at most max_flights */
    std::thread th, th_done;
    hipStream_t stream = nullptr;               /* the flights' match kernels run back to back on ONE stream (kernels of concurrent streams slow each other
                                                   down: measured 0.53 vs 0.38 ms per 256-query launch); uploads and downloads ride the scratches' own streams */
    bool stop = false;
    bool dispatcher_done = false;               /* the dispatcher has left its loop: no batch can be cut any more (the completer may then leave too) */
    uint32_t max_batch = 256;
    uint32_t max_flights = 2;
    uint64_t batches = 0, requests = 0;
    uint32_t inflight_reqs = 0;                 /* requests of the launched flights */
    double peak_outstanding = 0.0;              /* queued + in flight, a slowly decaying maximum: how many callers there are */
    uint32_t linger_us = 0;
};

static void batcher_finish(XgmBatcher* b, XgmBatchReq* r, int rc, const char* err) {
    std::lock_guard<std::mutex> lk(b->mu);
    r->rc = rc;
    if (rc < 0 && err) snprintf(r->err, sizeof r->err, "%s", err);
    r->done = true;
    r->cv.notify_one();
}

/* the dispatcher: whatever has accumulated becomes one batch, launched asynchronously; while it runs the next one fills — and is
 * launched too (up to max_flights batches in flight, each on its own stream) instead of waiting for the first to finish */
static void batcher_loop(xgm_index* idx, XgmBatcher* b) {
    std::vector<XgmBatchReq*> take;
    std::vector<xgm_query> qs;
    while (true) {
        {
            std::unique_lock<std::mutex> lk(b->mu);
            b->cv_work.wait(lk, [&] { return b->stop || !b->queue.empty(); });
            if (b->stop && b->queue.empty()) break;
            b->cv_room.wait(lk, [&] { return b->flights.size() < b->max_flights; });
            /* While a flight is on the GPU the next one may as well wait for the callers that are about to come back: the flights run back to
             * back on one stream and a launch costs ~50 us + ~1.2 us per query, so 64 callers in two flights of 32 get through faster than in
             * three groups of 21 (two in flight, one queued — what taking whatever has accumulated settles into; measured 189 k queries/s).
             * Linger until everyone who is not in a flight has queued up, a flight has come back (the GPU is about to idle), or linger_us. */
            const double outstanding = (double)(b->queue.size() + b->inflight_reqs);
            b->peak_outstanding = std::max(outstanding, b->peak_outstanding * 0.98);
            if (b->linger_us && !b->flights.empty() && !b->stop) {
                const size_t flights0 = b->flights.size();
                const double want = std::min<double>(b->max_batch, b->peak_outstanding - (double)b->inflight_reqs);
                if ((double)b->queue.size() < want)
                    b->cv_work.wait_for(lk, std::chrono::microseconds(b->linger_us),
                                        [&] { return b->stop || (double)b->queue.size() >= want || b->flights.size() < flights0; });
            }
            take.clear();
            while (!b->queue.empty() && take.size() < b->max_batch) { take.push_back(b->queue.front()); b->queue.pop_front(); }
        }
        const uint32_t n = (uint32_t)take.size();
        uint32_t ks = 1;
        qs.resize(n);
        for (uint32_t i = 0; i < n; ++i) { qs[i] = *take[i]->q; ks = std::max(ks, take[i]->k_stride); }
        XgmFlight* fl = new XgmFlight();
        fl->reqs = take; fl->ks = ks;
        /* the dispatcher's batches are small (a few dozen queries: the GPU is not the bottleneck, the calls per batch are): upload, match and
         * download in stream order — four HIP calls fewer than with the copies on streams of their own (XGM_BATCHER_COPY_STREAMS=1: A/B) */
        static const bool copy_streams = getenv("XGM_BATCHER_COPY_STREAMS") != nullptr;
        int rc = batch_begin(idx, qs.data(), n, ks, &fl->f, b->stream, !copy_streams);
        if (rc != XGM_OK) {
            /* one query of the batch was declined or failed: answer each on its own so that only that caller sees it */
            delete fl;
            for (uint32_t i = 0; i < n; ++i) {
                const int r1 = n > 1 ? search_batch_now(idx, take[i]->q, 1, take[i]->k_stride, take[i]->hits, take[i]->hdr, take[i]->known) : rc;
                batcher_finish(b, take[i], r1, xgm_last_error());
            }
            std::lock_guard<std::mutex> lk(b->mu);
            ++b->batches; b->requests += n;
            continue;
        }
        {
            std::lock_guard<std::mutex> lk(b->mu);
            b->flights.push_back(fl);
            b->inflight_reqs += n;
            ++b->batches; b->requests += n;
        }
        b->cv_flight.notify_one();
    }
    { std::lock_guard<std::mutex> lk(b->mu); b->stop = true; b->dispatcher_done = true; }
    b->cv_flight.notify_all();
}

/* the completer: waits for the oldest batch in flight, hands every caller its rows, wakes it */
static void batcher_done_loop(xgm_index* idx, XgmBatcher* b) {
    (void)idx;
    while (true) {
        XgmFlight* fl = nullptr;
        {
            std::unique_lock<std::mutex> lk(b->mu);
            /* (not `stop && queue.empty()`: the dispatcher may hold a batch it has cut but not yet put in flight — ADVICE r4) */
            b->cv_flight.wait(lk, [&] { return !b->flights.empty() || b->dispatcher_done; });
            if (b->flights.empty()) return;
            fl = b->flights.front();
        }
        const xgm_hit* hh = nullptr;
        const xgm_result_hdr* hd = nullptr;
        const int rc = xgm_batch_end(fl->f, &hh, &hd);
        const std::string err = rc < 0 ? xgm_last_error() : "";
        const uint32_t n = (uint32_t)fl->reqs.size();
        if (rc == XGM_OK) {
            const uint64_t* kn = nullptr;
            xgm_batch_known(fl->f, &kn);
            for (uint32_t i = 0; i < n; ++i) {
                *fl->reqs[i]->hdr = hd[i];
                memcpy(fl->reqs[i]->hits, hh + (size_t)i * fl->ks, (size_t)hd[i].n_hits * sizeof(xgm_hit));
                if (fl->reqs[i]->known) *fl->reqs[i]->known = kn ? kn[i] : 0;
            }
        }
        xgm_batch_release(fl->f);
        {
            std::lock_guard<std::mutex> lk(b->mu);
            b->flights.pop_front();
            b->inflight_reqs -= n;
            for (XgmBatchReq* r : fl->reqs) {
                r->rc = rc;
                if (rc < 0) snprintf(r->err, sizeof r->err, "%s", err.c_str());
                r->done = true;
                r->cv.notify_one();
            }
        }
        b->cv_room.notify_one();
        b->cv_work.notify_one();                   /* (a lingering dispatcher: the GPU is about to idle) */
        delete fl;
    }
}

static int batcher_submit(xgm_index* idx, const xgm_query* q, uint32_t k_stride, xgm_hit* hits, xgm_result_hdr* hdr, uint64_t* known) {
    XgmBatcher* b = idx->batcher;
    XgmBatchReq r;
    r.q = q; r.k_stride = k_stride; r.hits = hits; r.hdr = hdr; r.known = known; r.err[0] = 0;
    {
        std::unique_lock<std::mutex> lk(b->mu);
        b->queue.push_back(&r);
        b->cv_work.notify_one();                   /* always: a LINGERING dispatcher waits for the queue to reach a size, not to become non-empty (one futex call) */
        r.cv.wait(lk, [&] { return r.done; });
    }
    if (r.rc < 0) return xgm_set_error(r.rc, "%s", r.err);
    return r.rc;
}

void xgm_batcher_destroy(xgm_index* idx) {
    XgmBatcher* b = idx->batcher;
    if (!b) return;
    { std::lock_guard<std::mutex> lk(b->mu); b->stop = true; }
    b->cv_work.notify_all();
    b->cv_room.notify_all();
    if (b->th.joinable()) b->th.join();
    b->cv_flight.notify_all();
    if (b->th_done.joinable()) b->th_done.join();
    if (b->stream) { hipStreamSynchronize(b->stream); hipStreamDestroy(b->stream); }
    idx->batcher = nullptr;
    delete b;
}

extern "C" int xgm_index_set_batching(xgm_index* idx, uint32_t max_batch) {
    if (!idx) return xgm_set_error(XGM_E_INVALID, "null argument");
    if (idx->device == XGM_DEVICE_NONE) return xgm_set_error(XGM_E_NO_DEVICE, "the index was opened without a device");
    xgm_batcher_destroy(idx);
    if (max_batch == 0) return XGM_OK;
    XgmBatcher* b = new XgmBatcher();
    b->max_batch = std::min<uint32_t>(max_batch, 1024u);
    static const uint32_t flights_env = getenv("XGM_BATCHER_FLIGHTS") ? (uint32_t)std::max(1, atoi(getenv("XGM_BATCHER_FLIGHTS"))) : 2u;   /* A/B switch (1 = rounds 1-3; measured at 64 threads: 1 → 110 k, 2 → 195 k, 3 → 162 k queries/s) */
    b->max_flights = std::min(flights_env, 6u);
    static const int linger_env = getenv("XGM_BATCHER_LINGER_US") ? atoi(getenv("XGM_BATCHER_LINGER_US")) : 0;        /* A/B switch, off: measured at 64 threads 193 k queries/s (mean batch 21) without, 179 k (28) at 40 us, 162 k at 80 us — the flights are not bound by the GPU's share */
    b->linger_us = (uint32_t)std::max(0, linger_env);
    {
        int rc = use_device(idx->device);
        if (rc) { delete b; return rc; }
        hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete b; return xgm_launch_error("hipStreamCreate", (int)e, hipGetErrorString(e)); }
    }
    idx->batcher = b;
    b->th = std::thread(batcher_loop, idx, b);
    b->th_done = std::thread(batcher_done_loop, idx, b);
    return XGM_OK;
}

/* out3 = {batches launched, requests served, current max_batch} */
extern "C" int xgm_debug_batching_info(const xgm_index* idx, uint64_t* out3) {
    if (!idx || !out3 || !idx->batcher) return xgm_set_error(XGM_E_INVALID, "batching is off");
    std::lock_guard<std::mutex> lk(idx->batcher->mu);
    out3[0] = idx->batcher->batches; out3[1] = idx->batcher->requests; out3[2] = idx->batcher->max_batch;
    return XGM_OK;
}

/* Diagnostics / bench: n_threads host threads, each answering `per_thread` queries ONE AT A TIME through
 * xgm_get_mset_batch(nq = 1) — plan + search, the call a matcher hook makes — picking descs[(t * per_thread + i) % n].
 * lat_us receives n_threads * per_thread latencies (microseconds); returns the wall time in seconds (< 0 on error). */
extern "C" double xgm_debug_concurrent_searches(xgm_index* idx, const xgm_query_desc* descs, const xgm_global_stats* gs, uint32_t n,
                                                uint32_t n_threads, uint32_t per_thread, uint32_t k, double* lat_us) {
    if (!idx || !descs || !n || !n_threads || !per_thread || !lat_us) return -1.0;
    std::vector<std::thread> th;
    std::atomic<int> bad{0};
    std::atomic<uint32_t> ready{0};
    std::atomic<bool> go{false};
    for (uint32_t t = 0; t < n_threads; ++t) {
        th.emplace_back([&, t]() {
            std::vector<xgm_hit> hits(k ? k : 1);
            xgm_result_hdr hdr;
            ++ready;
            while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
            for (uint32_t i = 0; i < per_thread; ++i) {
                const uint32_t j = (t * per_thread + i) % n;
                const auto a = std::chrono::steady_clock::now();
                const int rc = xgm_get_mset_batch(idx, &descs[j], gs ? &gs[j] : nullptr, 1, k ? k : 1, hits.data(), &hdr);
                lat_us[(size_t)t * per_thread + i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
                if (rc != XGM_OK) ++bad;
            }
        });
    }
    while (ready.load() < n_threads) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    for (auto& x : th) x.join();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return bad.load() ? -1.0 : wall;
}

extern "C" int xgm_search(xgm_index* idx, const xgm_query* q, xgm_hit* hits, xgm_result_hdr* hdr) {
    if (!q) return xgm_set_error(XGM_E_INVALID, "null argument");
    uint32_t k = q->first + q->maxitems;
    return xgm_search_batch(idx, q, 1, k ? k : 1, hits, hdr);
}

/* Enqueue a batch on `stream` (NULL: the index's stream, or the scratch's own one followed by a host wait). */
static int search_batch_device_on(xgm_index* idx, hipStream_t on, const xgm_query* qs, uint32_t nq, uint32_t k_stride, void* d_hits,
                                  void* d_hdrs) {
    if (nq == 0) return XGM_OK;
    int rc = use_device(idx->device);
    if (rc) return rc;
    XgmScratch* s;
    if ((rc = scratch_acquire(idx, &s))) return rc;
    hipStream_t stream = on ? on : pick_stream(idx, s);
    rc = run_batch(idx, s, stream, qs, nq, k_stride, (xgm_hit*)d_hits, (xgm_result_hdr*)d_hdrs);
    /* the scratch (queries, candidates) is still in use by the enqueued kernels: mark it pending so
     * the next acquire waits for them.  (Measured and rejected: running every batch on its scratch's own stream with
     * hand-over events to and from the caller's stream, so that consecutive batches overlap — the three cross-stream
     * waits per batch cost more than the overlap gains: 0.51 -> 0.94 ms per step.) */
    if (hipEventRecord(s->ev_done, stream) == hipSuccess) s->pending = true; else hipStreamSynchronize(stream);
    if (!on && !idx->stream) hipStreamSynchronize(stream);
    scratch_release(idx, s);
    return rc;
}

extern "C" int xgm_search_batch_device(xgm_index* idx, const xgm_query* qs, uint32_t nq, uint32_t k_stride, void* d_hits,
                                       void* d_hdrs) {
    if (!idx || !qs || !d_hits || !d_hdrs) return xgm_set_error(XGM_E_INVALID, "null argument");
    return search_batch_device_on(idx, nullptr, qs, nq, k_stride, d_hits, d_hdrs);
}

/* plan (xgm_plan_query per description, with that query's merged statistics when given) + search, results in HBM */
extern "C" int xgm_get_mset_batch_device(xgm_index* idx, const xgm_query_desc* descs, const xgm_global_stats* gs, uint32_t nq,
                                         uint32_t k_stride, void* d_hits, void* d_hdrs) {
    if (!idx || !descs) return xgm_set_error(XGM_E_INVALID, "null argument");
    static thread_local std::vector<xgm_query> plans;
    plans.resize(nq);
    const uint64_t t0 = now_ns();
    for (uint32_t i = 0; i < nq; ++i) {
        int rc = xgm_plan_query(idx, &descs[i], gs ? &gs[i] : nullptr, &plans[i]);
        if (rc) return rc;
    }
    g_host_ns[0] += now_ns() - t0;
    return xgm_search_batch_device(idx, plans.data(), nq, k_stride, d_hits, d_hdrs);
}

extern "C" int xgm_get_mset_batch(xgm_index* idx, const xgm_query_desc* descs, const xgm_global_stats* gs, uint32_t nq,
                                  uint32_t k_stride, xgm_hit* hits, xgm_result_hdr* hdrs) {
    if (!idx || !descs) return xgm_set_error(XGM_E_INVALID, "null argument");
    static thread_local std::vector<xgm_query> plans;
    plans.resize(nq);
    for (uint32_t i = 0; i < nq; ++i) {
        int rc = xgm_plan_query(idx, &descs[i], gs ? &gs[i] : nullptr, &plans[i]);
        if (rc) return rc;
    }
    return xgm_search_batch(idx, plans.data(), nq, k_stride, hits, hdrs);
}

static int merge_shards_device_on(xgm_index* idx, hipStream_t on, const void* d_all_hits, const void* d_all_hdrs, uint32_t n_shards,
                                  uint32_t nq, uint32_t k_stride, const uint32_t* k, void* d_out_hits, void* d_out_hdrs, size_t shard_record_bytes = 0);

extern "C" int xgm_merge_shards_device(xgm_index* idx, const void* d_all_hits, const void* d_all_hdrs, uint32_t n_shards,
                                       uint32_t nq, uint32_t k_stride, const uint32_t* k, void* d_out_hits, void* d_out_hdrs) {
    if (!idx || !d_all_hits || !d_all_hdrs || !k || !d_out_hits || !d_out_hdrs) return xgm_set_error(XGM_E_INVALID, "null argument");
    return merge_shards_device_on(idx, nullptr, d_all_hits, d_all_hdrs, n_shards, nq, k_stride, k, d_out_hits, d_out_hdrs);
}

/* the same after ONE all-gather of packed per-shard records (include/xgm.h) */
extern "C" int xgm_merge_shards_packed_device(xgm_index* idx, const void* d_all_records, uint32_t n_shards, uint32_t nq, uint32_t k_stride,
                                              const uint32_t* k, void* d_out_hits, void* d_out_hdrs) {
    if (!idx || !d_all_records || !k || !d_out_hits || !d_out_hdrs) return xgm_set_error(XGM_E_INVALID, "null argument");
    const size_t rec = xgm_shard_record_bytes(nq, k_stride);
    return merge_shards_device_on(idx, nullptr, d_all_records, (const unsigned char*)d_all_records + (size_t)nq * k_stride * sizeof(xgm_hit), n_shards, nq, k_stride, k,
                                  d_out_hits, d_out_hdrs, rec);
}

extern "C" size_t xgm_shard_record_bytes(uint32_t nq, uint32_t k_stride) { return (size_t)nq * ((size_t)k_stride * sizeof(xgm_hit) + sizeof(xgm_result_hdr)); }

static int merge_shards_device_on(xgm_index* idx, hipStream_t on, const void* d_all_hits, const void* d_all_hdrs, uint32_t n_shards,
                                  uint32_t nq, uint32_t k_stride, const uint32_t* k, void* d_out_hits, void* d_out_hdrs, size_t shard_record_bytes) {
    if (nq == 0) return XGM_OK;
    int rc = use_device(idx->device);
    if (rc) return rc;
    XgmScratch* s;
    if ((rc = scratch_acquire(idx, &s))) return rc;
    hipStream_t stream = on ? on : pick_stream(idx, s);
    do {
        uint32_t k_max = 1;
        for (uint32_t i = 0; i < nq; ++i) k_max = std::max(k_max, k[i]);
        if (k_max > k_stride) { rc = xgm_set_error(XGM_E_INVALID, "k > k_stride"); break; }
        uint32_t cap = std::max(512u, next_pow2(n_shards * k_max));
        if (cap > XGM_MERGE_CAP) { rc = XGM_UNSUPPORTED; break; }
        if ((rc = grow(&s->d_mkq, &s->cap_mkq, (size_t)nq))) break;
        if ((rc = grow_pinned(&s->h_up, &s->cap_up, (size_t)nq * 4))) break;
        memcpy(s->h_up, k, (size_t)nq * 4);
        hipError_t e = hipMemcpyAsync(s->d_mkq, s->h_up, (size_t)nq * 4, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) { rc = xgm_launch_error("hipMemcpyAsync", (int)e, hipGetErrorString(e)); break; }
        rc = xgm_launch_merge_shards((const xgm_hit*)d_all_hits, (const xgm_result_hdr*)d_all_hdrs, n_shards, nq, k_stride, s->d_mkq,
                                     cap, (xgm_hit*)d_out_hits, (xgm_result_hdr*)d_out_hdrs, stream, shard_record_bytes);
    } while (0);
    if (hipEventRecord(s->ev_done, stream) == hipSuccess) s->pending = true; else hipStreamSynchronize(stream);
    if (!on && !idx->stream) hipStreamSynchronize(stream);
    scratch_release(idx, s);
    return rc;
}

/* ---- several shards in one process ----------------------------------------------------------------------------
 * Persistent context of one shard list, owned by shards[0]: per DEVICE a stream, an event and the gathered arrays
 * ([n_shards][nq][k_stride] hits, [n_shards][nq] headers; the device's shards write their slices in place), and the
 * exchange step that brings every slice to shards[0]'s device:
 *   - RCCL: ncclAllGather over a communicator the library owns (ncclCommInitAll on the shard devices, librccl.so.1
 *     loaded with dlopen on first use so that a single-GPU host never needs it) when every shard has its own device —
 *     the 8 x MI355X layout of BASELINE.json's C4;
 *   - otherwise (several shards per device, or no RCCL): hipMemcpyPeerAsync of the remote slices, ordered by events.
 * Nothing is allocated and no device is synchronised per call: launches on all shards first, then one wait for the
 * merged result.  Calls on the same shard list are serialised by the context's mutex (the per-index scratch pools
 * keep independent lists concurrent). */
#include <dlfcn.h>

namespace {

typedef void* xgm_nccl_comm;
struct RcclApi {
    void* lib = nullptr;
    int (*CommInitAll)(xgm_nccl_comm*, int, const int*) = nullptr;
    int (*CommDestroy)(xgm_nccl_comm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, xgm_nccl_comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    bool ok = false;
};

RcclApi& rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, []() {
        const char* off = getenv("XGM_SHARDED_RCCL");
        if (off && off[0] == '0') return;
        /* RTLD_LOCAL: a host that already carries another copy of RCCL (PyTorch bundles one) keeps its own symbols */
        api.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!api.lib) api.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!api.lib) return;
        api.CommInitAll = (int (*)(xgm_nccl_comm*, int, const int*))dlsym(api.lib, "ncclCommInitAll");
        api.CommDestroy = (int (*)(xgm_nccl_comm))dlsym(api.lib, "ncclCommDestroy");
        api.AllGather = (int (*)(const void*, void*, size_t, int, xgm_nccl_comm, hipStream_t))dlsym(api.lib, "ncclAllGather");
        api.GroupStart = (int (*)())dlsym(api.lib, "ncclGroupStart");
        api.GroupEnd = (int (*)())dlsym(api.lib, "ncclGroupEnd");
        api.ok = api.CommInitAll && api.CommDestroy && api.AllGather && api.GroupStart && api.GroupEnd;
    });
    return api;
}

struct ShardDev {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;
    unsigned char* all_rec = nullptr;     /* the gathered array on this device: n_shards packed records — [nq][k_stride] hits then [nq] headers each (xgm_shard_record_bytes) —
                                             dense for the CALL's nq and k_stride: ONE all-gather / peer copy per shard and call (round 6; two before) */
    xgm_nccl_comm comm = nullptr;
};

}  // namespace

struct XgmShardCtx {
    std::mutex mu;
    std::vector<xgm_index*> shards;
    std::vector<int> dev_of;              /* shard -> index into devs */
    std::vector<ShardDev> devs;           /* devs[0] = shards[0]'s device */
    size_t cap_hit = 0, cap_nq = 0;       /* per-shard capacities of the gathered arrays */
    xgm_hit* d_out_hits = nullptr; xgm_result_hdr* d_out_hdrs = nullptr;
    void* h_out = nullptr; size_t cap_hout = 0;
    bool use_rccl = false, rccl_tried = false;
    uint64_t calls = 0, rccl_calls = 0;
};

static void shard_ctx_free_buffers(XgmShardCtx* c) {
    for (ShardDev& d : c->devs) {
        if (hipSetDevice(d.device) != hipSuccess) continue;
        if (d.all_rec) hipFree(d.all_rec);
        d.all_rec = nullptr;
    }
    if (!c->devs.empty() && hipSetDevice(c->devs[0].device) == hipSuccess) {
        if (c->d_out_hits) hipFree(c->d_out_hits);
        if (c->d_out_hdrs) hipFree(c->d_out_hdrs);
    }
    c->d_out_hits = nullptr; c->d_out_hdrs = nullptr;
    c->cap_hit = c->cap_nq = 0;
}

void xgm_shard_ctx_destroy(XgmShardCtx* c) {
    if (!c) return;
    shard_ctx_free_buffers(c);
    for (ShardDev& d : c->devs) {
        if (hipSetDevice(d.device) != hipSuccess) continue;
        if (d.comm && rccl_api().ok) rccl_api().CommDestroy(d.comm);
        if (d.ev) hipEventDestroy(d.ev);
        if (d.stream) hipStreamDestroy(d.stream);
    }
    if (c->h_out) hipHostFree(c->h_out);
    delete c;
}

static int shard_ctx_get(xgm_index* const* shards, uint32_t n_shards, XgmShardCtx** out) {
    xgm_index* owner = shards[0];
    std::lock_guard<std::mutex> lk(owner->scratch_mu);
    XgmShardCtx* c = owner->shard_ctx;
    bool same = c && c->shards.size() == n_shards && std::equal(c->shards.begin(), c->shards.end(), shards);
    for (uint32_t s = 0; same && s < n_shards; ++s) same = c->devs[c->dev_of[s]].device == shards[s]->device;
    if (c && !same) {
        /* another shard list led by the same index: the old context may still be in use by a concurrent xgm_search_sharded (its
         * mutex is taken only after this function returns), so it is retired, not destroyed — freed when the index closes.  A server's
         * shard list is stable: this does not grow. */
        owner->retired_shard_ctx.push_back(c);
        c = owner->shard_ctx = nullptr;
    }
    if (!c) {
        c = new XgmShardCtx();
        c->shards.assign(shards, shards + n_shards);
        for (uint32_t s = 0; s < n_shards; ++s) {
            int di = -1;
            for (size_t j = 0; j < c->devs.size(); ++j) if (c->devs[j].device == shards[s]->device) di = (int)j;
            if (di < 0) {
                ShardDev d;
                d.device = shards[s]->device;
                int rc = use_device(d.device);
                if (rc) { xgm_shard_ctx_destroy(c); return rc; }
                if (hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking) != hipSuccess ||
                    hipEventCreateWithFlags(&d.ev, hipEventDisableTiming) != hipSuccess) {
                    xgm_shard_ctx_destroy(c);
                    return xgm_set_error(XGM_E_DEVICE, "cannot create the shard stream on device %d", d.device);
                }
                c->devs.push_back(d);
                di = (int)c->devs.size() - 1;
            }
            c->dev_of.push_back(di);
        }
        owner->shard_ctx = c;
    }
    *out = c;
    return XGM_OK;
}

/* One shard per device and more than one device (or XGM_SHARDED_RCCL=force on a single device: a 1-rank communicator,
 * which is how the single-GPU test box executes this path): the exchange is an RCCL all-gather. */
static void shard_ctx_try_rccl(XgmShardCtx* c) {
    if (c->rccl_tried) return;
    c->rccl_tried = true;
    const char* mode = getenv("XGM_SHARDED_RCCL");
    const bool force = mode && !strcmp(mode, "force");
    if (c->devs.size() != c->shards.size() || (c->devs.size() < 2 && !force)) return;
    RcclApi& api = rccl_api();
    if (!api.ok) return;
    std::vector<int> devlist;
    for (ShardDev& d : c->devs) devlist.push_back(d.device);
    std::vector<xgm_nccl_comm> comms(devlist.size(), nullptr);
    if (api.CommInitAll(comms.data(), (int)devlist.size(), devlist.data()) != 0) return;
    for (size_t i = 0; i < comms.size(); ++i) c->devs[i].comm = comms[i];
    c->use_rccl = true;
}

extern "C" int xgm_search_sharded(xgm_index* const* shards, uint32_t n_shards, const xgm_query_desc* descs, uint32_t nq,
                                  uint32_t k_stride, xgm_hit* hits, xgm_result_hdr* hdrs) {
    if (!shards || !descs || !hits || !hdrs || n_shards == 0) return xgm_set_error(XGM_E_INVALID, "null argument");
    for (uint32_t s = 0; s < n_shards; ++s) {
        if (!shards[s]) return xgm_set_error(XGM_E_INVALID, "null shard");
        if (shards[s]->device == XGM_DEVICE_NONE) return xgm_set_error(XGM_E_NO_DEVICE, "shard %u was opened without a device", s);
    }
    if (nq == 0) return XGM_OK;
    int rc;
    /* merged statistics and per-shard plans (first = 0: a shard returns its best first+maxitems) */
    static thread_local std::vector<std::vector<xgm_query>> plans;
    plans.resize(n_shards);
    for (auto& v : plans) v.resize(nq);
    std::vector<uint32_t> kq(nq);
    uint64_t docs_total = 0;
    for (uint32_t s = 0; s < n_shards; ++s) docs_total += shards[s]->hdr.doccount;
    for (uint32_t i = 0; i < nq; ++i) {
        xgm_query_desc d = descs[i];
        if (d.n_terms == 0 || d.n_terms > XGM_MAX_TERMS) return XGM_UNSUPPORTED;
        if ((uint64_t)d.first + d.maxitems > k_stride) return xgm_set_error(XGM_E_INVALID, "query %u: first+maxitems > k_stride", i);
        xgm_global_stats gs;
        memset(&gs, 0, sizeof gs);
        for (uint32_t s = 0; s < n_shards; ++s) {
            gs.total_length += shards[s]->hdr.total_length;
            gs.collection_size += shards[s]->hdr.doccount;
            if (shards[s]->hdr.has_positions) gs.full_db_has_positions = 1;
            for (uint32_t t = 0; t < d.n_terms; ++t) {
                uint32_t tf = 0;
                if (!d.terms[t]) return XGM_UNSUPPORTED;
                if ((rc = xgm_lookup_term(shards[s], d.terms[t], d.term_len[t], nullptr, &tf, nullptr, nullptr))) return rc;
                gs.termfreq[t] += tf;
            }
        }
        d.maxitems = d.first + d.maxitems;
        d.first = 0;
        for (uint32_t s = 0; s < n_shards; ++s)
            if ((rc = xgm_plan_query(shards[s], &d, &gs, &plans[s][i]))) return rc;
        /* Enquire::merge_mset clamps against the SUMMED doccount (enquire.cc:486-488); each shard's own plan is
         * clamped to that shard's doccount, and the merge keeps min(available, k) */
        kq[i] = (uint32_t)std::min<uint64_t>(d.maxitems, docs_total);
    }

    XgmShardCtx* c;
    if ((rc = shard_ctx_get(shards, n_shards, &c))) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    const size_t n_hit = (size_t)nq * k_stride;
    hipError_t e = hipSuccess;
    /* grow the persistent buffers (first call, or a larger batch than ever before) */
    if (n_hit > c->cap_hit || nq > c->cap_nq) {
        for (ShardDev& d : c->devs) { if (hipSetDevice(d.device) == hipSuccess) hipDeviceSynchronize(); }
        shard_ctx_free_buffers(c);
        const size_t ch = std::max(n_hit, c->cap_hit), cn = std::max<size_t>(nq, c->cap_nq);
        for (ShardDev& d : c->devs) {
            if ((rc = use_device(d.device))) return rc;
            if ((e = hipMalloc((void**)&d.all_rec, (size_t)n_shards * (ch * sizeof(xgm_hit) + cn * sizeof(xgm_result_hdr)))) != hipSuccess) break;
        }
        if (e == hipSuccess && !(rc = use_device(c->devs[0].device))) {
            if ((e = hipMalloc((void**)&c->d_out_hits, ch * sizeof(xgm_hit))) == hipSuccess)
                e = hipMalloc((void**)&c->d_out_hdrs, cn * sizeof(xgm_result_hdr));
        }
        if (rc) return rc;
        if (e != hipSuccess) { shard_ctx_free_buffers(c); return xgm_launch_error("hipMalloc(shard buffers)", (int)e, hipGetErrorString(e)); }
        c->cap_hit = ch; c->cap_nq = cn;
    }
    const size_t down = n_hit * sizeof(xgm_hit) + (size_t)nq * sizeof(xgm_result_hdr);
    if ((rc = grow_pinned(&c->h_out, &c->cap_hout, down))) return rc;
    shard_ctx_try_rccl(c);
    ++c->calls;

    /* the gathered records of THIS call are dense at the head of the buffers: shard s's at s * rec — its hits, then its headers */
    const size_t rec = xgm_shard_record_bytes(nq, k_stride), o_hdr = n_hit * sizeof(xgm_hit);
    /* 1. every shard's search, enqueued on its device's stream — nothing waits yet */
    for (uint32_t s = 0; s < n_shards; ++s) {
        ShardDev& d = c->devs[c->dev_of[s]];
        if ((rc = search_batch_device_on(shards[s], d.stream, plans[s].data(), nq, k_stride, d.all_rec + (size_t)s * rec, d.all_rec + (size_t)s * rec + o_hdr)))
            break;
    }
    /* 2. the exchange */
    ShardDev& d0 = c->devs[0];
    if (rc == XGM_OK && c->use_rccl) {
        RcclApi& api = rccl_api();
        int nrc = api.GroupStart();
        for (size_t r = 0; r < c->devs.size() && nrc == 0; ++r) {
            ShardDev& d = c->devs[r];
            nrc = api.AllGather(d.all_rec + r * rec, d.all_rec, rec, /*ncclUint8*/ 1, d.comm, d.stream);      /* (one shard per device: rank r holds shard r) */
        }
        const int nrc2 = api.GroupEnd();
        if (nrc || nrc2) rc = xgm_set_error(XGM_E_DEVICE, "ncclAllGather failed (%d)", nrc ? nrc : nrc2);
        ++c->rccl_calls;
    } else if (rc == XGM_OK) {
        for (uint32_t s = 0; s < n_shards && e == hipSuccess; ++s) {
            ShardDev& d = c->devs[c->dev_of[s]];
            if (c->dev_of[s] == 0) continue;
            e = hipMemcpyPeerAsync(d0.all_rec + (size_t)s * rec, d0.device, d.all_rec + (size_t)s * rec, d.device, rec, d.stream);
        }
        for (size_t r = 1; r < c->devs.size() && e == hipSuccess; ++r) {
            if (hipSetDevice(c->devs[r].device) != hipSuccess) { e = hipErrorInvalidDevice; break; }
            if ((e = hipEventRecord(c->devs[r].ev, c->devs[r].stream)) != hipSuccess) break;
        }
        if (e == hipSuccess && !(rc = use_device(d0.device)))
            for (size_t r = 1; r < c->devs.size() && e == hipSuccess; ++r) e = hipStreamWaitEvent(d0.stream, c->devs[r].ev, 0);
    }
    /* 3. merge on shards[0]'s device, one download, ONE wait */
    if (rc == XGM_OK && e == hipSuccess && !(rc = use_device(d0.device))) {
        rc = merge_shards_device_on(shards[0], d0.stream, d0.all_rec, d0.all_rec + o_hdr, n_shards, nq, k_stride, kq.data(), c->d_out_hits, c->d_out_hdrs, rec);
        xgm_hit* h_hits = (xgm_hit*)c->h_out;
        xgm_result_hdr* h_hdrs = (xgm_result_hdr*)(h_hits + n_hit);
        if (rc == XGM_OK) e = hipMemcpyAsync(h_hits, c->d_out_hits, n_hit * sizeof(xgm_hit), hipMemcpyDeviceToHost, d0.stream);
        if (rc == XGM_OK && e == hipSuccess) e = hipMemcpyAsync(h_hdrs, c->d_out_hdrs, (size_t)nq * sizeof(xgm_result_hdr), hipMemcpyDeviceToHost, d0.stream);
        if (rc == XGM_OK && e == hipSuccess) e = hipStreamSynchronize(d0.stream);
        if (rc == XGM_OK && e == hipSuccess) {
            memcpy(hdrs, h_hdrs, (size_t)nq * sizeof(xgm_result_hdr));
            for (uint32_t i = 0; i < nq; ++i)        /* only the valid prefix of each row is defined on the device */
                memcpy(hits + (size_t)i * k_stride, h_hits + (size_t)i * k_stride, (size_t)hdrs[i].n_hits * sizeof(xgm_hit));
        }
    }
    if (rc != XGM_OK || e != hipSuccess) {
        /* leave no work in flight behind a failed call */
        for (ShardDev& d : c->devs) { if (hipSetDevice(d.device) == hipSuccess) hipStreamSynchronize(d.stream); }
        if (rc == XGM_OK) rc = xgm_launch_error("xgm_search_sharded", (int)e, hipGetErrorString(e));
    }
    return rc;
}

/* Diagnostics: how the last xgm_search_sharded on this shard list exchanged the per-shard lists.
 * out[0] calls, out[1] calls that used the RCCL all-gather, out[2] devices, out[3] shards. */
extern "C" int xgm_debug_sharded_info(const xgm_index* owner, uint64_t* out4) {
    if (!owner || !out4 || !owner->shard_ctx) return xgm_set_error(XGM_E_INVALID, "no sharded search has run from this index");
    const XgmShardCtx* c = owner->shard_ctx;
    out4[0] = c->calls; out4[1] = c->rccl_calls; out4[2] = c->devs.size(); out4[3] = c->shards.size();
    return XGM_OK;
}

/* ------------------------------------------------------------------ diagnostics -------------- */

/* Decode one term's whole posting list on the DEVICE (K1 alone) into host arrays. */
extern "C" int64_t xgm_debug_decode_term_device(xgm_index* idx, uint32_t term_id, uint32_t* did, uint32_t* wdf, uint64_t cap) {
    if (!idx || !did || !wdf) return xgm_set_error(XGM_E_INVALID, "null argument");
    if (term_id >= idx->hdr.n_terms) return xgm_set_error(XGM_E_INVALID, "term id out of range");
    int rc = use_device(idx->device);
    if (rc) return rc;
    const uint32_t df = idx->term_df[term_id];
    if (cap < df) return xgm_set_error(XGM_E_INVALID, "buffer too small");
    const uint32_t b0 = (uint32_t)idx->term_blk[term_id], b1 = (uint32_t)idx->term_blk[term_id + 1];
    const uint32_t nblk = b1 - b0;
    std::vector<uint32_t> meta(nblk);
    HIP_TRY(hipMemcpy(meta.data(), (const uint32_t*)idx->d_sections[XGM_S_BLK_META] + b0, (size_t)nblk * 4, hipMemcpyDeviceToHost));
    std::vector<uint64_t> ord(nblk);
    uint64_t o = 0;
    for (uint32_t i = 0; i < nblk; ++i) { ord[i] = o; o += XGM_META_COUNT(meta[i]); }
    if (o != df) return xgm_set_error(XGM_E_INVALID, "block counts (%llu) != df (%u)", (unsigned long long)o, df);
    uint64_t* d_ord = nullptr; uint32_t *d_did = nullptr, *d_wdf = nullptr;
    HIP_TRY(hipMalloc((void**)&d_ord, (size_t)nblk * 8));
    HIP_TRY(hipMalloc((void**)&d_did, (size_t)df * 4));
    HIP_TRY(hipMalloc((void**)&d_wdf, (size_t)df * 4));
    HIP_TRY(hipMemcpy(d_ord, ord.data(), (size_t)nblk * 8, hipMemcpyHostToDevice));
    rc = xgm_launch_decode(idx->view, term_id, b0, nblk, d_ord, d_did, d_wdf, nullptr);
    if (rc == XGM_OK) {
        hipError_t e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(did, d_did, (size_t)df * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(wdf, d_wdf, (size_t)df * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = xgm_launch_error("decode", (int)e, hipGetErrorString(e));
    }
    hipFree(d_ord); hipFree(d_did); hipFree(d_wdf);
    return rc ? rc : (int64_t)df;
}

/* Copy the dense doclen array (entry d = length of docid d, entry 0 unused) to the host. */
extern "C" int64_t xgm_debug_read_doclen(xgm_index* idx, uint32_t* out, uint64_t cap) {
    if (!idx || !out) return xgm_set_error(XGM_E_INVALID, "null argument");
    uint64_t n = (uint64_t)idx->hdr.lastdocid + 1;
    if (cap < n) return xgm_set_error(XGM_E_INVALID, "buffer too small");
    int rc = use_device(idx->device);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, idx->d_sections[XGM_S_DOCLEN], n * 4, hipMemcpyDeviceToHost));
    return (int64_t)n;
}

/* Copy one term's positions (flat, in posting order: Σ wdf entries, widened to u32) to the host; returns their number. */
extern "C" int64_t xgm_debug_read_positions(xgm_index* idx, uint32_t term_id, uint32_t* out, uint64_t cap) {
    if (!idx || (!out && cap)) return xgm_set_error(XGM_E_INVALID, "null argument");
    if (term_id >= idx->hdr.n_terms) return xgm_set_error(XGM_E_INVALID, "term id out of range");
    int rc = use_device(idx->device);
    if (rc) return rc;
    if (!idx->hdr.has_positions || !(idx->term_flags[term_id] & XGM_TF_POS_OK)) return 0;
    uint64_t tp;
    HIP_TRY(hipMemcpy(&tp, (const uint64_t*)idx->d_sections[XGM_S_TERM_POS] + term_id, sizeof tp, hipMemcpyDeviceToHost));
    const uint64_t n = idx->term_cf[term_id];                       /* POS_OK: every posting has exactly wdf positions */
    if (cap < n) return xgm_set_error(XGM_E_INVALID, "buffer too small");
    if (idx->term_flags[term_id] & XGM_TF_POS16) {
        std::vector<uint16_t> tmp(n);
        if (n) HIP_TRY(hipMemcpy(tmp.data(), (const char*)idx->d_sections[XGM_S_POSITIONS] + tp, n * 2, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n; ++i) out[i] = tmp[i];
    } else if (n) {
        HIP_TRY(hipMemcpy(out, (const char*)idx->d_sections[XGM_S_POSITIONS] + tp, n * 4, hipMemcpyDeviceToHost));
    }
    return (int64_t)n;
}

/* Traffic model of the LAST batch launched on this index (wave kernels, tallying instantiation: xgm_index_set_profiling
 * bit 1): the per-unit tallies of what the match kernel requested from memory, summed over the batch's work units.
 * Waits for the device.  Layout of out[0..n): include/xgm.h. */
extern "C" int xgm_last_batch_traffic(xgm_index* idx, uint64_t* out, uint32_t n) {
    if (!idx || !out) return xgm_set_error(XGM_E_INVALID, "null argument");
    uint64_t t[XGM_TRAFFIC_FIELDS] = {};
    if (!idx->last_ghdr || idx->last_n_work == 0) return xgm_set_error(XGM_E_INVALID, "no batch has run on this index");
    int rc = use_device(idx->device);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    std::vector<xgm_group_hdr> h(idx->last_n_work);
    HIP_TRY(hipMemcpy(h.data(), idx->last_ghdr, h.size() * sizeof(xgm_group_hdr), hipMemcpyDeviceToHost));
    /* (a LIST launch: c_pos carries the unit's first weight, the positions tested ride in c_pad[1]; its entries are 24 bytes, 1.5 candidates) */
    const bool list = idx->last_kernel && !strcmp(idx->last_kernel, "xgm_andw_list_kernel");
    for (const xgm_group_hdr& g : h) {
        t[0] += g.c_bmp_words; t[1] += g.c_probes; t[2] += g.c_blk_words; t[3] += g.c_hdrs;
        t[4] += g.c_doclen; t[5] += g.c_aux_words; t[6] += list ? (g.n_cand * 3u + 1u) / 2u : g.n_cand; t[7] += list ? g.c_pad[1] : g.c_pos;
        t[8] += g.c_probes_raw; t[9] += g.c_doclen_raw;
    }
    for (uint32_t i = 0; i < n && i < XGM_TRAFFIC_FIELDS; ++i) out[i] = t[i];
    return XGM_OK;
}

/* Diagnostics: mean host time of xgm_plan_query per query, in microseconds, over `reps` passes of the list. */
extern "C" double xgm_debug_plan_us(const xgm_index* idx, const xgm_query_desc* descs, const xgm_global_stats* gs, uint32_t nq, uint32_t reps) {
    if (!idx || !descs || nq == 0 || reps == 0) return -1.0;
    xgm_query q;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t r = 0; r < reps; ++r)
        for (uint32_t i = 0; i < nq; ++i)
            if (xgm_plan_query(idx, &descs[i], gs ? &gs[i] : nullptr, &q) < 0) return -1.0;
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    return us / ((double)nq * reps);
}

int xgm_phase_cycles_fetch(unsigned long long* out8);
/* Diagnostics: per-phase s_memtime cycle sums of xgm_and_kernel (thread 0 of every workgroup) since
 * the last call; needs XGM_PHASE_TIMING=1 in the environment.  out[0..6] phases, out[7] stripes. */
extern "C" int xgm_debug_phase_cycles(unsigned long long* out8) { return xgm_phase_cycles_fetch(out8); }
int xgm_orw_cycles_fetch(unsigned long long* out8);
int xgm_merge_cycles_fetch(unsigned long long* out8);
extern "C" int xgm_debug_merge_cycles(unsigned long long* out8) { return xgm_merge_cycles_fetch(out8); }
extern "C" int xgm_debug_orw_phase_cycles(unsigned long long* out8) { return xgm_orw_cycles_fetch(out8); }

/* Diagnostics (host only — works on an XGM_DEVICE_NONE index): the decomposition a batch would be launched with.
 * kernel[32] receives the match kernel's name (and ":sided1" / ":sided2" / ":phrase" for the conjunction kernel's
 * instantiation); units receives (qi, s_begin, s_end, slot) per work unit in launch order; returns the number of
 * units (possibly > cap: only cap are written), or < 0 / XGM_UNSUPPORTED like a search would. */
extern "C" int64_t xgm_debug_plan_batch(const xgm_index* idx, const xgm_query* qs, uint32_t nq, char* kernel, uint32_t* units, uint64_t cap) {
    if (!idx || !qs || !kernel || (!units && cap)) return xgm_set_error(XGM_E_INVALID, "null argument");
    if (nq == 0) return 0;
    std::vector<xgm_dev_query> dq(nq);
    std::vector<uint32_t> kq(nq);
    std::vector<double> mp(nq);
    BatchPlan bp;
    int rc = plan_batch(idx, qs, nq, dq.data(), kq.data(), mp.data(), &bp);
    if (rc) return rc;
    snprintf(kernel, 32, "%s%s", bp.andw ? "xgm_andw_kernel" : bp.orw2 ? "xgm_orw2_kernel" : bp.orw ? "xgm_orw_kernel" : bp.and_only ? "xgm_and_kernel" : "xgm_match_kernel",
             bp.andw && bp.phrase ? ":phrase" : bp.andw && bp.sided == 2 ? ":sided2" : bp.andw && bp.sided == 1 ? ":sided1" : "");
    for (uint64_t i = 0; i < bp.work.size() && i < cap; ++i) {
        units[4 * i] = bp.work[i].qi; units[4 * i + 1] = bp.work[i].s_begin; units[4 * i + 2] = bp.work[i].s_end; units[4 * i + 3] = bp.work[i].slot;
    }
    return (int64_t)bp.work.size();
}

/* Diagnostics (host only): the disjunction kernel's pruning inputs of one planned query — the guess of the final k-th weight and the
 * per-term weight bounds (largest wdf / wdf = 1), in plan order. */
extern "C" int xgm_debug_or_bounds(const xgm_index* idx, const xgm_query* q, double* seed, double* ub, double* ub1) {
    if (!idx || !q) return xgm_set_error(XGM_E_INVALID, "null argument");
    xgm_dev_query d;
    const int w = to_dev_query(idx, q, &d);
    if (w <= 0) return XGM_UNSUPPORTED;
    if (seed) *seed = d.theta_seed;
    for (uint32_t t = 0; t < q->n_terms; ++t) { if (ub) ub[t] = d.ub[t]; if (ub1) ub1[t] = d.ub1[t]; }
    return XGM_OK;
}

/* Diagnostics (host only): the launches a batch is cut into — one per kernel class present (run_batch) — as
 * "<kernel>[:variant]*<queries>" joined by ';' in launch order; returns the number of launches, or < 0 /
 * XGM_UNSUPPORTED like a search would. */
extern "C" int xgm_debug_batch_launches(const xgm_index* idx, const xgm_query* qs, uint32_t nq, char* out, uint32_t cap) {
    if (!idx || !qs || !out || cap == 0) return xgm_set_error(XGM_E_INVALID, "null argument");
    out[0] = 0;
    std::vector<std::vector<xgm_query>> by(XGM_CLS_COUNT);
    for (uint32_t i = 0; i < nq; ++i) by[classify_query(idx, qs[i])].push_back(qs[i]);
    int n = 0;
    std::string acc;
    for (auto& v : by) {
        if (v.empty()) continue;
        char name[32];
        int64_t rc = xgm_debug_plan_batch(idx, v.data(), (uint32_t)v.size(), name, nullptr, 0);
        if (rc < 0 || rc == XGM_UNSUPPORTED) return (int)rc;
        if (n++) acc += ";";
        acc += name;
        acc += "*" + std::to_string(v.size());
    }
    snprintf(out, cap, "%s", acc.c_str());
    return n;
}

/* Diagnostics: per work-unit (qi, s_begin, s_end, slot, t_start, t_end, matches, documents weighed)
 * of the LAST batch launched on this index from any thread; out is u64[8 * cap]; returns the number
 * of units. */
static int64_t debug_last_units(xgm_index* idx, unsigned long long* out, unsigned long long* out_pos, uint64_t cap);
extern "C" int64_t xgm_debug_last_units(xgm_index* idx, unsigned long long* out, uint64_t cap) { return debug_last_units(idx, out, nullptr, cap); }
/* ... plus the units' c_pos field (xgm_orw_kernel's tallying build: bit 63 = the unit went round a second time, low bits = documents of essential block-decoded terms) */
extern "C" int64_t xgm_debug_last_units2(xgm_index* idx, unsigned long long* out, unsigned long long* out_pos, uint64_t cap) { return debug_last_units(idx, out, out_pos, cap); }
static int64_t debug_last_units(xgm_index* idx, unsigned long long* out, unsigned long long* out_pos, uint64_t cap) {
    if (!idx || !out || !g_last_ghdr) return -1;
    hipDeviceSynchronize();
    std::vector<xgm_group_hdr> h(g_last_work.size());
    if (hipMemcpy(h.data(), g_last_ghdr, h.size() * sizeof(xgm_group_hdr), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    uint64_t n = std::min<uint64_t>(cap, g_last_work.size());
    for (uint64_t i = 0; i < n; ++i) {
        const xgm_work& w = g_last_work[i];
        out[8 * i] = w.qi; out[8 * i + 1] = w.s_begin; out[8 * i + 2] = w.s_end; out[8 * i + 3] = (uint64_t)h[w.slot].c_pad[0] | ((uint64_t)h[w.slot].c_pad[1] << 32);   /* phase clocks (units of 64 cycles) of the tallying build */
        out[8 * i + 4] = h[w.slot].t_start; out[8 * i + 5] = h[w.slot].t_end;
        out[8 * i + 6] = h[w.slot].matches; out[8 * i + 7] = h[w.slot].pad;
        if (out_pos) out_pos[i] = h[w.slot].c_pos;
    }
    return (int64_t)n;
}
