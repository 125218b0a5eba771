/* Internal declarations shared by the translation units of libxgm.so (host side). */
#ifndef XGM_INTERNAL_H
#define XGM_INTERNAL_H

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/xgm.h"
#include "xgm_segment.h"

int xgm_set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

/* A complete segment as one host blob (header + sections). */
struct XgmSegmentBlob {
    std::vector<uint8_t> bytes;
    const uint8_t* map = nullptr;            /* instead of bytes: a read-only mapping of a segment file (xgm_map_segment_file) */
    size_t map_size = 0;
    XgmSegmentBlob() = default;
    XgmSegmentBlob(const XgmSegmentBlob&) = delete;
    XgmSegmentBlob& operator=(const XgmSegmentBlob&) = delete;
    ~XgmSegmentBlob();
    const uint8_t* data() const { return map ? map : bytes.data(); }
    size_t size() const { return map ? map_size : bytes.size(); }
    const xgm_seg_header* header() const { return reinterpret_cast<const xgm_seg_header*>(data()); }
    template <class T>
    const T* section(int s) const { return reinterpret_cast<const T*>(data() + header()->sec_off[s]); }
};

int xgm_build_segment_blob(const xgm_raw_postings* raw, uint32_t stripe_bits, XgmSegmentBlob* out);

/* The segment writer behind xgm_build_segment_blob and the incremental refresh (xgm_segment_build.cc). */
struct XgmSegmentWriter {
    uint32_t stripe_bits = 0, wdf_ub_db = 0;
    bool has_pos = false;
    std::vector<uint32_t> term_df, term_cf, term_wdfub, term_flags;
    std::vector<uint64_t> term_blk, term_word, term_pos, str_off;
    std::vector<uint32_t> blk_first, blk_meta, blk_word, blk_pos, words;
    std::vector<uint8_t> positions;          /* per term: u16 or u32 entries (XGM_TF_POS16) */
    std::vector<char> str_bytes;
    uint64_t n_pos_entries = 0, n_postings = 0;
    /* the term being written */
    bool t_pos_ok = false, t_pos16 = false, t_have_first = false;
    uint64_t t_entries = 0, t_df = 0, t_cf = 0;
    uint32_t t_first_wdf = 0;

    int begin(uint32_t stripe_bits, bool with_positions, uint32_t wdf_ub_of_the_database);
    void reserve_like(const XgmSegmentBlob& old);      /* the result will be about as large as that segment */
    /* pos_ok / pos16: the positional form of the WHOLE term (every posting has exactly wdf positions / all below 65536) */
    int begin_term(const char* name, uint32_t len, bool pos_ok, bool pos16);
    /* blocks [first block of old term t, b_end) verbatim (headers, payload, positions at this term's entry width): only as the first
     * thing of a term; cf_of_them = Σ wdf of their postings, first_wdf = wdf of the term's first posting */
    int copy_blocks(const XgmSegmentBlob& old, uint32_t t, uint64_t b_end, uint64_t cf_of_them, uint32_t first_wdf);
    /* postings in ascending docid order, after whatever the term holds already; pos_off is indexed like did (df + 1 entries) */
    int add_postings(const uint32_t* did, const uint32_t* wdf, uint32_t df, const uint64_t* pos_off, const uint32_t* pos);
    int end_term();
    /* out_path != NULL: the segment goes straight to that file (out is not touched) instead of being assembled in memory */
    int finish(const xgm_raw_postings* raw, uint32_t doclen_lb, uint32_t doclen_ub, XgmSegmentBlob* out, const char* out_path = nullptr);
    void put_position(uint32_t v);
};
int xgm_write_blob(const XgmSegmentBlob& blob, const char* path);
void xgm_database_bounds(const xgm_raw_postings* raw, uint32_t wdf_max_seen, uint32_t* doclen_lb, uint32_t* doclen_ub, uint32_t* wdf_ub);
void xgm_positional_form(const uint32_t* wdf, uint32_t df, const uint64_t* pos_off, const uint32_t* pos, bool* pos_ok, bool* pos16);
int xgm_read_raw_file(const char* path, std::vector<uint8_t>* storage, std::vector<const char*>* term_ptrs,
                      std::vector<uint32_t>* term_lens, xgm_raw_postings* raw);
int xgm_validate_header(const xgm_seg_header* h, uint64_t avail_bytes);
int xgm_validate_blob(const XgmSegmentBlob& blob);

/* Per-thread scratch for searches (device + pinned host buffers), see xgm_api.cc. */
struct XgmScratch;
struct XgmBatcher;       /* opt-in micro-batching queue + dispatcher thread (xgm_index_set_batching) */
struct XgmShardCtx;      /* persistent buffers / streams / RCCL communicator of one shard list (xgm_search_sharded) */

struct xgm_index {
    int device = -1;
    xgm_seg_header hdr{};
    /* host copies used for planning / lookup */
    std::vector<uint32_t> term_df, term_cf, term_wdfub, term_flags;
    std::vector<uint32_t> term_wdfmax;   /* the terms' true largest wdf where known (terms with probe containers), else term_wdfub; empty = use term_wdfub.
                                            Pruning bounds only: MSet::max_possible keeps glass's looser bound like the reference */
    std::vector<uint64_t> term_blk, term_word;
    std::vector<uint64_t> str_off;
    std::vector<char> str_bytes;
    std::vector<uint32_t> term_hash;   /* open-addressing table over the term strings → term id (xgm_lookup_term_id) */
    std::once_flag term_hash_once;
    /* device */
    void* d_blob = nullptr;            /* whole segment (file-loaded) or nullptr when sections are separate */
    void* d_sections[XGM_S_COUNT] = {};/* device pointer of each device-resident section             */
    bool sections_owned = false;       /* synthetic builder allocates sections one by one            */
    xgm_seg_dev view{};
    uint64_t device_bytes = 0;
    void* d_dense_id = nullptr;        /* probe containers (xgm_dense.hip) */
    void* d_dense_dir = nullptr;
    void* d_dense_data = nullptr;
    void* d_doclen_narrow = nullptr;   /* xgm_seg_dev::doclen_narrow */
    void* d_flat_off = nullptr;        /* flat posting arrays of the terms without containers (xgm_seg_dev::flat_*) */
    void* d_flat_did = nullptr;
    void* d_flat_wdf = nullptr;
    void* d_flat_pos = nullptr;
    uint64_t flat_bytes = 0, flat_postings = 0;
    uint64_t dense_bytes = 0;
    uint64_t dense_min_df = UINT64_MAX;   /* termfreq from which a term has probe containers */
    void* stream = nullptr;            /* hipStream_t                                                */
    bool own_stream = false;
    bool profiling = false;
    std::atomic<bool> near_colocated{false};   /* distinct terms may share a position in this shard: NEAR by the reference's full procedure (xgm_index_set_near_colocated) */
    bool tally = false;                /* launch the wave kernels' tallying instantiation (xgm_index_set_profiling bit 1) */
    const char* last_kernel = "";      /* diagnostics: which match kernel the last batch used */
    void* last_ghdr = nullptr;         /* device: per-unit summaries of the last batch (xgm_last_batch_traffic) */
    uint32_t last_n_work = 0;
    std::vector<std::pair<void*, void*>> prof_events;   /* hipEvent_t pairs around the match kernel */
    size_t prof_used = 0;
    std::mutex scratch_mu;
    std::vector<XgmScratch*> scratch_pool;
    uint32_t scratch_total = 0;        /* scratches created so far (pooled + in use) */
    XgmBatcher* batcher = nullptr;
    XgmShardCtx* shard_ctx = nullptr;  /* when this index is shards[0] of an xgm_search_sharded list */
    std::vector<XgmShardCtx*> retired_shard_ctx;   /* contexts of earlier shard lists led by this index: destroyed when it closes */
    std::map<uint32_t, std::pair<void*, uint32_t>> columns;   /* value slot → (device u32 ord[lastdocid + 1], distinct values): xgm_index_attach_column */
    std::vector<uint32_t> term_order;                          /* term ids in byte order of the terms (xgm_expand_prefix), made on first use */
    std::once_flag term_order_once;
    std::vector<void*> retired_columns;                        /* replaced columns' device arrays: freed when the index closes */
    std::mutex columns_mu;
};

int xgm_lookup_term_id(const xgm_index* idx, const char* term, size_t len, uint32_t* id);

/* Build the probe containers of the dense terms from the block-encoded postings already in HBM and
 * attach them to idx->view.  No-op (n_dense = 0) when no term qualifies. */
int xgm_build_dense(xgm_index* idx);

#endif
