/* Launch interface between the host API (xgm_api.cc) and the HIP translation units. */
#ifndef XGM_LAUNCH_H
#define XGM_LAUNCH_H

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "xgm_device.h"

struct xgm_match_launch {
    xgm_seg_dev seg;
    const xgm_dev_query* queries;     /* device, [nq] */
    uint32_t nq, n_work;              /* queries in the batch, work units (= workgroups)          */
    const xgm_work* work;             /* device, [n_work], heaviest first                         */
    uint32_t stripes_per_group;       /* max stripes of a unit (sizes the LDS run table)          */
    uint32_t sub_bits = 0;            /* workgroup kernels: a stripe is taken in 2^sub_bits passes over tables of W >> sub_bits slots
                                         (positional queries of more than XGM_PHRASE_MAX_TERMS_WG terms; the *_smem_bytes functions then
                                         take stripe_bits - sub_bits) */
    uint32_t tab_terms;               /* max n_terms in the batch (LDS table rows)               */
    uint32_t cap;                     /* top-k buffer capacity, power of two >= k_max + XGM_WG    */
    uint32_t k_stride;                /* candidates reserved per (query, group)                   */
    bool phrase, wide;                /* kernel variant: positional tables / 16-bit wdf tables    */
    bool tally = false;               /* wave kernels: also fill the traffic tallies of xgm_group_hdr (measurement)  */
    int sided = 0;                    /* conjunction batch with right-hand terms: 1 = AND_NOT only, 2 = AND_MAYBE too */
    bool or_flat = false;             /* disjunction batch whose every term without a container has a flat posting array: xgm_orw_kernel<…, FLAT> */
    int orw_planes = 6;               /* xgm_orw_kernel: planes of the bound sum, 4 where every query of the batch has 4-8 terms (plan_batch) */
    int orw2 = 0;                     /* disjunction batch for xgm_orw2_kernel: 1 = every term has a container, 2 = up to two terms per query come from flat arrays */
    uint32_t* hist = nullptr;         /* device, [nq][XGM_OR_HIST] zeroed: the query-wide weight histogram of xgm_orw_kernel and of the
                                         positional instantiation of xgm_andw_kernel (units of one query share their k-th weight bound) */
    xgm_cand* cand;                   /* device, [n_work][k_stride]                               */
    xgm_group_hdr* ghdr;              /* device, [n_work]                                         */
    const xgm_fuse* fuse = nullptr;   /* xgm_andw_kernel only: device copy of the parameters with which the kernel writes the final hits itself (no merge launch) */
    /* wave kernels: events carried by the kernel's OWN dispatch packet (hipExtLaunchKernel) — its start / its completion — instead of
     * hipEventRecord()s around it: every recorded event is one more barrier packet the command processor works through between two match
     * kernels of consecutive batches (measured: DESIGN.md 11) */
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    uint32_t spy_stride = 0;          /* xgm_match_sorted_kernel with a spy in a BATCH: counts of query qi at spy_counts + qi * spy_stride (0: one query, one row) */
};

/* launch with the events of L (if any) attached to the dispatch */
#define XGM_LAUNCH_TIMED(L_, kern, grid, block, smem, stream, ...)                                                                          \
    do {                                                                                                                                \
        if ((L_).ev_start || (L_).ev_stop) hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)(smem), stream, (L_).ev_start, (L_).ev_stop, 0u, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern, grid, block, smem, stream, __VA_ARGS__);                                                         \
    } while (0)

size_t xgm_match_smem_bytes(uint32_t stripe_bits, uint32_t tab_terms, bool phrase, uint32_t cap, bool wide, uint32_t stripes_per_group);
int xgm_launch_match(const xgm_match_launch& L, hipStream_t stream);
/* the same kernel under a value sort (mode 1 value, 2 value then relevance, 3 relevance then value; ord = the device column):
 * candidates of 32 bytes with two keys, L.cand unused */
size_t xgm_match_sorted_smem_bytes(uint32_t stripe_bits, uint32_t tab_terms, bool phrase, uint32_t cap, bool wide, uint32_t stripes_per_group);
int xgm_launch_match_sorted(const xgm_match_launch& L, const uint32_t* ord, uint32_t mode, uint32_t reverse, const uint32_t* spy_ord, uint32_t* spy_counts,
                            const uint32_t* cord, uint32_t cmax, xgm_cand_sorted* cand, hipStream_t stream,
                            unsigned long long* all_keys = nullptr, unsigned long long* all_vals = nullptr, unsigned long long* all_count = nullptr,
                            unsigned long long all_cap = 0);
/* all_keys != NULL (xgm_search_all): every matching document is also appended, in no particular order, to all_keys / all_vals
 * (docid << 32 | weighted leaves matched, weight bits) at the position the zeroed counter *all_count hands out; entries beyond all_cap
 * are counted, not written.  xgm_all_order_pack (xgm_all.hip) restores docid order: tmp = xgm_all_order_bytes(lastdocid) device bytes. */
size_t xgm_all_order_bytes(uint32_t lastdocid);
int xgm_all_order_pack(void* tmp, uint32_t lastdocid, const unsigned long long* keys, const unsigned long long* vals, size_t n, xgm_hit* out,
                       hipStream_t stream);
/* xgm_search_replay's second half (xgm_replay.hip): ProtoMSet's collation of a search by relevance replayed over the match list in docid
 * order — one workgroup; out_hits [max_size] gets the page in rank order, *out the figures.  frozen_mode: the list carries the underlying
 * conjunction of a positional query (entries flagged XGM_ALL_NOT_A_MATCH are not matches) and the weight freezes as SelectPostList's does. */
typedef struct {
    uint64_t known_matching_docs;
    double max_weight;                /* ProtoMSet::max_weight: the heaviest weight it was shown, 0 if none */
    double frozen_weight;
    uint32_t max_weight_subqs, n_hits, frozen, reserved;
} xgm_replay_out;
int xgm_launch_replay(const xgm_hit* list, uint64_t n, uint32_t max_size, uint64_t check_at_least, bool frozen_mode, uint64_t total_matches,
                      xgm_hit* out_hits, xgm_replay_out* out, hipStream_t stream);
/* (total_matches: the matching documents among the n entries — frozen mode stops walking once the rest of the loop can only count) */
/* the same in parallel (check_at_least <= K, no frozen weight): S segments of seg_len entries (> K), one wave each — per-segment top K, an
 * exclusive scan of the states, per-segment replay; *known_sum (device, zeroed here) receives known_matching_docs, out->n_hits / out_hits the page */
size_t xgm_replay_parallel_bytes(uint32_t S, uint32_t K);
int xgm_launch_replay_parallel(const xgm_hit* list, uint64_t n, uint32_t K, uint64_t check_at_least, uint32_t S, uint64_t seg_len, void* scratch,
                               xgm_hit* out_hits, xgm_replay_out* out, unsigned long long* known_sum, hipStream_t stream);
/* (mode 4 = relevance alone, ord may be NULL; spy_counts — device, zeroed, one u32 per ordinal of spy_ord — may be NULL; cord = the collapse
 *  column's ordinals or NULL, cmax = collapse_max) */
/* conjunction-only batches (every query: AND of >= 2 terms, no positional filter) */
size_t xgm_and_smem_bytes(uint32_t stripe_bits, uint32_t tab_terms, uint32_t cap, bool wide, uint32_t stripes_per_group);
int xgm_launch_and(const xgm_match_launch& L, hipStream_t stream);
/* wave-autonomous variant: one wave per work unit, no workgroup barriers (first+maxitems <= 192);
 * L.phrase selects the instantiation with the positional filter (every term block-decoded) */
size_t xgm_andw_smem_bytes(uint32_t stripe_bits, uint32_t tab_terms, uint32_t cap, bool wide, uint32_t spg, bool phrase, bool sided);
int xgm_launch_andw(const xgm_match_launch& L, hipStream_t stream);
/* positional batches in the reference-identical mode: the units' first matches in docid order (xgm_prefix_entry lists where the candidates would
 * be: L.k_stride = XGM_PREFIX_CAND_STRIDE(k_max)), then — xgm_frozen.hip — one wave per query walks its units' lists as ProtoMSet and
 * SelectPostList would and writes the page: hits [rows][k_stride_out], hdrs [rows], extra [rows] = known_matching_docs | status bits */
int xgm_launch_andw_list(const xgm_match_launch& L, hipStream_t stream);
int xgm_launch_frozen_finish(const xgm_dev_query* queries, uint32_t nq, const uint32_t* goff, const xgm_cand* cand, const xgm_group_hdr* ghdr, uint32_t k_stride_c,
                             const double* max_possible, const uint32_t* row_of, xgm_hit* hits, xgm_result_hdr* hdrs, unsigned long long* extra, uint32_t k_stride_out,
                             hipStream_t stream);
/* plain conjunctions with XGM_REPLAY_BATCH_COUNT: the units' top-k lists (L.cand / L.ghdr, no merge) + every match in docid order (out: arena, cursor, chunk
 * table); then — xgm_count.hip — per query an exclusive scan of the units' top-k lists (the page = its total; states [n_work + nq][k_stride_c] = the versions of what ProtoMSet keeps, unit_ver [n_work]
 * = the version the walk reaches a unit with, unit_before [n_work] = the documents it has seen by then) and per unit the count of what ProtoMSet::add is shown, added onto extra[row] */
int xgm_launch_andw_all(const xgm_match_launch& L, const xgm_all_out& out, hipStream_t stream);
int xgm_launch_count_finish(const xgm_dev_query* queries, uint32_t nq, const xgm_work* work, uint32_t n_work, const uint32_t* goff, const xgm_cand* cand,
                            const xgm_group_hdr* ghdr, uint32_t k_stride_c, const xgm_all_out& lists, xgm_cand* states, uint32_t* unit_ver, unsigned long long* unit_before,
                            const double* max_possible, const uint32_t* row_of, xgm_hit* hits, xgm_result_hdr* hdrs, unsigned long long* extra, uint32_t k_stride_out,
                            hipStream_t stream);
size_t xgm_body_wave_bytes(bool flat, bool phrase, uint32_t terms);      /* LDS a unit of xgm_flat_unit (flat) / xgm_dense_unit uses of the wave's slice */
/* conjunctions (or positional queries that prune by weight) whose every term has probe containers, <= xgm_dense_max_terms() terms,
 * k <= xgm_dense_max_k(), units of <= xgm_dense_max_stripes() stripes: xgm_dense_and.hip; L.phrase selects the positional instantiation */
size_t xgm_dense_smem_bytes(bool phrase);
uint32_t xgm_dense_max_terms();
uint32_t xgm_dense_max_k();
uint32_t xgm_dense_max_stripes();
bool xgm_dense_word_major();
int xgm_launch_dense(const xgm_match_launch& L, hipStream_t stream);
/* disjunction-only batches: one wave per work unit, MaxScore pruning; hist = [nq][XGM_OR_HIST] zeroed u32 */
size_t xgm_orw_smem_bytes(uint32_t stripe_bits, uint32_t tab_terms, uint32_t cap, bool wide, uint32_t spg);
/* ... xgm_orw2_kernel (round 5: <= 8 terms with containers or flat arrays, one-byte wdf; three / four waves per SIMD) */
size_t xgm_orw2_smem_bytes(uint32_t stripe_bits, uint32_t tab_terms, uint32_t cap, uint32_t spg, bool sparse);
bool xgm_orw2_enabled();
int xgm_launch_orw(const xgm_match_launch& L, uint32_t* hist, hipStream_t stream);
int xgm_launch_merge(const xgm_cand* cand, const xgm_group_hdr* ghdr, const uint32_t* goff, uint32_t k_stride_in,
                     const uint32_t* kq, uint32_t nq, uint32_t cap, uint32_t k_stride_out, xgm_hit* hits,
                     xgm_result_hdr* hdrs, const double* max_possible, const uint32_t* row_of, hipStream_t stream);
/* the parts of a query's units (plan_batch, parts > 1) merged once more: all_hits [n_parts][nq][k_stride], docids as they are; row q of
 * the result goes to row_of[q] (q when NULL) */
int xgm_launch_merge_parts(const xgm_hit* all_hits, const xgm_result_hdr* all_hdrs, uint32_t n_parts, uint32_t nq, uint32_t k_stride,
                           const uint32_t* kq, uint32_t cap, xgm_hit* hits, xgm_result_hdr* hdrs, const uint32_t* row_of, hipStream_t stream);
int xgm_launch_merge_shards(const xgm_hit* all_hits, const xgm_result_hdr* all_hdrs, uint32_t n_shards, uint32_t nq,
                            uint32_t k_stride, const uint32_t* kq, uint32_t cap, xgm_hit* hits, xgm_result_hdr* hdrs,
                            hipStream_t stream, size_t shard_record_bytes = 0);
/* (shard_record_bytes != 0: shard s's hits start at all_hits + s * shard_record_bytes, its headers at all_hdrs + s * shard_record_bytes —
 *  the packed records of ONE all-gather, xgm_merge_shards_packed_device) */
int xgm_launch_decode(const xgm_seg_dev& seg, uint32_t term_id, uint32_t b0, uint32_t nblk, const uint64_t* ord_base,
                      uint32_t* out_did, uint32_t* out_wdf, hipStream_t stream);

/* records "<what> failed: <msg>" as the thread's last error and returns XGM_E_DEVICE */
int xgm_launch_error(const char* what, int code, const char* msg);

#endif
