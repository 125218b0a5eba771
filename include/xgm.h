/* xgm.h — C ABI of the MI355X match/rank engine ("xgm" = Xapian GPU Matcher).
 *
 * This is the drop-in boundary for ONE hot path of Kronuz/Xapiand: the body of
 * Xapian::Enquire::get_mset as Xapiand calls it once per shard
 *   reference: src/database/handler.cc:1338      DocMatcher::get_mset → enquire.get_mset(...)
 *              src/xapian/api/enquire.cc:396-470  Enquire::Internal::get_mset
 *              src/xapian/matcher/matcher.cc:346-542  Matcher::get_local_mset (the hot loop :482-536)
 * The Xapian::Enquire / Query / MSet surface and Xapiand's query_dsl lowering stay as they are; a
 * hook in Matcher::get_local_mset lowers an eligible Xapian::Query to an
 * `xgm_query_desc`, calls the functions below, and builds the Xapian::MSet from `xgm_hit`s.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch types; no exceptions cross the boundary.
 *   - return 0 = ok; > 0 = "query shape not supported, use the CPU path" (never a user error);
 *     < 0 = hard failure (the hook throws Xapian::DatabaseError so Xapiand's retry logic,
 *     src/database/handler.cc:1348-1368, applies).  xgm_last_error() gives the message
 *     (thread-local).
 *   - The caller owns every buffer it passes; the library owns xgm_index and all device memory.
 *   - There is NO CPU fallback inside the library: search entry points fail with XGM_E_NO_DEVICE
 *     when no HIP device is usable.
 *   - Thread safety: any number of threads may call xgm_search* / xgm_plan_query / xgm_lookup_term
 *     concurrently on one xgm_index (per-call scratch is pooled per thread).  open/close are
 *     serialised by the caller (Xapiand's lock_shard / DatabasePool).
 */
#ifndef XGM_H
#define XGM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: exactly what this header declares is exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define XGM_MAX_TERMS 16      /* leaves per query handled on the device path            */
#define XGM_MAX_K 1024        /* first + maxitems handled on the device path            */

/* return codes */
#define XGM_OK 0
#define XGM_UNSUPPORTED 1           /* fall back to the CPU matcher                       */
#define XGM_E_INVALID (-1)          /* bad argument / corrupt segment                     */
#define XGM_E_IO (-2)
#define XGM_E_NO_DEVICE (-3)        /* HIP runtime/device unavailable                     */
#define XGM_E_DEVICE (-4)           /* a HIP call failed                                  */
#define XGM_E_REVISION (-5)         /* segment revision != requested (DatabaseModifiedError) */
#define XGM_E_NOMEM (-6)

/* query operators: the subset of Xapian::Query::op in BASELINE.json's configs
 * (reference: src/xapian/query.h:48-  OP_AND = 0, OP_OR = 1, OP_PHRASE = 7). */
#define XGM_OP_AND 1
#define XGM_OP_OR 2
#define XGM_OP_PHRASE 3
/* Two-sided operators (reference src/xapian/api/queryinternal.cc:2208-2283; Xapiand DSL _and_not /
 * _and_maybe / _filter, src/query_dsl.cc:285-296): left = the AND of the first n_required terms,
 * right = the other terms — any of them excludes the document (AND_NOT: AndNotPostList over an
 * unweighted OR), each adds its weight where it matches (AND_MAYBE: AndMaybePostList over a weighted
 * OR), all must match but carry no weight (FILTER: MultiAndPostList with an unweighted side). */
#define XGM_OP_AND_NOT 4
#define XGM_OP_AND_MAYBE 5
#define XGM_OP_FILTER 6
/* OP_NEAR (reference src/xapian/matcher/nearpostlist.cc:60-160): the AND of the terms, filtered to documents where one
 * occurrence of every term falls inside a span shorter than `window`, in any order.  Distinct terms only. */
#define XGM_OP_NEAR 7

/* Nested queries (reference src/xapian/api/queryinternal.cc: the trees Xapiand's DSL builds, src/query_dsl.cc:188-432):
 * a post-order program over desc->terms[].  XGM_T_TERM pushes terms[term]; the operators pop `arity` values:
 * AND / OR of >= 1, SYNONYM of >= 1 TERMS (OP_SYNONYM: one BM25 weight over the summed wdf, queryinternal.cc:1822-1898),
 * AND_NOT / AND_MAYBE of >= 2 (first = left side, the others the right side), FILTER of 2, SCALE of 1
 * (OP_SCALE_WEIGHT by tree_scale[i]).  Terms must be distinct; no positional operators inside a tree. */
#define XGM_OP_TREE 8
#define XGM_MAX_TREE 40
#define XGM_T_TERM 0
#define XGM_T_AND 1
#define XGM_T_OR 2
#define XGM_T_AND_NOT 3
#define XGM_T_AND_MAYBE 4
#define XGM_T_FILTER 5
#define XGM_T_SYNONYM 6
#define XGM_T_SCALE 7
/* OP_WILDCARD expanded by the caller (xgm_expand_prefix) over terms: its OP_SYNONYM form is a synonym group that is never
 * simplified to its only term (QueryWildcard::postlist wraps even one expansion in a SynonymPostList, api/queryinternal.cc:1441-
 * 1480; an explicit OP_SYNONYM of one term IS that term, QuerySynonym::done), its OP_OR form an OR tree of its own that an
 * enclosing OR does not flatten (QueryWildcard has no postlist_sub_or_like: queryinternal.cc:935-941) */
#define XGM_T_WILDCARD 8
#define XGM_T_WILDCARD_OR 9
typedef struct { uint8_t kind, arity; uint16_t term; } xgm_tree_op;

typedef struct xgm_index xgm_index; /* opaque: device-resident segment of ONE shard revision */
#define XGM_DEVICE_NONE (-1)

/* ---- raw postings: what the exporter hands to the segment builder -----------------------------
 * Produced by walking Xapian's public iterators over a glass shard (Database::allterms_begin /
 * postlist_begin / PostingIterator::get_wdf / positionlist_begin / get_doclength — SURVEY.md
 * Appendix A; reference src/xapian/api/database.cc:198,347).  Terms sorted bytewise, postings of a
 * term sorted by docid, positions of a posting ascending.  File form ("XGMRAW1"), little endian,
 * every array padded to a multiple of 8 bytes:
 *   char magic[8]; u32 n_terms, lastdocid, doccount, has_positions; u64 total_length, n_postings,
 *   n_positions, revision, term_bytes_total;           (64-byte header)
 *   u32 doclen[lastdocid+1]; u32 df[n_terms]; u32 did[n_postings]; u32 wdf[n_postings];
 *   if has_positions: u64 pos_off[n_postings+1]; u32 pos[n_positions];
 *   u32 term_len[n_terms]; char term_bytes[term_bytes_total] (concatenated). */
typedef struct {
    uint32_t n_terms;
    uint32_t lastdocid;          /* highest docid in the shard (docids are 1-based)               */
    uint32_t doccount;           /* number of documents (== lastdocid when there are no gaps)      */
    uint32_t has_positions;
    uint64_t total_length;       /* sum of document lengths (Weight::Internal::total_length)       */
    uint64_t n_postings;
    uint64_t n_positions;
    uint64_t revision;           /* Database::get_revision() of the shard this was exported from   */
    const uint32_t* doclen;      /* [lastdocid+1], entry 0 unused, 0 for absent docids             */
    const char* const* terms;    /* [n_terms] term bytes (may contain \0)                          */
    const uint32_t* term_len;    /* [n_terms]                                                      */
    const uint32_t* df;          /* [n_terms] termfreq; postings of term t are the next df[t]      */
    const uint32_t* did;         /* [n_postings]                                                   */
    const uint32_t* wdf;         /* [n_postings]                                                   */
    const uint64_t* pos_off;     /* [n_postings+1] or NULL                                         */
    const uint32_t* pos;         /* [n_positions] or NULL                                          */
    /* Database-wide statistics bounds as the backend reports them (Database::get_doclength_lower_bound /
     * get_wdf_upper_bound; glass keeps them in its version file and never tightens them when documents are deleted
     * or replaced).  0 = derive the tight bound from the postings.  They only enter BM25Weight::get_maxpart, i.e.
     * MSet::get_max_possible (and the pruning bounds never exceed them). */
    uint32_t doclen_lower_bound;
    uint32_t wdf_upper_bound;
    uint32_t doclen_upper_bound;
    uint32_t reserved;
} xgm_raw_postings;

/* Build the block-compressed device segment from raw postings on the host and write
 * it to `out_path`.  stripe_bits: log2 of the docid stripe width (0 = default 13).  Index-build
 * side: replaces nothing in the reference's query path; it is the exporter SURVEY.md §7 step 2. */
int xgm_segment_build(const xgm_raw_postings* raw, uint32_t stripe_bits, const char* out_path);

/* Same, reading the "XGMRAW1" file form. */
int xgm_segment_build_from_file(const char* raw_path, uint32_t stripe_bits, const char* out_path);

/* Export a committed glass shard of the reference straight from its on-disk B-trees (postlist.glass,
 * position.glass, iamglass — formats: reference src/xapian/backends/glass/glass_table.h:62-330,
 * glass_version.cc:100-235, glass_postlist.cc:677-695, glass_positionlist.cc:36-133, common/pack.h,
 * common/bitstream.cc) into a device segment: the exporter without Xapian in the loop (a sequential
 * scan of the leaf blocks instead of a B-tree cursor and a virtual call per posting).  The shard must
 * not be modified while it is read (hold Xapiand's shard lock, or export a checked-in revision). */
int xgm_segment_build_from_glass(const char* glass_dir, uint32_t stripe_bits, const char* out_path);

/* Refresh the segment of a shard that moved to a newer revision, reading from glass only what changed.  Contract: every
 * document with docid < first_changed_docid is the same in the old segment's revision and in glass now (Xapiand knows the
 * smallest docid its write-ahead log touched since a revision: reference src/database/wal.cc; appended documents only →
 * old lastdocid + 1).  The blocks of the stripes below the floor's are copied verbatim from the old segment, its postings
 * between that stripe's start and the floor are re-encoded together with what glass holds from the floor on (posting
 * chunks / position lists below it are skipped by their headers / keys; terms whose positions the old segment had dropped
 * are read in full): the output is byte for byte what xgm_segment_build_from_glass writes for the new revision.
 * stripe_bits: 0 or the old segment's.  XGM_E_INVALID when the part of the contract that is cheap to check fails (document
 * lengths below the floor differ, floor beyond the old segment, another stripe width, the position table appeared or
 * vanished): fall back to the full export.  Replaces: the reference reopening a shard whose revision changed
 * (src/database/handler.cc:1282, 1333). */
int xgm_segment_refresh_from_glass(const char* old_segment_path, const char* glass_dir, uint32_t first_changed_docid,
                                   uint32_t stripe_bits, const char* out_path);

/* One value slot of a committed glass shard as a column file (widening row (f).3: what a value sort / collapse on the device
 * will read): "XGMCOL1\0", u32 slot, u32 lastdocid, u32 n_distinct, u32 0, then u32 ord[lastdocid + 1] — 0 when the document
 * has no value, else 1 + the rank of its value among the slot's distinct values in bytewise order (what Xapian's value sorts
 * compare, reference src/xapian/matcher/msetcmp.cc:64-101) — then u64 off[n_distinct + 1] and the distinct values, ascending.
 * Read straight from the value chunks of postlist.glass (backends/glass/glass_values.h:41-47, glass_values.cc:72-95).
 * Replaces: Document::get_value per candidate in the matcher (matcher/matcher.cc:509-517). */
int xgm_glass_export_column(const char* glass_dir, uint32_t slot, const char* out_path);

/* The committed revision and statistics of a glass shard, from its version file alone (what
 * Database::get_revision / get_doccount / get_lastdocid / get_total_length would answer): the key under
 * which an exported segment is cached, cheap enough to poll.  Any output pointer may be NULL. */
int xgm_glass_info(const char* glass_dir, uint64_t* revision, uint32_t* doccount, uint32_t* lastdocid,
                   uint64_t* total_length);

/* Same reader, writing the "XGMRAW1" file form (what the iterator-based exporter produces): used to
 * verify the reader byte for byte against an export made through Xapian's public iterators. */
int xgm_glass_export_raw(const char* glass_dir, const char* raw_path);

/* Host-side decode of one term's postings out of a segment FILE (index utility used to verify an
 * export; not a search path).  Returns the number of postings written (<= cap) or < 0. */
int64_t xgm_segment_decode_term(const char* segment_path, const char* term, size_t len,
                                uint32_t* did, uint32_t* wdf, uint64_t cap);

/* ---- index lifetime ---------------------------------------------------------------------------*/

/* Load a segment file into HBM on `device`.  `revision` must equal the segment's revision unless it
 * is UINT64_MAX (don't care).  device == XGM_DEVICE_NONE loads the dictionary and statistics only
 * (host memory): xgm_lookup_term / xgm_index_termfreqs / xgm_plan_query work — a front end can plan
 * where no GPU is — and every search returns XGM_E_NO_DEVICE.  Replaces: opening GlassPostList cursors per query
 * (reference src/xapian/backends/glass/glass_database.cc:861-877, glass_postlist.cc:696-747). */
int xgm_index_open(const char* segment_path, int device, uint64_t revision, xgm_index** out);

/* Synthetic-corpus segment built directly in HBM by the GPU segment builder (bench/test tooling;
 * the corpus is tools/xgm_corpus.h).  Global docs g in 1..n_docs_global with
 * (g-1) % n_shards == shard land in this shard with local id (g-1)/n_shards + 1
 * (reference src/xapian/backends/multi.h:38-73). */
typedef struct {
    uint64_t seed;
    uint32_t vocab, len_lo, len_hi;
    uint64_t n_docs_global;
    uint32_t n_shards, shard;
    uint32_t stripe_bits;        /* 0 = default */
    uint32_t with_positions;
} xgm_synth_params;
int xgm_index_build_synthetic(const xgm_synth_params* p, int device, xgm_index** out);

void xgm_index_close(xgm_index*);

/* Write the device-resident segment back to a file (round-trips with xgm_index_open). */
int xgm_index_save(const xgm_index*, const char* segment_path);

typedef struct {
    uint32_t n_terms, lastdocid, doccount, has_positions;
    uint64_t total_length, revision, n_postings, n_positions, n_blocks;
    uint64_t device_bytes;       /* HBM held by the segment                                        */
    uint64_t payload_bytes;      /* bit-packed posting payload only                                 */
    uint32_t stripe_bits, block_size;
    uint32_t doclen_lower_bound, wdf_upper_bound;
} xgm_index_info;
int xgm_index_get_info(const xgm_index*, xgm_index_info* out);

/* Run all work of this index on the caller's HIP stream (hipStream_t cast to void*; NULL = the
 * index's own stream).  Lets a host that owns a stream (torch, the server) order our kernels with
 * its copies/collectives without extra synchronisation. */
int xgm_index_set_stream(xgm_index*, void* hip_stream);

/* Server mode (opt-in): Xapiand's HTTP worker threads call get_mset one query at a time (reference src/manager.cc:161,
 * src/database/handler.cc:1338).  With max_batch > 0, single-query calls (xgm_search, xgm_search_batch / xgm_get_mset_batch
 * with nq == 1) from any number of threads are queued and a dispatcher thread owned by the index launches whatever has
 * accumulated — up to max_batch queries, of any mix of shapes — as one batch, then returns each caller its own hits and
 * return code.  No timer: while one batch runs the next one fills — and is launched as soon as the dispatcher has cut it (two
 * batches in flight, their kernels back to back on one stream; a completer thread hands the rows back and wakes exactly the callers of
 * the finished batch; a small batch keeps the units-per-query ratio of a full one).  max_batch == 0 switches it off (the default).  Like
 * xgm_index_open / xgm_index_close the call is externally serialised with searches on the index (the matcher hook makes it once, when
 * a shard is registered). */
int xgm_index_set_batching(xgm_index*, uint32_t max_batch);

/* Dictionary lookup.  Replaces GlassPostListTable::get_freqs
 * (reference src/xapian/backends/glass/glass_postlist.cc:151-192) and get_wdf_upper_bound
 * (glass_database.cc:823-830).  Returns XGM_OK and termfreq 0 when the term is absent. */
int xgm_lookup_term(const xgm_index*, const char* term, size_t len, uint32_t* term_id,
                    uint32_t* termfreq, uint32_t* collfreq, uint32_t* wdf_ub);

/* Bulk per-term statistics for the cross-shard stats merge (Weight::Internal::operator+=,
 * reference src/xapian/weight/weightinternal.cc:55-71): copies termfreq[n_terms] (host). */
int xgm_index_termfreqs(const xgm_index*, uint32_t* termfreq, uint32_t cap);

/* OP_WILDCARD with a fixed prefix ("prefix*": Xapiand's DSL emits it for `*` values and partial terms, reference
 * src/query_dsl.cc:305, 634, 668, 724): the shard's terms that start with `prefix`, in the term order the reference's
 * Context<T>::expand_wildcard walks (db.open_allterms(prefix), src/xapian/api/queryinternal.cc:246-315).  Writes the first
 * min(cap, *n_total) term ids; *n_total = how many there are.  The caller applies the wildcard's limit (WILDCARD_LIMIT_ERROR /
 * _FIRST / _MOST_FREQUENT) and lowers the set to an OP_SYNONYM group, which the device weighs as the reference's
 * SynonymPostList does (QueryWildcard::postlist, queryinternal.cc:1441-1495). */
int xgm_expand_prefix(const xgm_index*, const char* prefix, size_t len, uint32_t cap, uint32_t* term_ids, uint32_t* n_total);
/* bytes (borrowed, valid while the index is open), termfreq and collection frequency of a term id */
int xgm_term_info(const xgm_index*, uint32_t term_id, const char** bytes, size_t* len, uint32_t* termfreq, uint32_t* collfreq);

/* ---- query planning (host) --------------------------------------------------------------------*/

/* What the hook extracts from an eligible Xapian::Query (reference src/xapian/query.h:575-622:
 * get_type / get_num_subqueries / get_subquery / get_leaf_wqf). */
typedef struct {
    uint32_t op;                          /* XGM_OP_*                                              */
    uint32_t n_terms;
    const char* terms[XGM_MAX_TERMS];     /* in QUERY order (phrase order for PHRASE)              */
    uint32_t term_len[XGM_MAX_TERMS];
    uint32_t window;                      /* PHRASE / NEAR: 0 = n_terms (exact phrase)             */
    uint32_t first, maxitems, check_at_least;
    /* BM25 parameters (reference src/xapian/weight.h:635-667 defaults 1, 0, 1, 0.5, 0.5) */
    double k1, k2, k3, b, min_normlen;
    uint32_t n_required;                  /* XGM_OP_AND_NOT / AND_MAYBE / FILTER: terms of the left-hand AND (>= 1) */
    uint32_t n_tree;                      /* XGM_OP_TREE: ops of the post-order program below                       */
    uint32_t wqf[XGM_MAX_TERMS];          /* within-query frequency of terms[i] (QueryTerm::get_wqf); 0 = 1         */
    xgm_tree_op tree[XGM_MAX_TREE];
    double tree_scale[XGM_MAX_TREE];      /* XGM_T_SCALE ops: the factor                                            */
    uint32_t replay;                      /* XGM_REPLAY_BATCH_* bits: answer this query as the reference's own collation does (copied into the plan) */
    uint32_t reserved;
} xgm_query_desc;

/* Collection statistics merged over all shards of the index, i.e. what Enquire::add_prepared_mset
 * accumulates (reference src/xapian/api/enquire.cc:385-394).  NULL → the shard's own. */
typedef struct {
    uint64_t total_length;
    uint32_t collection_size;
    uint32_t full_db_has_positions;       /* any shard has positions (handler.cc:1374)             */
    uint32_t termfreq[XGM_MAX_TERMS];     /* merged termfreq of desc->terms[i] (query order)       */
} xgm_global_stats;

typedef struct {
    uint32_t term_id;        /* id in THIS shard's dictionary (UINT32_MAX = absent in the shard)   */
    uint32_t phrase_index;   /* PHRASE: position of this leaf in the phrase (query order)          */
    double termweight;       /* BM25Weight::init result (host libm log), bm25weight.cc:46-115      */
} xgm_term;

/* The executable plan.  terms[] are in MultiAndPostList order for AND/PHRASE (ascending SHARD-LOCAL
 * termfreq, reference src/xapian/matcher/multiandpostlist.h:117-130) and in query order for OR.
 * sum_prog is the weight summation in post-order: entry >= 0 pushes terms[entry]'s weight,
 * XGM_SUM_ADD pops r, pops l, pushes l + r.  For AND this is the left-deep chain of
 * MultiAndPostList::get_weight (multiandpostlist.cc:150-160); for OR the Huffman-shaped OrPostList
 * tree of OrContext::postlist (reference src/xapian/api/queryinternal.cc:440-489). */
#define XGM_SUM_ADD (-1)
typedef struct {
    uint32_t op, n_terms;
    xgm_term terms[XGM_MAX_TERMS];
    int8_t sum_prog[2 * XGM_MAX_TERMS];
    uint32_t sum_len;
    uint32_t window;             /* PHRASE only; == n_terms means exact phrase                     */
    uint32_t phrase_active;      /* 0: PHRASE degraded to AND (no positions anywhere)              */
    double len_factor, k1, b, min_normlen;
    uint32_t first, maxitems, check_at_least;
    double max_possible;         /* Σ get_maxpart (bm25weight.cc:183-207); MSet field only         */
    uint32_t req_mask;           /* bit p: terms[p] must index the document (all terms for AND / PHRASE, 0 for OR) */
    uint32_t neg_mask;           /* bit p: terms[p] must NOT index it (right-hand side of AND_NOT)   */
    /* XGM_OP_TREE: the leaves of the lowered tree are GROUPS of terms — a term, or the members of an OP_SYNONYM that
     * share one weight over their summed wdf — combined by binary nodes in evaluation order.  Operand < n_groups: a
     * group, else n_groups + an earlier node.  A node's value is (matches, weight, weighted leaves matched):
     *   AND  l & r,  l + r          OR   l | r,  l + r (an absent side adds nothing)      AND_NOT  l & !r,  l
     *   MAYBE  l,  l + r where r matches (AndMaybePostList::get_weight, andmaybepostlist.cc:57-64)
     * n-ary conjunctions are chains in MultiAndPostList order, disjunctions the Huffman tree of OrContext::postlist, both
     * ordered by the sub-trees' termfreq ESTIMATES as the reference does. */
    uint32_t tree_len, n_groups, tree_root, total_subqs;
    uint32_t group_scored;       /* bit g: group g carries weight (counts in percentages)          */
    uint8_t group_of[XGM_MAX_TERMS];
    double group_weight[XGM_MAX_TERMS];
    uint8_t tree_op[XGM_MAX_TREE], tree_a[XGM_MAX_TREE], tree_b[XGM_MAX_TREE];
    /* PostList::get_termfreq_min / _est / _max of the tree the reference builds for this query on this shard — the
     * static inputs of MSet::get_matches_lower_bound / _estimated / _upper_bound (protomset.h:484-619; xgm_mset_bounds) */
    uint32_t est_min, est_est, est_max;
    /* XGM_REPLAY_BATCH_* bits (0 = none): the batch entry points answer this query as the reference's own collation would — see below */
    uint32_t replay;
} xgm_query;

/* The reference-identical answer INSIDE a batch (xgm_search_batch*, xgm_get_mset_batch*, the dispatcher of xgm_index_set_batching): what
 * xgm_search_replay gives one query at a time, at batch throughput.
 *   XGM_REPLAY_BATCH_FROZEN   OP_PHRASE / OP_NEAR: the page, weights and max_attained of the REFERENCE — SelectPostList's frozen weight
 *                             included (matcher/selectpostlist.cc:28-55): the query's units list their first matches in docid order with
 *                             their successors in the conjunction, one wave per query replays ProtoMSet over them (xgm_frozen.hip).
 *                             hdr->matches_exact may be a lower bound (units stop listing early).  Ignored on other operators.  Queries
 *                             the listing kernel does not take (terms without probe containers / flat arrays, > 4 terms, a page > 64,
 *                             check_at_least beyond the page) are answered by xgm_search_replay when the batch is collected: same rows.
 * xgm_batch_known (below) hands out ProtoMSet's known_matching_docs per query where a mode computed it. */
#define XGM_REPLAY_BATCH_FROZEN 1u
/*   XGM_REPLAY_BATCH_COUNT    ProtoMSet's known_matching_docs, EXACT (what MSet::get_matches_lower_bound / _estimated are derived from, protomset.h:484-619:
 *                             Xapiand's HTTP "total").  With XGM_REPLAY_BATCH_FROZEN on a positional query: the listing units walk their whole docid
 *                             range (every document of the conjunction is tested: the match count is exact) — slower than the page alone, still one
 *                             launch per batch.  On other operators: the page is the batch's, the count comes from xgm_search_replay(XGM_REPLAY_COUNT)
 *                             when the batch is collected. */
#define XGM_REPLAY_BATCH_COUNT 2u


#define XGM_N_AND 1
#define XGM_N_OR 2
#define XGM_N_ANDNOT 3
#define XGM_N_MAYBE 4

/* Lower a query description to a plan against this shard.  Replaces, for the supported shapes,
 * LocalSubMatch::get_postlist / open_post_list (reference src/xapian/matcher/localsubmatch.cc:164-
 * 196, 231-309), Query::Internal::postlist (src/xapian/api/queryinternal.cc:1049, 2083, 2193,
 * 2300-2354) and BM25Weight::init.  Returns XGM_UNSUPPORTED for shapes the device path declines
 * (duplicate terms, > XGM_MAX_TERMS leaves, first+maxitems > XGM_MAX_K, k2 != 0, wdf too large...). */
int xgm_plan_query(const xgm_index*, const xgm_query_desc*, const xgm_global_stats*, xgm_query* out);

/* ---- search -----------------------------------------------------------------------------------*/

typedef struct {
    uint32_t docid;              /* shard-local (xgm_search*) or global (xgm_merge_shards*)        */
    uint32_t subqs_matched;      /* leaves matching this doc (PostList::count_matching_subqs)      */
    double weight;
} xgm_hit;                       /* 16 bytes                                                        */

#define XGM_MATCHES_LOWER_BOUND (1ull << 63)
#define XGM_MATCHES_COUNT(m) ((m) & ~XGM_MATCHES_LOWER_BOUND)

typedef struct {
    uint32_t n_hits;             /* hits written: min(matches, first + maxitems)                   */
    uint32_t max_weight_subqs_matched; /* of the top-weighted doc (protomset.h:174-183)            */
    uint64_t matches_exact;      /* exact match count (the reference only estimates it); with XGM_MATCHES_LOWER_BOUND
                                    set: a lower bound — a positional query (PHRASE / NEAR) whose check_at_least lies
                                    within the page drops candidates that cannot enter the top k BEFORE testing their
                                    positions (Xapiand's default, check_at_least = 0; ask check_at_least >= doccount
                                    for the exact count) */
    double max_attained;         /* weight of the best doc, 0 if none                              */
    double max_possible;
} xgm_result_hdr;                /* 32 bytes                                                        */

/* MSet::get_matches_lower_bound / _estimated / _upper_bound as ProtoMSet::finalise derives them (protomset.h:484-619, no
 * collapsing / decider / percent cut-off) from the plan's static bounds and the search result.  The reference's
 * known_matching_docs — how many documents its matcher happened to weigh before pruning — depends on its traversal and is
 * not reproduced: the number of documents RETURNED stands in for it (never larger), so the upper bound is the reference's,
 * the estimate is the reference's whenever its static estimate dominates (the common case), and the lower bound is valid
 * but may be looser.  The exact count is in hdr->matches_exact. */
void xgm_mset_bounds(const xgm_query* plan, const xgm_result_hdr* hdr, uint32_t* lower, uint32_t* estimated, uint32_t* upper);
/* ... with the matcher's known_matching_docs supplied: where the reference's main loop shows ProtoMSet EVERY matching document
 * (a sort the value leads — min_weight stays 0, matcher.cc:482-536 with protomset.h:249-283 — or check_at_least beyond the match)
 * that number is the exact match count (hdr->matches_exact) and the three figures are the reference's. */
void xgm_mset_bounds_known(const xgm_query* plan, const xgm_result_hdr* hdr, uint64_t known_matching_docs, uint32_t* lower, uint32_t* estimated,
                           uint32_t* upper);
/* known_matching_docs of the reference's matcher for a search by relevance over operators that visit every matching document (a term,
 * AND, FILTER, AND_NOT, PHRASE, NEAR), from the weights of ALL matching documents in ascending docid order.  A document counts when
 * its weight is >= min_weight (matcher.cc:500-505 `if (weight < min_weight) continue`, then ProtoMSet::add counts it); min_weight is 0
 * until ProtoMSet holds max_size = first + maxitems documents and check_at_least have been counted, and is then the weight of the worst
 * document kept — set when the heap is made (the (max_size + 1)-th document) and afterwards only by a document that REPLACES the worst
 * (protomset.h:340-400), so with check_at_least beyond the page it starts at the first replacement after check_at_least documents.
 * check_at_least as Enquire::get_mset clamps it.  Pinned to the compiled reference's own figures by
 * tests/test_oracle_vs_reference.py::test_known_matching_docs_is_a_function_of_the_match_in_docid_order. */
uint64_t xgm_known_matching_docs(const double* weights_in_docid_order, uint64_t n, uint32_t max_size, uint32_t check_at_least);
/* MSet::get_matches_estimated(): the estimate as the API reports it — rounded to the significant figures the bounds justify
 * (reference src/xapian/api/roundestimate.h:36-69).  Xapiand's HTTP "total" field is this number (src/server/http_client.cc:2554). */
uint32_t xgm_round_estimate(uint32_t lower, uint32_t upper, uint32_t estimated);

/* One query on one shard: hits[0 .. first+maxitems) sorted by (weight desc, docid asc) — the order
 * of msetcmp_by_relevance<true> (reference src/xapian/matcher/msetcmp.cc:55-62); the caller drops
 * the first `first`.  Replaces the hot loop of Matcher::get_local_mset + ProtoMSet. */
int xgm_search(xgm_index*, const xgm_query*, xgm_hit* hits, xgm_result_hdr* hdr);

/* ---- searches under a value sort, with spies, collapsed (widening row (f).3; on the MI355X since round 3) ------------------ */

/* Load the ordinals of a column file (xgm_glass_export_column) into HBM next to the index: 4 bytes per document.  The column's
 * lastdocid must be the index's (same shard revision).  Attaching a slot again replaces it. */
int xgm_index_attach_column(xgm_index* idx, const char* column_path);
/* The same from memory: ord[0 .. lastdocid] (ord[0] unused), ord[d] = 0 when document d has no value, else 1 + the index of its value among
 * the column's n_distinct distinct values in ascending byte order.  What the matcher hook attaches for a value slot it read through the
 * shard's ValueIterator, or for the keys a Xapian::KeyMaker makes of every document (Enquire::set_sort_by_key*, which is how Xapiand
 * sorts: reference src/database/handler.cc:1269); any slot number may be used for such a synthetic column. */
int xgm_index_attach_column_ordinals(xgm_index* idx, uint32_t slot, const uint32_t* ord, uint32_t n_ord, uint32_t n_distinct);

#define XGM_SORT_VALUE 1u                 /* Enquire::set_sort_by_value                  (msetcmp.cc:64-73)   */
#define XGM_SORT_VALUE_RELEVANCE 2u       /* Enquire::set_sort_by_value_then_relevance   (msetcmp.cc:75-86)   */
#define XGM_SORT_RELEVANCE_VALUE 3u       /* Enquire::set_sort_by_relevance_then_value   (msetcmp.cc:88-101)  */
typedef struct {
    uint32_t sort_by;                     /* XGM_SORT_*                                                        */
    uint32_t slot;                        /* value slot of an attached column                                  */
    uint32_t reverse;                     /* the API's `reverse` flag: larger values first                     */
    uint32_t reserved;
} xgm_sort_spec;

/* xgm_search for a planned query with a value sort in force: the first + maxitems best documents under the chosen comparison
 * (the caller drops the first `first`).  hit_ord[i] (may be NULL) receives the ordinal of hit i's value in the column — 0 = no
 * value, else 1 + index into the column file's distinct values: the MSet item's sort key.  hdr->max_attained is the best weight
 * of the WHOLE match, as the reference reports it under a value sort (ProtoMSet::update_max_weight sees every document,
 * protomset.h:174-183, 249-283), matches_exact the number of matching documents.  Every query shape xgm_search takes;
 * XGM_UNSUPPORTED when the slot has no column attached.
 * Replaces: the value-sorted branch of the matcher's main loop (matcher/matcher.cc:482-536) and
 * ValueStreamDocument::get_value per candidate (matcher/valuestreamdocument.cc). */
int xgm_search_sorted(xgm_index* idx, const xgm_query* q, const xgm_sort_spec* sort, xgm_hit* hits, uint32_t* hit_ord,
                      xgm_result_hdr* hdr);

/* nq searches under ONE sort in one launch: hits [nq][k_stride] (k_stride >= every first + maxitems), hit_ord [nq][k_stride] or NULL, hdrs [nq]
 * — what xgm_search_sorted answers for each, without a launch, a copy each way and a synchronisation per query: Xapiand's HTTP threads all sort by
 * the same few fields (DocMatcher::get_mset, reference src/database/handler.cc:1338, once per request).  XGM_UNSUPPORTED if any query is. */
int xgm_search_sorted_batch(xgm_index* idx, const xgm_query* qs, uint32_t nq, const xgm_sort_spec* sort, uint32_t k_stride, xgm_hit* hits,
                            uint32_t* hit_ord, xgm_result_hdr* hdrs);

/* ... with a Xapian::ValueCountMatchSpy (what Xapiand's AggregationMatchSpy derives from, src/aggregations/) on value slot
 * spy_slot in the same pass: counts[o] = matching documents whose value has ordinal o in that slot's column, counts[0] those
 * without a value; n_counts = the column's distinct values + 1; hdr->matches_exact is the spy's total.  The counts are those of
 * EVERY matching document.  The reference's matcher shows a spy every matching document where the value leads the sort
 * (XGM_SORT_VALUE, XGM_SORT_VALUE_RELEVANCE: matcher/protomset.h:268-275) or where the match does not exceed check_at_least;
 * elsewhere (sort == NULL = by relevance, or XGM_SORT_RELEVANCE_VALUE, with a larger match) what its spy sees depends on its
 * traversal, and a caller that needs the reference's own counts keeps such a search on the CPU (the matcher hook does).
 * Replaces: api/matchspy.cc:307-313 called per document from the matcher's loop (matcher/matcher.cc:519-527). */
int xgm_search_sorted_spy(xgm_index* idx, const xgm_query* q, const xgm_sort_spec* sort, xgm_hit* hits, uint32_t* hit_ord,
                          xgm_result_hdr* hdr, uint32_t spy_slot, uint32_t* counts, uint32_t n_counts);

/* xgm_search_sorted_batch with that spy on every search: counts [nq][n_counts], one row per query (Xapiand's `_aggregations` over a field
 * ride on every search of a dashboard: many HTTP threads, the same sort, the same spy slot — one launch).  sort may be NULL (by relevance), as
 * for the single search.  The same caveat on what a spy sees applies per query. */
int xgm_search_sorted_spy_batch(xgm_index* idx, const xgm_query* qs, uint32_t nq, const xgm_sort_spec* sort, uint32_t k_stride, xgm_hit* hits,
                                uint32_t* hit_ord, xgm_result_hdr* hdrs, uint32_t spy_slot, uint32_t* counts, uint32_t n_counts);

/* ... with Enquire::set_collapse_key(collapse_slot, collapse_max) in force (sort == NULL: ranked by relevance): of the documents
 * sharing a value in the collapse slot only the best collapse_max under the ranking stay; documents without a value are never
 * collapsed.  hit_collapse_ord[i] = ordinal of hit i's collapse key in that slot's column (MSetIterator::get_collapse_key),
 * hit_collapse_count[i] = matching documents of that key beyond the collapse_max that stay (get_collapse_count);
 * *collapsed_lower_bound = documents without a key + per key min(matches, collapse_max); hdr->matches_exact counts the
 * uncollapsed match.  Any output pointer but hits / hdr may be NULL.  These are the INTENDED semantics (the best documents of
 * a key stay): the reference snapshot's collapser keeps the last-seen ones instead, so a hook that needs
 * byte-compatibility with it keeps collapsed searches on the CPU.  Replaces: matcher/collapser.cc in ProtoMSet::process. */
int xgm_search_collapsed(xgm_index* idx, const xgm_query* q, const xgm_sort_spec* sort, uint32_t collapse_slot, uint32_t collapse_max,
                         xgm_hit* hits, uint32_t* hit_ord, uint32_t* hit_collapse_ord, uint32_t* hit_collapse_count, xgm_result_hdr* hdr,
                         uint64_t* collapsed_lower_bound);

/* nq searches under ONE collapse key (and one sort, or NULL = by relevance) in one launch: the per-hit arrays are [nq][k_stride],
 * hdrs and collapsed_lower_bound [nq]; same answers as xgm_search_collapsed for each.  XGM_UNSUPPORTED if any query is, or if a row of
 * per-key counters for every query would not fit (a collapse column of very many distinct keys under a large batch). */
int xgm_search_collapsed_batch(xgm_index* idx, const xgm_query* qs, uint32_t nq, const xgm_sort_spec* sort, uint32_t collapse_slot,
                               uint32_t collapse_max, uint32_t k_stride, xgm_hit* hits, uint32_t* hit_ord, uint32_t* hit_collapse_ord,
                               uint32_t* hit_collapse_count, xgm_result_hdr* hdrs, uint64_t* collapsed_lower_bound);

/* EVERY matching document of a planned query — no page, no pruning — in ASCENDING DOCID order, each with its weight and the number
 * of weighted leaves matching it: the sequence the reference's matcher loop is shown by its posting-list tree (Matcher::get_local_mset,
 * matcher/matcher.cc:482-536) before ProtoMSet, the collapser, the spies or a cut-off look at it.  The plan's first / maxitems /
 * check_at_least are ignored (positional queries test every candidate's positions).  hits[0 .. *n_matches) are written when
 * *n_matches <= cap; otherwise nothing is written and *n_matches tells the room needed (plan->est_max — the tree's own
 * get_termfreq_max — always suffices).  hdr: matches_exact = *n_matches, max_attained / max_weight_subqs_matched of the whole match,
 * n_hits = hits written.  Every query shape xgm_search takes (positional queries of up to 8 terms: beyond 3 the kernel takes every stripe
 * in several passes over narrower LDS tables); one query per call.
 * What it is for: answers that must be BYTE-COMPATIBLE with the reference where the reference's answer depends on its traversal —
 * known_matching_docs behind MSet::get_matches_lower_bound / _estimated (protomset.h:497-619; xgm_known_matching_docs), the frozen
 * weight of PHRASE / NEAR (selectpostlist.cc:28-55), the snapshot's collapser, cut-offs and spies by relevance — for matches of ANY
 * size (rounds 1-3 could fetch at most XGM_MAX_K of them).  The matcher hook replays the reference's own loop over this list. */
int xgm_search_all(xgm_index*, const xgm_query* q, xgm_hit* hits, uint64_t cap, uint64_t* n_matches, xgm_result_hdr* hdr);

/* The reference's own COLLATION of a search by relevance, replayed on the device over that list — without the list ever leaving HBM.
 * Matcher::get_local_mset's loop (matcher/matcher.cc:482-536) drops a document whose weight is below ProtoMSet's min_weight and shows
 * the others to ProtoMSet::add (matcher/protomset.h:340-400), which counts them (known_matching_docs), keeps the best
 * first + maxitems and raises min_weight when the heap is made and at every replacement — if check_at_least documents have been
 * counted by then.  That sequence decides MSet::get_matches_lower_bound / _estimated (Xapiand's HTTP "total",
 * src/server/http_client.cc:2554) and, for OP_PHRASE / OP_NEAR, the page itself: once min_weight is positive SelectPostList::vet
 * (matcher/selectpostlist.cc:28-55) weighs the next document of the underlying conjunction, caches that weight and serves it for
 * every later match.  One workgroup walks the docid-ordered match exactly so (xgm_replay.hip): only the page and the figures
 * cross PCIe.  q->first + q->maxitems and q->check_at_least (as Enquire::get_mset clamps it) are ProtoMSet's max_size and
 * check_at_least.
 *   XGM_REPLAY_COUNT          any shape xgm_search_all takes: hits[0 .. hdr->n_hits) = the page as ProtoMSet keeps it (the best
 *                             first + maxitems — what xgm_search returns), *known_matching_docs = ProtoMSet's count
 *                             (= xgm_known_matching_docs over the weights of xgm_search_all's list)
 *   XGM_REPLAY_FROZEN_WEIGHT  OP_PHRASE / OP_NEAR (<= 8 terms): the page, weights and count of the REFERENCE, frozen weight included
 * hdr: matches_exact = the exact match count, max_attained / max_weight_subqs_matched = ProtoMSet's max_weight and the leaves of the
 * document that set it (under XGM_REPLAY_FROZEN_WEIGHT this can be the frozen weight), n_hits = hits written. */
#define XGM_REPLAY_COUNT 0u
#define XGM_REPLAY_FROZEN_WEIGHT 1u
int xgm_search_replay(xgm_index*, const xgm_query* q, uint32_t mode, xgm_hit* hits, xgm_result_hdr* hdr, uint64_t* known_matching_docs);


/* nq queries in one launch; hits is [nq][k_stride] with k_stride >= max(first+maxitems). */
int xgm_search_batch(xgm_index*, const xgm_query* qs, uint32_t nq, uint32_t k_stride,
                     xgm_hit* hits, xgm_result_hdr* hdrs);

/* xgm_search_batch honouring the queries' XGM_REPLAY_BATCH_* bits with their figures handed back: known [nq] = ProtoMSet's known_matching_docs
 * (xgm_batch_known's array; may be NULL).  Like xgm_search_batch, a single-query call rides in the dispatcher's shared launches when
 * xgm_index_set_batching is on: the matcher hook's byte-compatible modes at the throughput of the others. */
int xgm_search_batch_known(xgm_index*, const xgm_query* qs, uint32_t nq, uint32_t k_stride, xgm_hit* hits, xgm_result_hdr* hdrs, uint64_t* known);

/* Same, but results stay in HBM: d_hits ([nq][k_stride] xgm_hit) and d_hdrs ([nq] xgm_result_hdr)
 * are DEVICE pointers owned by the caller (e.g. torch tensors feeding an RCCL all-gather).
 * Asynchronous on the index's stream. */
int xgm_search_batch_device(xgm_index*, const xgm_query* qs, uint32_t nq, uint32_t k_stride,
                            void* d_hits, void* d_hdrs);

/* The body of Enquire::get_mset for a batch, in one call: lower every description against this shard
 * (xgm_plan_query, with gs[q] = the merged statistics of query q when gs != NULL — what
 * Enquire::set_prepared_mset installed, reference src/xapian/api/enquire.cc:378-394) and search.  This is what
 * the matcher hook calls per shard; bench.py times it, so that dictionary lookups, BM25Weight::init and leaf
 * ordering are inside the measured step.  Returns XGM_UNSUPPORTED if any query's shape is declined. */
int xgm_get_mset_batch(xgm_index*, const xgm_query_desc* descs, const xgm_global_stats* gs, uint32_t nq,
                       uint32_t k_stride, xgm_hit* hits, xgm_result_hdr* hdrs);
int xgm_get_mset_batch_device(xgm_index*, const xgm_query_desc* descs, const xgm_global_stats* gs, uint32_t nq,
                              uint32_t k_stride, void* d_hits, void* d_hdrs);

/* A batch in flight: *_begin plans the work units, uploads, launches the kernels and the copy of the results into pinned host memory
 * the library owns, and returns at once; xgm_batch_end waits for THAT batch and hands out its results — hits [nq][k_stride] (the
 * valid prefix of row q is hdrs[q].n_hits) and hdrs [nq], valid until xgm_batch_release, which returns the batch's buffers to the
 * index.  xgm_batch_poll: 1 when the batch has finished, 0 when not.  Several batches may be in flight per index (each on a stream
 * of its own unless xgm_index_set_stream bound one): the host plans batch i + 1 while the GPU runs batch i, and consecutive batches
 * overlap on the chip.  This is how a server keeps the device busy AND gets every hit on the host: the dispatcher of
 * xgm_index_set_batching is built on it, bench.py times it.  Begin a batch only while fewer than 8 are unreleased on the index.
 * Replaces: Enquire::get_mset blocking its HTTP worker thread for the whole match (reference src/database/handler.cc:1338). */
typedef struct xgm_inflight xgm_inflight;
int xgm_search_batch_begin(xgm_index*, const xgm_query* qs, uint32_t nq, uint32_t k_stride, xgm_inflight** out);
int xgm_get_mset_batch_begin(xgm_index*, const xgm_query_desc* descs, const xgm_global_stats* gs, uint32_t nq, uint32_t k_stride,
                             xgm_inflight** out);
int xgm_batch_end(xgm_inflight*, const xgm_hit** hits, const xgm_result_hdr** hdrs);
/* After xgm_batch_end: *known = [nq] ProtoMSet's known_matching_docs of the queries that carried XGM_REPLAY_BATCH_* bits (0 for the others), valid until
 * xgm_batch_release; XGM_KNOWN_LOWER_BOUND set where the figure is a lower bound (a frozen-weight page whose units stopped listing early).
 * *known = NULL when no query of the batch asked for a replay. */
#define XGM_KNOWN_LOWER_BOUND (1ull << 63)
int xgm_batch_known(xgm_inflight*, const uint64_t** known);
int xgm_batch_poll(xgm_inflight*);
void xgm_batch_release(xgm_inflight*);

/* Merge per-shard results after the all-gather.  d_all_hits is [n_shards][nq][k_stride], d_all_hdrs
 * [n_shards][nq] (device); output [nq][k_stride] / [nq] (device) with GLOBAL docids
 * did = (local-1)*n_shards + shard + 1.  Replaces MSet::unshard_docids + Matcher::merge_mset +
 * MSet::Internal::merge_stats (reference src/xapian/api/mset.cc:367-395,
 * src/xapian/matcher/matcher.cc:653-781).  k[q] = first+maxitems of query q. */
int xgm_merge_shards_device(xgm_index*, const void* d_all_hits, const void* d_all_hdrs,
                            uint32_t n_shards, uint32_t nq, uint32_t k_stride, const uint32_t* k,
                            void* d_out_hits, void* d_out_hdrs);

/* The same after ONE all-gather per batch: every rank packs its results as one record — [nq][k_stride] xgm_hit immediately followed by
 * [nq] xgm_result_hdr, xgm_shard_record_bytes(nq, k_stride) bytes: hand xgm_*_batch_device d_hits = record, d_hdrs = record +
 * nq * k_stride * sizeof(xgm_hit) — and d_all_records is the n_shards records one after the other, as ncclAllGather / torch's
 * all_gather_into_tensor leave them.  (Two collectives per batch — hits and headers — cost a second launch + ring latency on xGMI.) */
size_t xgm_shard_record_bytes(uint32_t nq, uint32_t k_stride);
int xgm_merge_shards_packed_device(xgm_index*, const void* d_all_records, uint32_t n_shards, uint32_t nq, uint32_t k_stride,
                                   const uint32_t* k, void* d_out_hits, void* d_out_hdrs);

/* The whole per-shard protocol of the reference for a node whose shards all live in THIS process
 * (one xgm_index per shard, on any mix of devices): what DocMatcher does around Enquire in
 * reference src/database/handler.cc:1485-1549 — prepare_mset on every shard and add_prepared_mset
 * (merged statistics, src/xapian/api/enquire.cc:319-394), get_mset(0, first+maxitems) per shard,
 * unshard_docids and merge_mset (src/xapian/api/mset.cc:367-395, src/xapian/matcher/matcher.cc:
 * 653-781).  descs[q] is planned once per shard with the merged statistics; every shard searches its
 * batch on its own device, all launched before anything is waited for; the shards' top lists reach shards[0]'s
 * device through an RCCL all-gather (one shard per device: a communicator the library owns, librccl loaded on
 * first use; XGM_SHARDED_RCCL=0 disables it, =force uses it even on one device) or peer copies (several shards per
 * device), and are merged there.  Buffers, streams and the communicator persist between calls.
 * hits is [nq][k_stride] with GLOBAL docids ((local-1)*n_shards + shard + 1), best first+maxitems of
 * each query (the caller skips `first`, as with xgm_search); hdrs[q].matches_exact is the sum over
 * the shards.  Returns XGM_UNSUPPORTED if any shard declines any query (caller: CPU matcher). */
int xgm_search_sharded(xgm_index* const* shards, uint32_t n_shards, const xgm_query_desc* descs,
                       uint32_t nq, uint32_t k_stride, xgm_hit* hits, xgm_result_hdr* hdrs);

/* Diagnostics: out4 = {calls, calls whose exchange was an RCCL all-gather, devices, shards} of the shard list whose
 * shards[0] is this index. */
int xgm_debug_sharded_info(const xgm_index* shards0, uint64_t* out4);

/* OP_NEAR on a shard whose indexer may put DISTINCT TERMS AT ONE POSITION (Xapiand's schema can index several terms per token position):
 * NearPostList::test_doc then wants the terms at pairwise distinct positions inside the window and settles coinciding heads by its
 * duplicate-position step (reference src/xapian/matcher/nearpostlist.cc:106-140).  may_exist != 0: every later OP_NEAR on this index runs
 * the reference's procedure in full (the serial predicate near_colocated of xgm_posfilter.h; the answers are the reference's either way,
 * the fast predicate is simply not valid when heads can coincide).  Default 0: one term per position, the wave-parallel predicate. */
int xgm_index_set_near_colocated(xgm_index*, int may_exist);

/* Kernel timing with HIP events on the launch stream.  xgm_index_set_profiling(idx, 1) makes every
 * later xgm_search* call record an event pair around the dominant kernel (xgm_match_kernel) without
 * synchronising; xgm_last_kernel_ms waits for the recorded launches, returns their MEAN duration in
 * milliseconds (< 0 if none) and starts a new window. */
/* on: bit 0 = record the event pairs; bit 1 = launch the wave kernels' TALLYING instantiation, which also counts
 * what each work unit requests from memory (xgm_last_batch_traffic) — the same code plus scalar tallies; never
 * combined with a timing run. */
int xgm_index_set_profiling(xgm_index*, int on);
double xgm_last_kernel_ms(const xgm_index*);
/* Name of the match kernel the last xgm_search* call on this index launched ("xgm_andw_kernel",
 * "xgm_orw_kernel", "xgm_and_kernel", "xgm_match_kernel"; "" before the first search). */
const char* xgm_last_kernel_name(const xgm_index*);

/* Traffic model of the LAST batch launched on this index while the tallying instantiation was selected
 * (xgm_index_set_profiling bit 1): what the wave kernels requested from memory, tallied per work unit and summed
 * here (waits for the device).  out[0..n), n <= XGM_TRAFFIC_FIELDS:
 * [0] container bitmap words (4 B each, streamed)       [1] distinct 64-B sectors touched by the container probes
 * [2] posting-block payload words (4 B)                 [3] 12-byte block headers
 * [4] distinct 64-B sectors touched by doclen gathers   [5] other streamed 4-byte words (run table, directory, histogram)
 * [6] candidates written (16 B each)                    [7] PHRASE: positions of every query term in every
 *                                                           conjunction survivor (P of SURVEY.md §8(d))
 * [8] container probes issued (lanes, 1 B each)         [9] doclen gathers issued (lanes, 4 B each)
 * bench.py turns these into roofline.model_min_bytes. */
#define XGM_TRAFFIC_FIELDS 10
int xgm_last_batch_traffic(xgm_index*, uint64_t* out, uint32_t n);

/* Algorithmic bytes of a planned query on this shard, SURVEY.md §8(d):
 * Σ_t df_t·8 + S·4 (+ P·4) + k·16, with S and P taken from the last executed result header
 * when given (hdr may be NULL → only the postings term). */
uint64_t xgm_query_postings_bytes(const xgm_index*, const xgm_query*);

/* Diagnostics: decode one term's whole posting list on the DEVICE (kernel K1 alone) into host
 * arrays; returns df or < 0.  Used to verify a segment against its source postings. */
int64_t xgm_debug_decode_term_device(xgm_index*, uint32_t term_id, uint32_t* did, uint32_t* wdf, uint64_t cap);

/* Diagnostics: copy the dense doclen array (u32[lastdocid+1]) to the host; returns its length. */
int64_t xgm_debug_read_doclen(xgm_index*, uint32_t* out, uint64_t cap);

/* Diagnostics: copy one term's positions (flat u32, posting order, Σ wdf entries) to the host; returns
 * their number (0 when the segment has no positions) or < 0. */
int64_t xgm_debug_read_positions(xgm_index*, uint32_t term_id, uint32_t* out, uint64_t cap);

/* Diagnostics: mean host time of xgm_plan_query per query (microseconds) over `reps` passes of descs[0..nq). */
double xgm_debug_plan_us(const xgm_index*, const xgm_query_desc* descs, const xgm_global_stats* gs, uint32_t nq, uint32_t reps);

/* Diagnostics (host only, works on an XGM_DEVICE_NONE index): the launches a batch of planned queries is cut into,
 * one per kernel class present, as "<kernel>[:variant]*<queries>;..." in launch order; returns their number. */
int xgm_debug_batch_launches(const xgm_index*, const xgm_query* qs, uint32_t nq, char* out, uint32_t cap);

/* Diagnostics: nanoseconds the calling threads spent on the host side of batch calls since the last fetch:
 * out8[0] planning the descriptions (xgm_plan_query), [1] cutting the batch into work units, [2] staging, copies
 * and launches ([4] staging memcpys, [5] upload enqueue + events, [6] match launch, [7] merge launch), [3] number
 * of launches. */
int xgm_debug_host_ns(uint64_t* out8);

/* Diagnostics: out2 = {xgm_search_replay calls walked by one workgroup, calls walked by segments in parallel} of this process. */
int xgm_debug_replay_info(uint64_t* out2);
/* Diagnostics: out3 = {rows with XGM_REPLAY_BATCH_* bits the listing kernel took, rows it declined on the device, rows answered by xgm_search_replay when their
 * batch was collected} of this process. */
int xgm_debug_batch_replay_info(uint64_t* out3);
/* Diagnostics: out3 = {batches the dispatcher launched, requests it served, max_batch}. */
int xgm_debug_batching_info(const xgm_index*, uint64_t* out3);
/* Diagnostics / bench.py's server leg: n_threads host threads, each answering per_thread queries one at a time through
 * xgm_get_mset_batch(nq = 1); lat_us gets n_threads * per_thread latencies; returns the wall time in seconds (< 0: error). */
double xgm_debug_concurrent_searches(xgm_index*, const xgm_query_desc* descs, const xgm_global_stats* gs, uint32_t n,
                                     uint32_t n_threads, uint32_t per_thread, uint32_t k, double* lat_us);

/* Section cycle counters of the -DXGM_PHASE_TIMERS / -DXGM_MERGE_TIMERS / -DXGM_ORW_TIMERS measurement builds (tools/phase.py);
 * zeros in a product build. */
int xgm_debug_phase_cycles(unsigned long long* out8);
int xgm_debug_merge_cycles(unsigned long long* out8);
int xgm_debug_orw_phase_cycles(unsigned long long* out8);
/* The work units plan_batch would cut a batch into (tests/test_batch_plan.py), the disjunction's threshold seed and weight bounds
 * (tests/test_planner.py), and the last batch's per-unit cycle records (tools/units.py). */
int64_t xgm_debug_plan_batch(const xgm_index*, const xgm_query* qs, uint32_t nq, char* kernel, uint32_t* units, uint64_t cap);
int xgm_debug_or_bounds(const xgm_index*, const xgm_query* q, double* seed, double* ub, double* ub1);
int64_t xgm_debug_last_units(xgm_index*, unsigned long long* out, uint64_t cap);
int64_t xgm_debug_last_units2(xgm_index*, unsigned long long* out, unsigned long long* out_pos, uint64_t cap);

const char* xgm_last_error(void);
const char* xgm_version(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* XGM_H */
