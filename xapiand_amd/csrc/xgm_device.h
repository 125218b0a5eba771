/* Host↔device structures of the match engine (not part of the public ABI). */
#ifndef XGM_DEVICE_H
#define XGM_DEVICE_H

#include <stdint.h>

#include "../../include/xgm.h"
#include "xgm_segment.h"

#define XGM_WG 256u                 /* threads per workgroup of the match kernel (4 waves)        */
#define XGM_WAVES (XGM_WG / 64u)
#define XGM_MERGE_CAP 8192u         /* candidates one merge workgroup can sort in LDS             */
#define XGM_OR_HIST 256u            /* weight-histogram buckets per query of xgm_orw_kernel        */
#define XGM_PHRASE_MAX_TERMS 8u     /* terms of a PHRASE / NEAR on the wave kernel (per-lane cursors: 5 bits each in one register) */
#define XGM_PHRASE_MAX_TERMS_WG 3u  /* ... on the workgroup kernel: position tables are 4 B/slot/term of LDS */

#define XGM_QF_PHRASE 1u            /* apply the positional filter                                 */
#define XGM_QF_EXACT 2u             /* window == n_terms: ExactPhrasePostList semantics            */
#define XGM_QF_EMPTY 4u             /* provably no match on this shard (absent AND term, ...)      */
#define XGM_QF_NEAR 8u              /* the positional filter is NearPostList's (any order, span < window) */
#define XGM_QF_NEAR_COLOC 256u      /* ... on a shard whose indexer may put DISTINCT TERMS AT ONE POSITION (xgm_index_set_near_colocated): NearPostList's
                                       duplicate-position step decides then (nearpostlist.cc:106-140) — the serial restatement near_colocated() */
#define XGM_QF_POSPRUNE 32u         /* positional query whose match count may be a lower bound: weigh first, test positions only of
                                       candidates that can still enter the top k (include/xgm.h, XGM_MATCHES_LOWER_BOUND) */
#define XGM_QF_DENSE 64u            /* a plain conjunction / FILTER of 2..4 terms that ALL have probe containers, k <= 64: its units run
                                       xgm_dense_unit (xgm_dense_body.inc) inside xgm_andw_kernel */
#define XGM_QF_FLAT 128u            /* a plain conjunction / FILTER of 2..4 terms, k <= 64, led by a term WITHOUT containers that has a flat posting array
                                       (every other term: containers or a flat array): its units run xgm_flat_unit (xgm_flat_body.inc) inside xgm_andw_kernel */
#define XGM_QF_TREE 16u             /* a nested query: match and weigh by the node program over term GROUPS */
#define XGM_QF_LIST_CONJ 512u       /* xgm_search_replay, frozen-weight mode: the match list of a positional query also carries the documents of the
                                       underlying conjunction that FAIL the positional test, flagged XGM_ALL_NOT_A_MATCH in their subqs word */
#define XGM_QF_COUNT_ALL 1024u      /* xgm_andw_list_kernel (XGM_REPLAY_BATCH_COUNT): the unit lists its first matches AND walks its whole range: the match count is exact */
#define XGM_ALL_NOT_A_MATCH 0x80000000u

/* Executable form of xgm_query, one per query of a batch, read with scalar loads. */
typedef struct {
    uint32_t op, n_terms, k, window;
    uint32_t flags, pad0;
    double len_factor, k1, b, min_normlen;
    double termweight[XGM_MAX_TERMS];
    uint32_t term_id[XGM_MAX_TERMS];      /* UINT32_MAX: absent in this shard                      */
    uint8_t phrase_index[XGM_MAX_TERMS];
    /* weight summation as a node list: node j (0 <= j < n_terms-1) = val[node_a[j]] + val[node_b[j]]
     * where val[0..n_terms) are the leaves and val[n_terms + j] the inner nodes; root = last node */
    uint8_t node_a[XGM_MAX_TERMS];
    uint8_t node_b[XGM_MAX_TERMS];
    /* the same summation "in place": step j does val[ip_a[j]] += val[ip_b[j]] over the T leaf slots
     * (an inner node lives in its left operand's slot); the root ends up in val[ip_root] */
    uint8_t ip_a[XGM_MAX_TERMS];
    uint8_t ip_b[XGM_MAX_TERMS];
    uint32_t ip_root;
    uint32_t n_nodes;                     /* additions of the summation program (leaves in it - 1)      */
    uint32_t sum_root;                    /* index (leaf or T + node) holding the document's weight     */
    uint32_t req_mask, neg_mask;          /* term t must / must not index a matching document (non-OR)  */
    uint32_t score_mask;                  /* term t is a weighted leaf (LeafPostList::count_matching_subqs) */
    uint32_t n_req, pad1;                 /* popcount(req_mask): the required terms are plan positions [0, n_req) */
    /* safe upper bound of leaf t's weight over the whole shard (0 for an absent term): drives the
     * MaxScore pruning of xgm_orw_kernel; never part of a result */
    double ub[XGM_MAX_TERMS];
    /* xgm_orw_kernel: the same bound for a document whose wdf is 1 (the containers' wdf >= 2 bitmap tells which bound applies), and a
     * first guess of the final k-th weight: weights of (wdf = 1, longest document) summed over the term subsets that — were the terms
     * independent — at least a few k documents match.  A guess only: the kernel repairs a guess that was too high (xgm_or.hip) */
    double ub1[XGM_MAX_TERMS];
    double theta_seed;
    /* XGM_QF_TREE (include/xgm.h, xgm_query): termweight[g] is then the weight of GROUP g */
    uint32_t tree_len, n_groups, tree_root, group_scored;
    uint8_t group_of[XGM_MAX_TERMS];
    uint8_t tnode_op[XGM_MAX_TREE], tnode_a[XGM_MAX_TREE], tnode_b[XGM_MAX_TREE];
} xgm_dev_query;

/* One top-k candidate: 16 bytes. */
typedef struct {
    uint64_t wbits;        /* IEEE bits of the (non-negative) weight: orders like the double       */
    uint32_t did;
    uint32_t subqs;
} xgm_cand;

/* ... under a value sort (xgm_match_sorted_kernel): two keys, larger first — (ordinal key, weight bits), or (weight bits, ordinal
 * key) when the weight leads; the ordinal key is the column's ordinal, complemented for an ascending sort.  24 bytes. */
typedef struct {
    uint64_t kw, kx;
    uint32_t did;
    uint32_t subqs;
    uint32_t cord;         /* ordinal of the document's collapse key (0: none / not collapsing) */
    uint32_t pad;
} xgm_cand_sorted;

/* One unit of work of the match kernels: a query and a contiguous range of docid stripes.  The host
 * cuts every query into units of roughly equal posting-block counts (heavy queries get more units)
 * and sorts the list heaviest-first, so the chip stays full until the end of the launch. */
typedef struct {
    uint32_t qi;           /* query of the batch                                                   */
    uint32_t s_begin;      /* first stripe                                                         */
    uint32_t s_end;        /* one past the last stripe                                             */
    uint32_t slot;         /* where the unit's candidates / header go: goff[qi] + index in query   */
} xgm_work;

/* Per work-unit summary written by the match kernel. */
typedef struct {
    uint64_t matches;
    uint32_t n_cand;
    uint32_t pad;
    uint64_t t_start, t_end;   /* s_memtime at unit start / end (diagnostics: occupancy timeline) */
    /* Traffic model of the wave kernels: what the unit REQUESTED from memory, tallied in scalar registers
     * (wave-uniform counts, no vector register cost).  Host: xgm_last_batch_traffic → bench.py's
     * roofline.model_min_bytes. */
    uint64_t c_pos;            /* PHRASE: positions of the query's terms in the conjunction's survivors (P of SURVEY §8(d)) */
    uint32_t c_bmp_words;      /* container bitmap words read (streamed, 16 B per lane)                         */
    uint32_t c_probes;         /* distinct 64-B memory sectors touched by the one-byte container probes (per round and term) */
    uint32_t c_blk_words;      /* bit-packed payload words of the posting blocks decoded (streamed)             */
    uint32_t c_hdrs;           /* 12-byte block headers read                                                     */
    uint32_t c_doclen;         /* distinct 64-B sectors touched by the doclen[] gathers (per round)              */
    uint32_t c_aux_words;      /* other streamed words: run-table pass over blk_first, container directory, histogram */
    uint32_t c_probes_raw;     /* container probes issued (lanes)                                                  */
    uint32_t c_doclen_raw;     /* doclen gathers issued (lanes)                                                    */
    uint32_t c_pad[2];
} xgm_group_hdr;

/* xgm_andw_kernel's LIST instantiation (the reference-identical batch mode of positional queries, include/xgm.h XGM_REPLAY_BATCH_FROZEN):
 * what a work unit reports instead of a top-k list — the unit's FIRST matches in docid order, each with the weight of the document of
 * the underlying conjunction that follows it inside the unit (SelectPostList::vet weighs that document when min_weight turns positive and
 * serves its weight from then on, selectpostlist.cc:28-55).  A unit stops after 2 (k + 1) matches: ProtoMSet's page is decided by the first
 * k + 1 matches of the QUERY, the frozen weight and at most k later matches (xgm_frozen.hip).  The entries live where the unit's candidates
 * would (cand_out + slot * k_stride, k_stride = XGM_PREFIX_CAND_STRIDE(k_max) candidates); the unit's xgm_group_hdr carries: matches = the
 * matches it found (| XGM_MATCHES_LOWER_BOUND when it stopped early), n_cand = entries written, pad = XGM_PFX_* flags, c_pos = weight bits
 * of the unit's first conjunction document. */
typedef struct {
    uint64_t wbits;        /* the match's own weight */
    uint64_t next_wbits;   /* weight of the conjunction's next document in the unit (a match or not); valid when has_next */
    uint32_t did;
    uint32_t has_next;     /* 0: the match is the unit's last conjunction document — the successor is the first one of a later unit */
} xgm_prefix_entry;        /* 24 bytes */
#define XGM_PFX_HAS_FIRST 1u    /* the unit holds a document of the conjunction (c_pos valid) */
#define XGM_PFX_COMPLETE 2u     /* the unit walked its whole docid range: `matches` is exact — and the entries are all its matches when matches == n_cand */
#define XGM_PFX_DECLINED 4u     /* the unit's query has no LIST body (neither xgm_dense_unit nor xgm_flat_unit): answered by the per-query replay */
/* per-row extra word of a batch with replay bits: known_matching_docs | ... */
#define XGM_EXTRA_LOWER_BOUND (1ull << 63)   /* = XGM_KNOWN_LOWER_BOUND */
#define XGM_EXTRA_FALLBACK (1ull << 62)      /* the device declined: xgm_batch_end answers the row with xgm_search_replay */
#define XGM_PREFIX_ENTRIES(k) (2u * ((k) + 1u))
#define XGM_PREFIX_CAND_STRIDE(k_max) (3u * ((k_max) + 1u))      /* 2 (k + 1) entries of 24 bytes in candidates of 16 */

/* xgm_andw_all_kernel (XGM_REPLAY_BATCH_COUNT on plain conjunctions): the units leave their top-k lists as always AND every match — docid, weight — in
 * docid order, for ProtoMSet's collation to be replayed over (xgm_count.hip).  A unit's list grows in chunks handed out of one arena by an atomic cursor:
 * chunk c holds XGM_ALL_CHUNK0 << c entries (13 chunks: 524 224 entries, more than a unit's 32 stripes can hold), its first entry in chunk_tab.  The
 * unit's header: matches = entries listed (exact), c_pad[0] = chunks, c_pad[1] = XGM_ALL_OVERFLOW when the arena ran out (the query is then counted
 * by the per-query replay when the batch is collected). */
#define XGM_ALL_CHUNKS 13u
#define XGM_ALL_CHUNK0 64u
#define XGM_ALL_OVERFLOW 1u
#define XGM_ALL_DECLINED 2u     /* the unit's query has no ALL body */
typedef struct {
    xgm_hit* arena;
    unsigned long long* cursor;        /* entries handed out (zeroed before the launch) */
    unsigned long long cap;            /* entries of the arena (< 2^32) */
    uint32_t* chunk_tab;               /* [n_work][XGM_ALL_CHUNKS] */
} xgm_all_out;

/* xgm_andw_kernel finishing its queries itself (xgm_unit_finish.h): the LAST unit of a query to arrive merges the units' lists into
 * the final hits — no merge launch.  arrive == NULL: the units only write their lists (xgm_merge_kernel follows). */
typedef struct {
    uint32_t* arrive;              /* [nq] zero between launches: units of the query that have written their list */
    const uint32_t* goff;          /* [nq + 1] unit slots of query qi: [goff[qi], goff[qi + 1]) */
    const double* max_possible;    /* [nq] or NULL */
    const uint32_t* row_of;        /* [nq] row of the caller's batch, or NULL (= qi) */
    xgm_hit* hits;                 /* [rows][k_stride_out] */
    xgm_result_hdr* hdrs;          /* [rows] */
    uint32_t k_stride_out, pad;
} xgm_fuse;

#endif
