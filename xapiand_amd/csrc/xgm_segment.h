/* Device segment format ("XGMSEG1"), shared by the host builder, the GPU synthetic builder, the
 * loader and the kernels.  One segment = one shard revision.  DESIGN.md §3 describes the layout and
 * why it looks like this; the short version:
 *
 *   - docid space is cut into STRIPES of W = 2^stripe_bits docids;
 *   - a posting list is a sequence of BLOCKS of <= 128 postings that never straddle a stripe;
 *   - a block stores its first docid in a header array and the rest as frame-of-reference bit-packed
 *     (gap-1) values plus bit-packed wdf values in a contiguous little-endian u32 payload;
 *   - positions (when present) are one array per term, 2 bytes wide when every position of the term fits
 *     (XGM_TF_POS16 — any document shorter than 65 536 tokens) and 4 bytes otherwise; the array of term t starts
 *     at BYTE term_pos[t] of the section, and a posting's positions start at entry
 *     blk_pos[b] + Σ wdf of the earlier postings in its block.  The section ends with XGM_POS_PAD spare bytes
 *     so that the positional filter may fetch a whole 32-byte window per (document, term).
 *
 * This replaces, for the query path, glass's chunked-varint posting lists and doclen list
 * (reference src/xapian/backends/glass/glass_postlist.cc:677-695 format comment) and its
 * interpolative-coded position table (glass_positionlist.cc:36-52).
 */
#ifndef XGM_SEGMENT_H
#define XGM_SEGMENT_H

#include <stdint.h>

#define XGM_SEG_MAGIC "XGMSEG1"
#define XGM_SEG_VERSION 2u             /* 2: positions stored 2 or 4 bytes wide per term, term_pos in bytes */
#define XGM_BLOCK 128u                 /* postings per block (two per lane of a wave64)          */
#define XGM_DEFAULT_STRIPE_BITS 13u    /* 8192 docids per stripe                                  */
#define XGM_MIN_STRIPE_BITS 8u
#define XGM_MAX_STRIPE_BITS 13u        /* bounded by the LDS tables of the match kernel           */
#define XGM_WORD_PAD 8u                /* zero words after the payload so 2-word windows may overrun */

/* blk_meta: bits 0-7 count-1, 8-15 gap bit width (0..32), 16-23 wdf bit width (0..32) */
#define XGM_META(count, bwg, bww) (((count)-1u) | ((uint32_t)(bwg) << 8) | ((uint32_t)(bww) << 16))
#define XGM_META_COUNT(m) (((m)&0xFFu) + 1u)
#define XGM_META_BWG(m) (((m) >> 8) & 0xFFu)
#define XGM_META_BWW(m) (((m) >> 16) & 0xFFu)

/* term_flags */
#define XGM_TF_POS_OK 1u               /* every posting has exactly wdf positions → phrase capable */
#define XGM_TF_POS16 2u                /* the term's positions are stored as u16 (else u32)         */
#define XGM_POS_PAD 32u                /* spare bytes after the positions                           */

enum xgm_section {
    XGM_S_DOCLEN = 0,   /* u32[lastdocid+1]                                    */
    XGM_S_TERM_DF,      /* u32[n_terms]   termfreq                              */
    XGM_S_TERM_CF,      /* u32[n_terms]   collection frequency (Σ wdf)          */
    XGM_S_TERM_WDFUB,   /* u32[n_terms]   glass-style wdf upper bound           */
    XGM_S_TERM_FLAGS,   /* u32[n_terms]                                         */
    XGM_S_TERM_BLK,     /* u64[n_terms+1] first block of each term              */
    XGM_S_TERM_WORD,    /* u64[n_terms+1] first payload word of each term       */
    XGM_S_TERM_POS,     /* u64[n_terms+1] BYTE offset of each term's position array */
    XGM_S_BLK_FIRST,    /* u32[n_blocks]  first docid of the block              */
    XGM_S_BLK_META,     /* u32[n_blocks]                                        */
    XGM_S_BLK_WORD,     /* u32[n_blocks]  payload word offset, relative to term */
    XGM_S_BLK_POS,      /* u32[n_blocks]  position ENTRY offset, relative to term */
    XGM_S_WORDS,        /* u32[n_words + XGM_WORD_PAD]                          */
    XGM_S_POSITIONS,    /* per-term u16 / u32 arrays + XGM_POS_PAD bytes         */
    XGM_S_STR_OFF,      /* u64[n_terms+1] (host only)                           */
    XGM_S_STR_BYTES,    /* term bytes, sorted (host only)                       */
    XGM_S_COUNT
};

typedef struct {
    char magic[8];
    uint32_t version, stripe_bits, block_size, n_terms;
    uint32_t lastdocid, doccount, has_positions, doclen_lower_bound;
    uint32_t wdf_upper_bound, doclen_upper_bound;    /* (doclen_upper_bound: the wdf bound of an OP_SYNONYM, weight.cc:86-115) */
    uint64_t total_length, revision, n_postings, n_positions, n_blocks, n_words;
    uint64_t file_bytes;
    uint64_t sec_off[XGM_S_COUNT];     /* byte offset from the start of the blob, 256-B aligned   */
    uint64_t sec_bytes[XGM_S_COUNT];
} xgm_seg_header;

/* Device-side view: raw pointers into the HBM blob. */
typedef struct {
    const uint32_t* doclen;
    const uint64_t* term_blk;
    const uint64_t* term_word;
    const uint64_t* term_pos;
    const uint32_t* blk_first;
    const uint32_t* blk_meta;
    const uint32_t* blk_word;
    const uint32_t* blk_pos;
    const uint32_t* words;
    const unsigned char* positions;
    const uint32_t* term_flags;
    uint32_t stripe_bits;
    uint32_t lastdocid;
    /* Probe containers (built in HBM when the index is opened, never stored in the segment file): for
     * every term dense enough to average >= XGM_DENSE_MIN_AVG postings per stripe, each non-empty
     * (term, stripe) run also exists uncompressed as  u32 bits[W/32]  (membership bitmap) followed by
     * u8 wdf1[W]  (wdf + 1 of the docid at that slot, 0 = absent).  A conjunction probes wdf1[slot]
     * with ONE load per candidate instead of decoding the run's blocks; when every term of a query is
     * dense the candidates are the AND of the bitmaps.  9 KiB per run at W = 8192: HBM is spent to
     * make the hot path O(candidates). */
    const uint32_t* dense_id;       /* [n_terms] dense index of the term, 0xFFFFFFFF if it has none   */
    const uint32_t* dense_dir;      /* [n_dense][n_stripes] container offset in 16-byte units, 0 = none */
    const unsigned char* dense_data;
    uint32_t n_dense;
    uint32_t n_stripes;
    uint32_t dense_pos;             /* the containers end with u32 pos_base[W/64]: position-entry offset (relative to the term) of the
                                       first posting of each 64-slot bucket — the positional filter on the probe path */
    /* Document lengths once more, narrow (built in HBM next to the containers): doclen[d] - doclen_base as u8 when the shard's
     * lengths span < 256, as u16 when < 65 536 (doclen_narrow_bits = 8 / 16; 0: none).  A gather of the lengths of densely matching
     * documents then touches a quarter / half of the 64-byte sectors the u32 array costs (xgm_dense_unit). */
    const unsigned char* doclen_narrow;
    uint32_t doclen_narrow_bits, doclen_base;
    uint32_t dense_plane;           /* byte offset inside a container of u32 bits2[W/32]: the documents whose wdf is >= 2 — the disjunction's
                                       weight bound of a (document, term) is then the wdf = 1 bound or the term's maximum (0: no plane) */
    /* Flat posting arrays (round 4; built in HBM next to the containers, never stored): every term WITHOUT containers (the long tail:
     * under XGM_DENSE_MIN_AVG postings per stripe) whose wdf fits a byte also exists decoded — docids ascending, one wdf byte each,
     * 5 bytes per posting (1.9 GB for the 375 M such postings of the 10 M-document corpus).  Stripe alignment leaves such a term ~1 block
     * of a dozen postings per stripe it touches, and a conjunction led by it paid a header + payload round trip per stripe for a handful
     * of candidates (41 % of C2's wave cycles, tools/qcost.py); from the flat array a unit streams the term's postings of its docid range
     * 64 per round, whatever stripes they fall in (xgm_flat_unit).  flat_off[t] .. flat_off[t + 1]: the term's slice (empty: no array). */
    const uint64_t* flat_off;       /* [n_terms + 1] or NULL */
    const uint32_t* flat_did;
    const unsigned char* flat_wdf;
    const uint32_t* flat_pos;       /* indexes with positions: position-entry offset (relative to the term's list) of every flat posting's first position, or NULL */
} xgm_seg_dev;

#define XGM_DENSE_MIN_AVG 32u          /* postings per stripe (on average) that make a term dense     */

static inline uint32_t xgm_bits_needed(uint32_t v) {
    uint32_t b = 0;
    while (v) { ++b; v >>= 1; }
    return b;
}

#endif /* XGM_SEGMENT_H */
