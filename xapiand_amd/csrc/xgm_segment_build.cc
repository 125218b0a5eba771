/* Host-side segment builder / exporter back end: raw postings → "XGMSEG1" blob.
 *
 * Index-build side.  The query path never runs this code.  The GPU synthetic
 * builder (xgm_synth.hip) must produce bit-identical sections for the same postings; the tests
 * check that.
 */
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstring>

#include "xgm_internal.h"

namespace {

constexpr uint64_t kAlign = 256;

uint64_t align_up(uint64_t v) { return (v + kAlign - 1) / kAlign * kAlign; }

struct BitWriter {
    std::vector<uint32_t>& words;
    uint64_t base;       /* word index where the current section starts */
    explicit BitWriter(std::vector<uint32_t>& w) : words(w), base(w.size()) {}
    void begin(uint32_t n_values, uint32_t bw) {
        base = words.size();
        words.resize(base + ((uint64_t)n_values * bw + 31) / 32, 0u);
    }
    void put(uint32_t i, uint32_t bw, uint32_t v) {
        if (!bw) return;
        uint64_t bit = (uint64_t)i * bw;
        uint64_t w = base + (bit >> 5);
        uint32_t sh = (uint32_t)(bit & 31);
        words[w] |= v << sh;
        if (sh + bw > 32) words[w + 1] |= v >> (32 - sh);
    }
};

}  // namespace

/* A segment file is untrusted input: everything the loader, the planner and the kernels index by header counts is
 * checked against the section sizes here, with overflow-safe arithmetic, before the blob is adopted. */
int xgm_validate_header(const xgm_seg_header* h, uint64_t avail_bytes) {
    if (memcmp(h->magic, XGM_SEG_MAGIC, 8) != 0) return xgm_set_error(XGM_E_INVALID, "not an XGMSEG1 segment");
    if (h->version != XGM_SEG_VERSION) return xgm_set_error(XGM_E_INVALID, "segment version %u unsupported", h->version);
    if (h->block_size != XGM_BLOCK) return xgm_set_error(XGM_E_INVALID, "segment block size %u unsupported", h->block_size);
    if (h->stripe_bits < XGM_MIN_STRIPE_BITS || h->stripe_bits > XGM_MAX_STRIPE_BITS)
        return xgm_set_error(XGM_E_INVALID, "segment stripe_bits %u out of range", h->stripe_bits);
    if (h->file_bytes > avail_bytes) return xgm_set_error(XGM_E_INVALID, "segment truncated");
    for (int s = 0; s < XGM_S_COUNT; ++s)
        if (h->sec_off[s] > h->file_bytes || h->sec_bytes[s] > h->file_bytes - h->sec_off[s] || (h->sec_off[s] & 7u))
            return xgm_set_error(XGM_E_INVALID, "segment section %d out of bounds", s);
    /* section sizes against the element counts */
    const uint64_t T = h->n_terms, B = h->n_blocks;
    if (B >= 0xFFFFFFFFull || h->n_words >= (1ull << 40) || h->n_positions >= (1ull << 40))
        return xgm_set_error(XGM_E_INVALID, "segment counts out of range");
    const struct { int sec; uint64_t need; } want[] = {
        {XGM_S_DOCLEN, ((uint64_t)h->lastdocid + 1) * 4}, {XGM_S_TERM_DF, T * 4}, {XGM_S_TERM_CF, T * 4}, {XGM_S_TERM_WDFUB, T * 4},
        {XGM_S_TERM_FLAGS, T * 4}, {XGM_S_TERM_BLK, (T + 1) * 8}, {XGM_S_TERM_WORD, (T + 1) * 8}, {XGM_S_TERM_POS, (T + 1) * 8},
        {XGM_S_BLK_FIRST, B * 4}, {XGM_S_BLK_META, B * 4}, {XGM_S_BLK_WORD, B * 4}, {XGM_S_BLK_POS, B * 4},
        {XGM_S_WORDS, (h->n_words + XGM_WORD_PAD) * 4}, {XGM_S_POSITIONS, h->n_positions * 2 + (h->n_positions ? XGM_POS_PAD : 0)}, {XGM_S_STR_OFF, (T + 1) * 8}};
    for (const auto& w : want)
        if (h->sec_bytes[w.sec] < w.need) return xgm_set_error(XGM_E_INVALID, "segment section %d is smaller than its header counts imply", w.sec);
    if (h->doccount > h->lastdocid) return xgm_set_error(XGM_E_INVALID, "segment doccount exceeds lastdocid");
    return XGM_OK;
}

/* The tables behind the header: monotone offsets ending at the section sizes, blocks inside their term's payload and
 * docid range.  O(n_terms + n_blocks) once per load. */
int xgm_validate_blob(const XgmSegmentBlob& blob) {
    const xgm_seg_header* h = blob.header();
    const uint32_t T = h->n_terms;
    const uint64_t B = h->n_blocks;
    const uint64_t* so = blob.section<uint64_t>(XGM_S_STR_OFF);
    const uint64_t* tb = blob.section<uint64_t>(XGM_S_TERM_BLK);
    const uint64_t* tw = blob.section<uint64_t>(XGM_S_TERM_WORD);
    const uint64_t* tp = blob.section<uint64_t>(XGM_S_TERM_POS);
    if (so[0] != 0 || tb[0] != 0 || tw[0] != 0 || tp[0] != 0) return xgm_set_error(XGM_E_INVALID, "segment tables do not start at 0");
    for (uint32_t t = 0; t < T; ++t)
        if (so[t + 1] < so[t] || tb[t + 1] < tb[t] || tw[t + 1] < tw[t] || tp[t + 1] < tp[t])
            return xgm_set_error(XGM_E_INVALID, "segment term table %u is not monotone", t);
    if (so[T] > h->sec_bytes[XGM_S_STR_BYTES] || tb[T] != B || tw[T] != h->n_words || (tp[T] && tp[T] + XGM_POS_PAD > h->sec_bytes[XGM_S_POSITIONS]))
        return xgm_set_error(XGM_E_INVALID, "segment term tables do not end at the section sizes");
    const uint32_t* bf = blob.section<uint32_t>(XGM_S_BLK_FIRST);
    const uint32_t* bm = blob.section<uint32_t>(XGM_S_BLK_META);
    const uint32_t* bw = blob.section<uint32_t>(XGM_S_BLK_WORD);
    const uint32_t* bp = blob.section<uint32_t>(XGM_S_BLK_POS);
    const uint32_t* df = blob.section<uint32_t>(XGM_S_TERM_DF);
    const uint32_t* tf = blob.section<uint32_t>(XGM_S_TERM_FLAGS);
    for (uint32_t t = 0; t < T; ++t) {
        const uint64_t pwid = (tf[t] & XGM_TF_POS16) ? 2u : 4u;
        if ((tf[t] & XGM_TF_POS_OK) && (tp[t] % pwid)) return xgm_set_error(XGM_E_INVALID, "segment term %u: misaligned positions", t);
        const uint64_t words_t = tw[t + 1] - tw[t], pos_t = (tp[t + 1] - tp[t]) / pwid;
        uint64_t n = 0;
        uint32_t prev = 0;
        for (uint64_t b = tb[t]; b < tb[t + 1]; ++b) {
            const uint32_t meta = bm[b], cnt = XGM_META_COUNT(meta), g = XGM_META_BWG(meta), w = XGM_META_BWW(meta);
            if (cnt > XGM_BLOCK || g > 32u || w > 32u) return xgm_set_error(XGM_E_INVALID, "segment block %llu: bad meta", (unsigned long long)b);
            const uint64_t pw = (((uint64_t)cnt * g + 31u) >> 5) + (((uint64_t)cnt * w + 31u) >> 5);
            if ((uint64_t)bw[b] + pw > words_t) return xgm_set_error(XGM_E_INVALID, "segment block %llu: payload outside its term", (unsigned long long)b);
            if (bf[b] == 0 || bf[b] > h->lastdocid || (b > tb[t] && bf[b] <= prev)) return xgm_set_error(XGM_E_INVALID, "segment block %llu: docids out of order", (unsigned long long)b);
            if (pos_t && bp[b] > pos_t) return xgm_set_error(XGM_E_INVALID, "segment block %llu: positions outside its term", (unsigned long long)b);
            prev = bf[b];
            n += cnt;
        }
        if (n != df[t]) return xgm_set_error(XGM_E_INVALID, "segment term %u: block counts (%llu) != termfreq (%u)", t, (unsigned long long)n, df[t]);
    }
    return XGM_OK;
}

/* ---- the segment writer: terms are added in ascending byte order, each as (optionally) a run of blocks copied verbatim from an
 * older segment followed by postings encoded into new blocks.  xgm_build_segment_blob (everything encoded) and the incremental
 * refresh (xgm_glass.cc: unchanged stripes copied) go through the same code, so their outputs agree byte for byte. ---- */

int XgmSegmentWriter::begin(uint32_t sb, bool with_positions, uint32_t wdf_ub_of_the_database) {
    if (sb == 0) sb = XGM_DEFAULT_STRIPE_BITS;
    if (sb < XGM_MIN_STRIPE_BITS || sb > XGM_MAX_STRIPE_BITS)
        return xgm_set_error(XGM_E_INVALID, "stripe_bits %u out of range [%u, %u]", sb, XGM_MIN_STRIPE_BITS, XGM_MAX_STRIPE_BITS);
    stripe_bits = sb; has_pos = with_positions; wdf_ub_db = wdf_ub_of_the_database;
    return XGM_OK;
}

int XgmSegmentWriter::begin_term(const char* name, uint32_t len, bool pos_ok, bool pos16) {
    if (!term_df.empty()) {
        /* terms must be strictly ascending bytewise */
        const char* prev = str_bytes.data() + str_off.back();
        const uint32_t la = (uint32_t)(str_bytes.size() - str_off.back());
        int c = memcmp(prev, name, std::min(la, len));
        if (c > 0 || (c == 0 && la >= len)) return xgm_set_error(XGM_E_INVALID, "terms not sorted at index %zu", term_df.size());
    }
    str_off.push_back(str_bytes.size());
    str_bytes.insert(str_bytes.end(), name, name + len);
    term_blk.push_back(blk_first.size());
    term_word.push_back(words.size());
    t_pos_ok = has_pos && pos_ok; t_pos16 = t_pos_ok && pos16;
    const size_t pw = t_pos16 ? 2u : 4u;
    positions.resize((positions.size() + pw - 1) / pw * pw, 0);          /* the term's array is aligned to its entry width */
    term_pos.push_back(positions.size());
    t_entries = 0; t_df = 0; t_cf = 0; t_first_wdf = 0; t_have_first = false;
    return XGM_OK;
}

void XgmSegmentWriter::put_position(uint32_t v) {
    positions.push_back((uint8_t)v); positions.push_back((uint8_t)(v >> 8));
    if (!t_pos16) { positions.push_back((uint8_t)(v >> 16)); positions.push_back((uint8_t)(v >> 24)); }
    ++t_entries; ++n_pos_entries;
}

void XgmSegmentWriter::reserve_like(const XgmSegmentBlob& old) {
    const xgm_seg_header* h = old.header();
    const auto more = [](uint64_t n) { return (size_t)(n + n / 8 + 4096); };
    blk_first.reserve(more(h->n_blocks)); blk_meta.reserve(more(h->n_blocks)); blk_word.reserve(more(h->n_blocks)); blk_pos.reserve(more(h->n_blocks));
    words.reserve(more(h->n_words));
    positions.reserve(more(h->sec_bytes[XGM_S_POSITIONS]));
    str_bytes.reserve(more(h->sec_bytes[XGM_S_STR_BYTES]));
    const size_t T = more(h->n_terms);
    term_df.reserve(T); term_cf.reserve(T); term_wdfub.reserve(T); term_flags.reserve(T);
    term_blk.reserve(T); term_word.reserve(T); term_pos.reserve(T); str_off.reserve(T);
}

int XgmSegmentWriter::copy_blocks(const XgmSegmentBlob& old, uint32_t t, uint64_t b_end, uint64_t cf_of_them, uint32_t first_wdf) {
    const xgm_seg_header* h = old.header();
    if (h->stripe_bits != stripe_bits) return xgm_set_error(XGM_E_INVALID, "copy_blocks: stripe widths differ");
    const uint64_t* tb = old.section<uint64_t>(XGM_S_TERM_BLK);
    const uint64_t* tw = old.section<uint64_t>(XGM_S_TERM_WORD);
    const uint64_t* tp = old.section<uint64_t>(XGM_S_TERM_POS);
    const uint32_t* oflags = old.section<uint32_t>(XGM_S_TERM_FLAGS);
    const uint32_t* bf = old.section<uint32_t>(XGM_S_BLK_FIRST);
    const uint32_t* bm = old.section<uint32_t>(XGM_S_BLK_META);
    const uint32_t* bwd = old.section<uint32_t>(XGM_S_BLK_WORD);
    const uint32_t* bps = old.section<uint32_t>(XGM_S_BLK_POS);
    const uint32_t* ow = old.section<uint32_t>(XGM_S_WORDS);
    const uint8_t* op = old.section<uint8_t>(XGM_S_POSITIONS);
    const uint64_t b0 = tb[t];
    if (b_end <= b0 || t_df) return XGM_OK;                  /* nothing to copy (only ever the first thing added to a term) */
    /* payload words of the copied blocks: [0, end of the last one) of the old term's payload */
    const uint64_t last = b_end - 1;
    const uint32_t lc = XGM_META_COUNT(bm[last]), lg = XGM_META_BWG(bm[last]), lw = XGM_META_BWW(bm[last]);
    const uint64_t wend = (uint64_t)bwd[last] + ((uint64_t)lc * lg + 31) / 32 + ((uint64_t)lc * lw + 31) / 32;
    words.insert(words.end(), ow + tw[t], ow + tw[t] + wend);
    uint64_t n = 0;
    blk_first.insert(blk_first.end(), bf + b0, bf + b_end);
    blk_meta.insert(blk_meta.end(), bm + b0, bm + b_end);
    blk_word.insert(blk_word.end(), bwd + b0, bwd + b_end);
    if (t_pos_ok) blk_pos.insert(blk_pos.end(), bps + b0, bps + b_end);
    else blk_pos.insert(blk_pos.end(), (size_t)(b_end - b0), 0u);
    for (uint64_t b = b0; b < b_end; ++b) n += XGM_META_COUNT(bm[b]);
    if (t_pos_ok) {
        /* positions of the copied postings: the old term's entries [0, e_end) — the entry of the block after the last copied one,
         * or all the term's when every block is copied — at the width this term now has */
        const bool o16 = (oflags[t] & XGM_TF_POS16) != 0;
        const uint64_t total = (tp[t + 1] - tp[t]) / (o16 ? 2u : 4u);      /* (alignment padding follows the term, never inside) */
        uint64_t e_end = b_end < tb[t + 1] ? bps[b_end] : cf_of_them;
        if (e_end > total) e_end = total;
        const uint8_t* src = op + tp[t];
        if (o16 == t_pos16) {
            positions.insert(positions.end(), src, src + e_end * (o16 ? 2u : 4u));
            t_entries += e_end; n_pos_entries += e_end;
        } else {
            for (uint64_t e = 0; e < e_end; ++e)
                put_position(o16 ? (uint32_t)src[2 * e] | ((uint32_t)src[2 * e + 1] << 8)
                                 : (uint32_t)src[4 * e] | ((uint32_t)src[4 * e + 1] << 8) | ((uint32_t)src[4 * e + 2] << 16) | ((uint32_t)src[4 * e + 3] << 24));
        }
    }
    t_df += n; t_cf += cf_of_them;
    t_first_wdf = first_wdf; t_have_first = true;
    return XGM_OK;
}

int XgmSegmentWriter::add_postings(const uint32_t* did, const uint32_t* wdf, uint32_t df, const uint64_t* pos_off, const uint32_t* pos) {
    BitWriter bw(words);
    uint32_t i = 0;
    if (df && !t_have_first) { t_first_wdf = wdf[0]; t_have_first = true; }
    for (uint32_t q = 0; q < df; ++q) t_cf += wdf[q];
    /* cut into blocks: same stripe, <= XGM_BLOCK postings */
    while (i < df) {
        uint32_t stripe = did[i] >> stripe_bits;
        uint32_t n = 1;
        while (n < XGM_BLOCK && i + n < df && (did[i + n] >> stripe_bits) == stripe) ++n;
        uint32_t maxgap = 0, maxwdf = 0;
        for (uint32_t j = 0; j < n; ++j) {
            if (j) maxgap = std::max(maxgap, did[i + j] - did[i + j - 1] - 1);
            maxwdf = std::max(maxwdf, wdf[i + j]);
        }
        uint32_t bwg = xgm_bits_needed(maxgap), bww = xgm_bits_needed(maxwdf);
        uint64_t woff = words.size() - term_word.back();
        uint64_t poff = t_pos_ok ? t_entries : 0;
        if (woff > 0xFFFFFFFFull || poff > 0xFFFFFFFFull) return xgm_set_error(XGM_E_INVALID, "term %zu too large for 32-bit block offsets", term_df.size());
        blk_first.push_back(did[i]);
        blk_meta.push_back(XGM_META(n, bwg, bww));
        blk_word.push_back((uint32_t)woff);
        blk_pos.push_back((uint32_t)poff);
        bw.begin(n, bwg);
        for (uint32_t j = 1; j < n; ++j) bw.put(j, bwg, did[i + j] - did[i + j - 1] - 1);
        bw.begin(n, bww);
        for (uint32_t j = 0; j < n; ++j) bw.put(j, bww, wdf[i + j]);
        if (t_pos_ok) {
            for (uint32_t j = 0; j < n; ++j)
                for (uint64_t q = pos_off[i + j]; q < pos_off[i + j + 1]; ++q) put_position(pos[q]);
        }
        i += n;
    }
    t_df += df;
    return XGM_OK;
}

int XgmSegmentWriter::end_term() {
    if (t_df == 0) return xgm_set_error(XGM_E_INVALID, "term %zu has df 0", term_df.size());
    if (t_df > 0xFFFFFFFFull) return xgm_set_error(XGM_E_INVALID, "term %zu: too many postings", term_df.size());
    const uint64_t cf = t_cf > 0xFFFFFFFFull ? 0xFFFFFFFFull : t_cf;
    term_df.push_back((uint32_t)t_df);
    term_cf.push_back((uint32_t)cf);
    /* GlassPostListTable::get_freqs (glass_postlist.cc:175-189) capped as in
     * GlassDatabase::get_wdf_upper_bound (glass_database.cc:823-830). */
    uint32_t ub = (cf == 0 || t_df == 1) ? (uint32_t)cf : std::max((uint32_t)cf - t_first_wdf, t_first_wdf);
    term_wdfub.push_back(std::min(ub, wdf_ub_db));
    term_flags.push_back(t_pos_ok ? (XGM_TF_POS_OK | (t_pos16 ? XGM_TF_POS16 : 0u)) : 0u);
    n_postings += t_df;
    return XGM_OK;
}

int XgmSegmentWriter::finish(const xgm_raw_postings* raw, uint32_t doclen_lb, uint32_t doclen_ub, XgmSegmentBlob* out, const char* out_path) {
    const uint32_t T = (uint32_t)term_df.size();
    if (n_postings != raw->n_postings) return xgm_set_error(XGM_E_INVALID, "sum of df (%llu) != n_postings (%llu)", (unsigned long long)n_postings, (unsigned long long)raw->n_postings);
    term_blk.push_back(blk_first.size());
    term_word.push_back(words.size());
    term_pos.push_back(positions.size());
    if (has_pos) positions.resize(positions.size() + XGM_POS_PAD, 0);
    str_off.push_back(str_bytes.size());
    const uint64_t n_words = words.size();
    words.resize(n_words + XGM_WORD_PAD, 0u);

    xgm_seg_header h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, XGM_SEG_MAGIC, 8);
    h.version = XGM_SEG_VERSION;
    h.stripe_bits = stripe_bits;
    h.block_size = XGM_BLOCK;
    h.n_terms = T;
    h.lastdocid = raw->lastdocid;
    h.doccount = raw->doccount;
    h.has_positions = has_pos ? 1u : 0u;
    h.doclen_lower_bound = doclen_lb;
    h.doclen_upper_bound = doclen_ub;
    h.wdf_upper_bound = wdf_ub_db;
    h.total_length = raw->total_length;
    h.revision = raw->revision;
    h.n_postings = raw->n_postings;
    h.n_positions = n_pos_entries;
    h.n_blocks = blk_first.size();
    h.n_words = n_words;

    struct Sec { int id; const void* p; uint64_t bytes; };
    const Sec secs[] = {
        {XGM_S_DOCLEN, raw->doclen, ((uint64_t)raw->lastdocid + 1) * 4},
        {XGM_S_TERM_DF, term_df.data(), (uint64_t)T * 4},
        {XGM_S_TERM_CF, term_cf.data(), (uint64_t)T * 4},
        {XGM_S_TERM_WDFUB, term_wdfub.data(), (uint64_t)T * 4},
        {XGM_S_TERM_FLAGS, term_flags.data(), (uint64_t)T * 4},
        {XGM_S_TERM_BLK, term_blk.data(), (uint64_t)(T + 1) * 8},
        {XGM_S_TERM_WORD, term_word.data(), (uint64_t)(T + 1) * 8},
        {XGM_S_TERM_POS, term_pos.data(), (uint64_t)(T + 1) * 8},
        {XGM_S_BLK_FIRST, blk_first.data(), blk_first.size() * 4},
        {XGM_S_BLK_META, blk_meta.data(), blk_meta.size() * 4},
        {XGM_S_BLK_WORD, blk_word.data(), blk_word.size() * 4},
        {XGM_S_BLK_POS, blk_pos.data(), blk_pos.size() * 4},
        {XGM_S_WORDS, words.data(), words.size() * 4},
        {XGM_S_POSITIONS, positions.data(), positions.size()},
        {XGM_S_STR_OFF, str_off.data(), (uint64_t)(T + 1) * 8},
        {XGM_S_STR_BYTES, str_bytes.data(), str_bytes.size()},
    };
    uint64_t off = align_up(sizeof h);
    for (const Sec& s : secs) {
        h.sec_off[s.id] = off;
        h.sec_bytes[s.id] = s.bytes;
        off = align_up(off + s.bytes);
    }
    h.file_bytes = off;
    if (out_path) {
        /* straight to the file, section by section (the same bytes as the assembled blob: zero padding between sections) */
        FILE* f = fopen(out_path, "wb");
        if (!f) return xgm_set_error(XGM_E_IO, "cannot create %s: %s", out_path, strerror(errno));
        static const char zeros[256] = {0};
        uint64_t at = 0;
        bool ok = true;
        auto put = [&](const void* p, uint64_t n) { if (n && fwrite(p, 1, (size_t)n, f) != (size_t)n) ok = false; at += n; };
        auto pad_to = [&](uint64_t to) { while (ok && at < to) put(zeros, std::min<uint64_t>(sizeof zeros, to - at)); };
        put(&h, sizeof h);
        for (const Sec& s : secs) { pad_to(h.sec_off[s.id]); put(s.p, s.bytes); }
        pad_to(off);
        if (fclose(f) != 0 || !ok) return xgm_set_error(XGM_E_IO, "short write on %s", out_path);
        return XGM_OK;
    }
    out->bytes.assign(off, 0);
    memcpy(out->bytes.data(), &h, sizeof h);
    for (const Sec& s : secs)
        if (s.bytes) memcpy(out->bytes.data() + h.sec_off[s.id], s.p, s.bytes);
    return XGM_OK;
}

/* DB-wide bounds the way glass tracks them (reference glass_version.h:252-270): doclen lower bound = smallest non-zero length,
 * wdf upper bound = largest wdf; the backend's own (looser) bounds win when given: they are what the reference's BM25Weight sees */
void xgm_database_bounds(const xgm_raw_postings* raw, uint32_t wdf_max_seen, uint32_t* doclen_lb_out, uint32_t* doclen_ub_out, uint32_t* wdf_ub_out) {
    uint32_t doclen_lb = 0, doclen_ub = 0, wdf_ub_db = wdf_max_seen;
    for (uint64_t d = 1; d <= raw->lastdocid; ++d) {
        uint32_t l = raw->doclen[d];
        if (l && (doclen_lb == 0 || l < doclen_lb)) doclen_lb = l;
        doclen_ub = std::max(doclen_ub, l);
    }
    if (raw->doclen_upper_bound > doclen_ub) doclen_ub = raw->doclen_upper_bound;
    if (raw->doclen_lower_bound && (doclen_lb == 0 || raw->doclen_lower_bound <= doclen_lb)) doclen_lb = raw->doclen_lower_bound;
    if (raw->wdf_upper_bound >= wdf_ub_db && raw->wdf_upper_bound) wdf_ub_db = raw->wdf_upper_bound;
    *doclen_lb_out = doclen_lb; *doclen_ub_out = doclen_ub; *wdf_ub_out = wdf_ub_db;
}

/* which positional form a run of postings allows: every posting has exactly wdf positions (XGM_TF_POS_OK), all of them < 65536 */
void xgm_positional_form(const uint32_t* wdf, uint32_t df, const uint64_t* pos_off, const uint32_t* pos, bool* pos_ok, bool* pos16) {
    bool ok = true;
    for (uint32_t i = 0; ok && i < df; ++i) ok = pos_off[i + 1] - pos_off[i] == wdf[i];
    bool p16 = ok;
    if (ok) for (uint64_t q = pos_off[0]; p16 && q < pos_off[df]; ++q) p16 = pos[q] < 65536u;
    *pos_ok = ok; *pos16 = p16;
}

int xgm_build_segment_blob(const xgm_raw_postings* raw, uint32_t stripe_bits, XgmSegmentBlob* out) {
    if (!raw || !out) return xgm_set_error(XGM_E_INVALID, "null argument");
    const uint32_t T = raw->n_terms;
    const bool has_pos = raw->has_positions && raw->pos_off && (raw->pos || raw->n_positions == 0);
    uint32_t wdf_seen = 0, doclen_lb, doclen_ub, wdf_ub_db;
    for (uint64_t i = 0; i < raw->n_postings; ++i) wdf_seen = std::max(wdf_seen, raw->wdf[i]);
    xgm_database_bounds(raw, wdf_seen, &doclen_lb, &doclen_ub, &wdf_ub_db);
    XgmSegmentWriter w;
    int rc = w.begin(stripe_bits, has_pos, wdf_ub_db);
    if (rc) return rc;
    uint64_t p0 = 0;
    for (uint32_t t = 0; t < T; ++t) {
        const uint32_t df = raw->df[t];
        if (df == 0) return xgm_set_error(XGM_E_INVALID, "term %u has df 0", t);
        if (p0 + df > raw->n_postings) return xgm_set_error(XGM_E_INVALID, "df overruns postings");
        for (uint32_t i = 0; i < df; ++i) {
            if (i && raw->did[p0 + i] <= raw->did[p0 + i - 1]) return xgm_set_error(XGM_E_INVALID, "docids of term %u not ascending", t);
            if (raw->did[p0 + i] == 0 || raw->did[p0 + i] > raw->lastdocid) return xgm_set_error(XGM_E_INVALID, "docid out of range in term %u", t);
        }
        bool pos_ok = false, pos16 = false;
        if (has_pos) xgm_positional_form(raw->wdf + p0, df, raw->pos_off + p0, raw->pos, &pos_ok, &pos16);
        if ((rc = w.begin_term(raw->terms[t], raw->term_len[t], pos_ok, pos16))) return rc;
        if ((rc = w.add_postings(raw->did + p0, raw->wdf + p0, df, has_pos ? raw->pos_off + p0 : nullptr, raw->pos))) return rc;
        if ((rc = w.end_term())) return rc;
        p0 += df;
    }
    if (p0 != raw->n_postings) return xgm_set_error(XGM_E_INVALID, "sum of df (%llu) != n_postings (%llu)", (unsigned long long)p0, (unsigned long long)raw->n_postings);
    return w.finish(raw, doclen_lb, doclen_ub, out);
}

int xgm_read_raw_file(const char* path, std::vector<uint8_t>* storage, std::vector<const char*>* term_ptrs,
                      std::vector<uint32_t>* term_lens, xgm_raw_postings* raw) {
    FILE* f = fopen(path, "rb");
    if (!f) return xgm_set_error(XGM_E_IO, "cannot open %s: %s", path, strerror(errno));
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    storage->resize((size_t)sz);
    size_t got = fread(storage->data(), 1, (size_t)sz, f);
    fclose(f);
    if (got != (size_t)sz) return xgm_set_error(XGM_E_IO, "short read on %s", path);
    const uint8_t* p = storage->data();
    const uint8_t* e = p + sz;
    auto pad8 = [](uint64_t n) { return (n + 7) / 8 * 8; };
    if (sz < 64 || memcmp(p, "XGMRAW1", 8) != 0) return xgm_set_error(XGM_E_INVALID, "%s is not an XGMRAW1 file", path);
    memset(raw, 0, sizeof *raw);
    uint64_t str_total;
    memcpy(&raw->n_terms, p + 8, 4);
    memcpy(&raw->lastdocid, p + 12, 4);
    memcpy(&raw->doccount, p + 16, 4);
    memcpy(&raw->has_positions, p + 20, 4);
    memcpy(&raw->total_length, p + 24, 8);
    memcpy(&raw->n_postings, p + 32, 8);
    memcpy(&raw->n_positions, p + 40, 8);
    memcpy(&raw->revision, p + 48, 8);
    memcpy(&str_total, p + 56, 8);
    p += 64;
    uint64_t need = pad8(((uint64_t)raw->lastdocid + 1) * 4) + pad8((uint64_t)raw->n_terms * 4) * 2 +
                    pad8(raw->n_postings * 4) * 2 + str_total +
                    (raw->has_positions ? (raw->n_postings + 1) * 8 + pad8(raw->n_positions * 4) : 0);
    if ((uint64_t)(e - p) < need) return xgm_set_error(XGM_E_INVALID, "raw file %s truncated", path);
    raw->doclen = reinterpret_cast<const uint32_t*>(p); p += pad8(((uint64_t)raw->lastdocid + 1) * 4);
    raw->df = reinterpret_cast<const uint32_t*>(p); p += pad8((uint64_t)raw->n_terms * 4);
    raw->did = reinterpret_cast<const uint32_t*>(p); p += pad8(raw->n_postings * 4);
    raw->wdf = reinterpret_cast<const uint32_t*>(p); p += pad8(raw->n_postings * 4);
    if (raw->has_positions) {
        raw->pos_off = reinterpret_cast<const uint64_t*>(p); p += (raw->n_postings + 1) * 8;
        raw->pos = reinterpret_cast<const uint32_t*>(p); p += pad8(raw->n_positions * 4);
    }
    const uint32_t* lens = reinterpret_cast<const uint32_t*>(p); p += pad8((uint64_t)raw->n_terms * 4);
    term_ptrs->resize(raw->n_terms);
    term_lens->assign(lens, lens + raw->n_terms);
    uint64_t o = 0;
    for (uint32_t t = 0; t < raw->n_terms; ++t) {
        (*term_ptrs)[t] = reinterpret_cast<const char*>(p + o);
        o += lens[t];
    }
    if (o != str_total) return xgm_set_error(XGM_E_INVALID, "raw file %s: term bytes mismatch", path);
    raw->terms = term_ptrs->data();
    raw->term_len = term_lens->data();
    return XGM_OK;
}

int xgm_write_blob(const XgmSegmentBlob& blob, const char* path) {
    FILE* f = fopen(path, "wb");
    if (!f) return xgm_set_error(XGM_E_IO, "cannot create %s: %s", path, strerror(errno));
    size_t w = fwrite(blob.bytes.data(), 1, blob.bytes.size(), f);
    int rc = fclose(f);
    if (w != blob.bytes.size() || rc != 0) return xgm_set_error(XGM_E_IO, "short write on %s", path);
    return XGM_OK;
}

extern "C" int xgm_segment_build(const xgm_raw_postings* raw, uint32_t stripe_bits, const char* out_path) {
    XgmSegmentBlob blob;
    int rc = xgm_build_segment_blob(raw, stripe_bits, &blob);
    if (rc) return rc;
    return xgm_write_blob(blob, out_path);
}

extern "C" int xgm_segment_build_from_file(const char* raw_path, uint32_t stripe_bits, const char* out_path) {
    std::vector<uint8_t> storage;
    std::vector<const char*> tp;
    std::vector<uint32_t> tl;
    xgm_raw_postings raw;
    int rc = xgm_read_raw_file(raw_path, &storage, &tp, &tl, &raw);
    if (rc) return rc;
    return xgm_segment_build(&raw, stripe_bits, out_path);
}

static int read_whole(const char* path, std::vector<uint8_t>* out) {
    FILE* f = fopen(path, "rb");
    if (!f) return xgm_set_error(XGM_E_IO, "cannot open %s: %s", path, strerror(errno));
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    out->resize((size_t)sz);
    size_t got = fread(out->data(), 1, (size_t)sz, f);
    fclose(f);
    if (got != (size_t)sz) return xgm_set_error(XGM_E_IO, "short read on %s", path);
    return XGM_OK;
}

XgmSegmentBlob::~XgmSegmentBlob() { if (map) munmap(const_cast<uint8_t*>(map), map_size); }

/* xgm_read_segment_file without the copy: the file mapped read-only (the refresh reads most of an old segment once, to copy it) */
int xgm_map_segment_file(const char* path, XgmSegmentBlob* blob) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return xgm_set_error(XGM_E_IO, "cannot open %s: %s", path, strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return xgm_set_error(XGM_E_IO, "cannot stat %s", path); }
    if ((size_t)st.st_size < sizeof(xgm_seg_header)) { close(fd); return xgm_set_error(XGM_E_INVALID, "%s too small", path); }
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return xgm_set_error(XGM_E_IO, "cannot map %s: %s", path, strerror(errno));
    blob->map = (const uint8_t*)m;
    blob->map_size = (size_t)st.st_size;
    int rc = xgm_validate_header(blob->header(), blob->size());
    if (rc) return rc;
    return xgm_validate_blob(*blob);
}

int xgm_read_segment_file(const char* path, XgmSegmentBlob* blob) {
    int rc = read_whole(path, &blob->bytes);
    if (rc) return rc;
    if (blob->bytes.size() < sizeof(xgm_seg_header)) return xgm_set_error(XGM_E_INVALID, "%s too small", path);
    if ((rc = xgm_validate_header(blob->header(), blob->bytes.size()))) return rc;
    return xgm_validate_blob(*blob);
}

static uint32_t get_bits(const uint32_t* w, uint32_t i, uint32_t bw) {
    if (!bw) return 0;
    uint64_t bit = (uint64_t)i * bw;
    const uint32_t* p = w + (bit >> 5);
    uint64_t v = (uint64_t)p[0] | ((uint64_t)p[1] << 32);
    return (uint32_t)((v >> (bit & 31)) & ((bw == 32) ? 0xFFFFFFFFull : ((1ull << bw) - 1)));
}

extern "C" int64_t xgm_segment_decode_term(const char* segment_path, const char* term, size_t len, uint32_t* did,
                                           uint32_t* wdf, uint64_t cap) {
    XgmSegmentBlob blob;
    int rc = xgm_read_segment_file(segment_path, &blob);
    if (rc) return rc;
    const xgm_seg_header* h = blob.header();
    const uint64_t* so = blob.section<uint64_t>(XGM_S_STR_OFF);
    const char* sb = blob.section<char>(XGM_S_STR_BYTES);
    uint32_t lo = 0, hi = h->n_terms;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        size_t ml = (size_t)(so[mid + 1] - so[mid]);
        int c = memcmp(sb + so[mid], term, std::min(ml, len));
        if (c < 0 || (c == 0 && ml < len)) lo = mid + 1; else hi = mid;
    }
    if (lo == h->n_terms || (size_t)(so[lo + 1] - so[lo]) != len || memcmp(sb + so[lo], term, len) != 0) return 0;
    const uint64_t* tb = blob.section<uint64_t>(XGM_S_TERM_BLK);
    const uint64_t* tw = blob.section<uint64_t>(XGM_S_TERM_WORD);
    const uint32_t* bf = blob.section<uint32_t>(XGM_S_BLK_FIRST);
    const uint32_t* bm = blob.section<uint32_t>(XGM_S_BLK_META);
    const uint32_t* bwd = blob.section<uint32_t>(XGM_S_BLK_WORD);
    const uint32_t* words = blob.section<uint32_t>(XGM_S_WORDS);
    uint64_t n = 0;
    for (uint64_t b = tb[lo]; b < tb[lo + 1]; ++b) {
        uint32_t cnt = XGM_META_COUNT(bm[b]), bwg = XGM_META_BWG(bm[b]), bww = XGM_META_BWW(bm[b]);
        const uint32_t* gw = words + tw[lo] + bwd[b];
        const uint32_t* ww = gw + ((uint64_t)cnt * bwg + 31) / 32;
        uint32_t d = bf[b];
        for (uint32_t j = 0; j < cnt; ++j) {
            if (j) d += get_bits(gw, j, bwg) + 1;
            if (n < cap) { did[n] = d; wdf[n] = get_bits(ww, j, bww); }
            ++n;
        }
    }
    return (int64_t)n;
}
