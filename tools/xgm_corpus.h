/* Deterministic synthetic corpus shared by the CPU tools (oracle, reference driver) and the GPU
 * segment builder.  Plain C99 / HIP-compatible: every function is `static inline` integer code,
 * so the CPU and the GPU derive bit-identical documents from (seed, global docid, position).
 *
 * Corpus model: vocabulary of V terms named "t<rank>",
 * document g (global docid, 1-based) has length L_g ~ UniformInt[len_lo, len_hi] and its token at
 * position p (1-based) is drawn i.i.d. Zipf(s = 1) over ranks 1..V by inverse CDF.  The reference
 * plan uses std::mt19937_64; we use a counter-based hash instead so any (doc, pos) can be produced
 * independently (required to generate 10^9 tokens in parallel on the GPU).
 *
 * This is tooling for tests/bench, not part of the product path.
 */
#ifndef XGM_CORPUS_H
#define XGM_CORPUS_H
#include <stdint.h>

#ifdef __HIPCC__
#define XGM_HD __host__ __device__
#else
#define XGM_HD
#endif

typedef struct {
    uint64_t seed;
    uint32_t vocab;    /* V */
    uint32_t len_lo;   /* inclusive */
    uint32_t len_hi;   /* inclusive, < 256 so a position fits in 8 bits */
} xgm_corpus_params;

XGM_HD static inline uint64_t xgm_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

XGM_HD static inline uint64_t xgm_hash3(uint64_t seed, uint64_t a, uint64_t b) {
    return xgm_mix64(xgm_mix64(seed ^ (a * 0xD6E8FEB86659FD93ull)) ^ (b * 0xA0761D6478BD642Full));
}

/* Document length of global doc g. */
XGM_HD static inline uint32_t xgm_doc_len(const xgm_corpus_params* p, uint64_t g) {
    uint64_t h = xgm_hash3(p->seed, g, 0xFFFFFFFFull);
    return p->len_lo + (uint32_t)(h % (uint64_t)(p->len_hi - p->len_lo + 1));
}

/* thresholds[r-1] = floor(2^64 * H_r / H_V) (last entry forced to 2^64-1); token rank is the
 * first r with u < thresholds[r-1], or V when u is larger than all (only u == 2^64-1). */
XGM_HD static inline uint32_t xgm_zipf_rank(const uint64_t* thresholds, uint32_t vocab, uint64_t u) {
    uint32_t lo = 0, hi = vocab - 1;          /* answer index in [lo, hi] */
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (u < thresholds[mid]) hi = mid; else lo = mid + 1;
    }
    return lo + 1;
}

/* Token (term rank, 1-based) of global doc g at 1-based position pos. */
XGM_HD static inline uint32_t xgm_token(const xgm_corpus_params* p, const uint64_t* thresholds,
                                        uint64_t g, uint32_t pos) {
    return xgm_zipf_rank(thresholds, p->vocab, xgm_hash3(p->seed, g, pos));
}

/* Value slots of global doc g (widening row (f).3: sort by value / collapse), as short byte strings whose bytewise
 * order is the intended order: slot 0 = a category "cNN" (37 of them; about one document in 29 has none), slot 1 =
 * a six-digit number (ties are rare), slot 2 = one digit 0..4 (many ties: the second sort criterion decides).
 * Writes at most 7 bytes + NUL to buf; returns the length (0 = the document has no value in that slot). */
static inline uint32_t xgm_doc_value(const xgm_corpus_params* p, uint64_t g, uint32_t slot, char* buf) {
    const uint64_t h = xgm_hash3(p->seed, g, 0xFFFFFFF0ull + slot);
    uint32_t n = 0;
    if (slot == 0) {
        if (h % 29u == 0) { buf[0] = 0; return 0; }
        const uint32_t c = (uint32_t)((h >> 8) % 37u);
        buf[n++] = 'c'; buf[n++] = (char)('0' + c / 10u); buf[n++] = (char)('0' + c % 10u);
    } else if (slot == 1) {
        uint32_t v = (uint32_t)(h % 1000000u);
        for (int i = 5; i >= 0; --i) { buf[i] = (char)('0' + v % 10u); v /= 10u; }
        n = 6;
    } else {
        buf[n++] = (char)('0' + (uint32_t)(h % 5u));
    }
    buf[n] = 0;
    return n;
}

/* Host-only: fill thresholds[0..V-1].  Sequential double summation of 1/r: only IEEE add and
 * divide, so every host computes the same table. */
static inline void xgm_zipf_thresholds(uint32_t vocab, uint64_t* thresholds) {
    double hv = 0.0;
    for (uint32_t r = 1; r <= vocab; ++r) hv += 1.0 / (double)r;
    double h = 0.0;
    for (uint32_t r = 1; r <= vocab; ++r) {
        h += 1.0 / (double)r;
        double f = h / hv;                       /* in (0, 1] */
        if (r == vocab || f >= 1.0) { thresholds[r - 1] = 0xFFFFFFFFFFFFFFFFull; continue; }
        /* f * 2^64 with f < 1: scale in two exact power-of-two steps to stay inside uint64. */
        double hi = f * 4294967296.0;            /* f * 2^32 */
        uint64_t hi_i = (uint64_t)hi;
        double lo = (hi - (double)hi_i) * 4294967296.0;
        thresholds[r - 1] = (hi_i << 32) | (uint64_t)lo;
    }
}

#endif /* XGM_CORPUS_H */
