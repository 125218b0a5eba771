/* Native reader of the reference's on-disk shard format ("glass"): walks the postlist and position
 * B-trees of a committed glass database directly — no Xapian in the loop — and hands the postings to
 * the segment builder.  This is the exporter of SURVEY.md §8(f).1: the iterator-based export
 * (Database::allterms_begin / postlist_begin / positionlist_begin) pays a B-tree
 * cursor and a virtual call per posting; reading the leaf blocks in key order is a sequential scan.
 *
 * Formats restated from the reference (paths under /root/reference/src/xapian/):
 *   version file  <dir>/iamglass: magic, format version, uuid, revision, one RootInfo per table,
 *                 database statistics                          backends/glass/glass_version.cc:100-235, 423-468
 *   B-tree block  REVISION u32 | LEVEL u8 | MAX_FREE u16 | TOTAL_FREE u16 | DIR_END u16 | directory of
 *                 u16 item offsets from byte 11; all integers big-endian  backends/glass/glass_table.h:62-124
 *   leaf item     I2 (size - 3 in the low 13 bits; 0x80 compressed, 0x40 last, 0x20 first component) |
 *                 K1 key length | key | X2 component number (absent on the first) | tag piece   :138-215
 *   branch item   u32 child block | K1 | key | X2                                               :268-330
 *   postlist keys / chunk tags   common/pack.h:183-290, 520-594; backends/glass/glass_postlist.cc:100-150, 677-695
 *   position keys / tags         backends/glass/glass_positionlist.h:39-44, glass_positionlist.cc:36-52, 96-133;
 *                                interpolative code common/bitstream.cc:93-255
 * The postlist and position tables are never compressed (compress_min 0, glass_version.cc:398-405).
 */
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "xgm_internal.h"

namespace {

constexpr int kDirStart = 11;
constexpr unsigned kTablePostlist = 0, kTablePosition = 3, kTableCount = 6;

inline uint32_t be16(const uint8_t* p) { return ((uint32_t)p[0] << 8) | p[1]; }
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

/* unpack_uint: 7 bits per byte, least significant group first, 0x80 = more follow (common/pack.h:325-389) */
bool get_varint(const uint8_t** p, const uint8_t* end, uint64_t* out) {
    uint64_t v = 0;
    unsigned shift = 0;
    while (*p < end) {
        const uint8_t b = *(*p)++;
        if (shift < 64) v |= (uint64_t)(b & 0x7F) << shift;
        shift += 7;
        if (!(b & 0x80)) { *out = v; return true; }
    }
    return false;
}

/* unpack_uint_preserving_sort (common/pack.h:232-283) */
bool get_sortable_uint(const uint8_t** p, const uint8_t* end, uint64_t* out) {
    if (*p >= end) return false;
    uint8_t lb = *(*p)++;
    if (lb < 0x80) {
        if (*p >= end) return false;
        *out = ((uint64_t)lb << 8) | *(*p)++;
        return true;
    }
    if (lb == 0xFF) return false;
    size_t len = 2;
    for (uint8_t m = 0x40; lb & m; m >>= 1) ++len;
    if ((size_t)(end - *p) < len || len > 8) return false;
    const unsigned mask = 0xFFu << (9 - len);
    uint64_t r = lb & ~mask & 0xFFu;
    for (size_t i = 0; i < len; ++i) r = (r << 8) | *(*p)++;
    *out = r;
    return true;
}

struct RootInfo {
    uint64_t root = 0, num_entries = 0, blocksize = 0;
    unsigned level = 0;
    bool fake = true;
};

struct GlassVersion {
    uint64_t revision = 0, doccount = 0, last_docid = 0, total_doclen = 0;
    uint64_t doclen_ubound_delta = 0;
    uint64_t doclen_lbound = 0, wdf_ubound = 0;      /* glass's own, never tightened on delete / replace (glass_version.h:252-270) */
    RootInfo root[kTableCount];
};

int read_version(const std::string& dir, GlassVersion* v) {
    const std::string path = dir + "/iamglass";
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return xgm_set_error(XGM_E_IO, "cannot open %s: %s", path.c_str(), strerror(errno));
    uint8_t buf[1024];
    const size_t n = fread(buf, 1, sizeof buf, f);
    fclose(f);
    static const char magic[] = "\x0f\x0dXapian Glass";
    if (n < 33 || memcmp(buf, magic, 14) != 0) return xgm_set_error(XGM_E_INVALID, "%s: not a glass version file", path.c_str());
    const unsigned version = ((unsigned)buf[14] << 8) | buf[15];
    const unsigned want = ((2016u - 2014u) << 9) | (3u << 5) | 14u;            /* DATE_TO_VERSION(2016,03,14), glass_version.cc:53-60 */
    if (version != want) return xgm_set_error(XGM_E_INVALID, "%s: glass format version %u, expected %u", path.c_str(), version, want);
    const uint8_t* p = buf + 16 + 16;                                            /* magic + version, uuid */
    const uint8_t* end = buf + n;
    if (!get_varint(&p, end, &v->revision)) return xgm_set_error(XGM_E_INVALID, "%s: truncated", path.c_str());
    for (unsigned t = 0; t < kTableCount; ++t) {
        uint64_t val, bs, cmin, fl_len;
        RootInfo& r = v->root[t];
        if (!get_varint(&p, end, &r.root) || !get_varint(&p, end, &val) || !get_varint(&p, end, &r.num_entries) ||
            !get_varint(&p, end, &bs) || !get_varint(&p, end, &cmin) || !get_varint(&p, end, &fl_len) || (uint64_t)(end - p) < fl_len)
            return xgm_set_error(XGM_E_INVALID, "%s: root info of table %u is truncated", path.c_str(), t);
        p += fl_len;
        r.level = (unsigned)(val >> 2);
        r.fake = (val & 1) != 0;
        r.blocksize = bs << 11;
    }
    uint64_t ld_minus, skip;
    if (p == end) return XGM_OK;                                                 /* empty database: no statistics */
    if (!get_varint(&p, end, &v->doccount) || !get_varint(&p, end, &ld_minus) || !get_varint(&p, end, &v->doclen_lbound) ||
        !get_varint(&p, end, &v->wdf_ubound) || !get_varint(&p, end, &v->doclen_ubound_delta) /* doclen_ubound - wdf_ubound */ ||
        !get_varint(&p, end, &skip) /* oldest_changeset */ || !get_varint(&p, end, &v->total_doclen))
        return xgm_set_error(XGM_E_INVALID, "%s: database statistics are truncated", path.c_str());
    v->last_docid = ld_minus + v->doccount;
    return XGM_OK;
}

/* One table file, mapped read-only; walk() visits every (key, tag) of the committed tree in key order. */
class Table {
  public:
    ~Table() { if (map_ && map_ != MAP_FAILED) munmap((void*)map_, size_); if (fd_ >= 0) close(fd_); }

    int open(const std::string& path, const RootInfo& root) {
        root_ = root;
        path_ = path;
        if (root.fake || root.num_entries == 0) return XGM_OK;                   /* nothing committed in this table */
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) return xgm_set_error(XGM_E_IO, "cannot open %s: %s", path.c_str(), strerror(errno));
        struct stat st;
        if (fstat(fd_, &st) != 0) return xgm_set_error(XGM_E_IO, "cannot stat %s", path.c_str());
        size_ = (size_t)st.st_size;
        if (root.blocksize < 2048 || root.blocksize > 65536 || (root.root + 1) * root.blocksize > size_)
            return xgm_set_error(XGM_E_INVALID, "%s: root block %llu outside the file", path.c_str(), (unsigned long long)root.root);
        map_ = (const uint8_t*)mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (map_ == MAP_FAILED) return xgm_set_error(XGM_E_IO, "cannot map %s: %s", path.c_str(), strerror(errno));
        madvise((void*)map_, size_, MADV_SEQUENTIAL);
        return XGM_OK;
    }

    bool empty() const { return map_ == nullptr; }

    template <class F>
    int walk(F&& emit) {
        auto no4 = [](const uint8_t*, uint32_t, const uint8_t*, uint32_t) { return false; };
        return walk(emit, no4, no4, [](const uint8_t*, uint32_t) { return false; });
    }

    /* ... passing over what the caller does not want without assembling it — or without touching it:
     *   skip_child(separator of a leaf block, separator of the next one) → true: the leaf block is not read at all.  Every key k
     *     of the block satisfies separator <= k < next separator bytewise (separators are prefixes of a block's first key cut
     *     after the first byte that differs from the block before, glass_table.cc:645-685);
     *   skip_block(first key, last key of a leaf block) → true drops the block (asked only when it holds complete items);
     *   skip_item(key) → true drops one item. */
    template <class F, class SC, class SB, class SI>
    int walk(F&& emit, SC&& skip_child, SB&& skip_block, SI&& skip_item) {
        if (empty()) return XGM_OK;
        key_.clear(); tag_.clear(); in_item_ = false; skipping_ = false; after_skip_ = false;
        int rc = visit(root_.root, root_.level, emit, skip_child, skip_block, skip_item);
        if (rc == XGM_OK && (in_item_ || skipping_)) rc = xgm_set_error(XGM_E_INVALID, "%s: last tag is incomplete", path_.c_str());
        return rc;
    }

  private:
    template <class F, class SC, class SB, class SI>
    int visit(uint64_t block, unsigned level, F& emit, SC& skip_child, SB& skip_block, SI& skip_item) {
        if ((block + 1) * root_.blocksize > size_) return xgm_set_error(XGM_E_INVALID, "%s: block %llu outside the file", path_.c_str(), (unsigned long long)block);
        const uint8_t* b = map_ + block * root_.blocksize;
        if (b[4] != level) return xgm_set_error(XGM_E_INVALID, "%s: block %llu has level %u, expected %u", path_.c_str(), (unsigned long long)block, b[4], level);
        const uint32_t dir_end = be16(b + 9);
        if (dir_end < (uint32_t)kDirStart || dir_end > root_.blocksize) return xgm_set_error(XGM_E_INVALID, "%s: bad directory in block %llu", path_.c_str(), (unsigned long long)block);
        if (level == 0 && !in_item_ && !skipping_ && !after_skip_ && dir_end >= (uint32_t)kDirStart + 4u) {
            const uint32_t off_a = be16(b + kDirStart), off_z = be16(b + dir_end - 2);
            if (off_a + 3 <= root_.blocksize && off_z + 3 <= root_.blocksize) {
                const uint8_t* a = b + off_a;
                const uint8_t* z = b + off_z;
                if ((a[0] & 0x20) && (z[0] & 0x40) && a[2] && off_a + 3u + a[2] <= root_.blocksize && off_z + 3u + z[2] <= root_.blocksize &&
                    skip_block(a + 3, (uint32_t)a[2], z + 3, (uint32_t)z[2])) return XGM_OK;
            }
        }
        for (uint32_t c = kDirStart; c < dir_end; c += 2) {
            const uint32_t off = be16(b + c);
            if (off + 3 > root_.blocksize) return xgm_set_error(XGM_E_INVALID, "%s: bad item offset in block %llu", path_.c_str(), (unsigned long long)block);
            const uint8_t* it = b + off;
            if (level > 0) {
                if (level == 1 && !in_item_ && !skipping_ && c + 2 < dir_end) {
                    /* branch item: block number (4), key length (1), key, component count (2) — glass_table.h:301-331 */
                    const uint32_t off2 = be16(b + c + 2);
                    if (off + 5u + it[4] <= root_.blocksize && off2 + 5u <= root_.blocksize && off2 + 5u + b[off2 + 4] <= root_.blocksize &&
                        skip_child(it + 5, (uint32_t)it[4], b + off2 + 5, (uint32_t)b[off2 + 4])) { after_skip_ = true; continue; }
                }
                int rc = visit(be32(it), level - 1, emit, skip_child, skip_block, skip_item);      /* BItem::block_given_by */
                if (rc) return rc;
                continue;
            }
            const uint32_t size = (be16(it) & 0x1FFFu) + 3u;                     /* LeafItem::size */
            const bool compressed = it[0] & 0x80, last = it[0] & 0x40, first = it[0] & 0x20;
            const uint32_t klen = it[2];
            uint32_t cd = 3u + klen + (first ? 0u : 2u);
            if (off + size > root_.blocksize || cd > size) return xgm_set_error(XGM_E_INVALID, "%s: bad item in block %llu", path_.c_str(), (unsigned long long)block);
            if (compressed) return xgm_set_error(XGM_E_INVALID, "%s: compressed tag (not expected in this table)", path_.c_str());
            if (skipping_) {                          /* further components of an item the caller passed over */
                if (first) return xgm_set_error(XGM_E_INVALID, "%s: tag components out of sequence", path_.c_str());
                if (last) skipping_ = false;
                continue;
            }
            if (after_skip_) {                        /* the block before was not read: it may have ended inside an (unwanted) item */
                after_skip_ = false;
                if (!first && !in_item_) { skipping_ = !last; continue; }
            }
            if (first) {
                if (in_item_) return xgm_set_error(XGM_E_INVALID, "%s: tag components out of sequence", path_.c_str());
                if (klen && skip_item(it + 3, klen)) { skipping_ = !last; continue; }
                key_.assign((const char*)it + 3, klen);
                tag_.clear();
                in_item_ = true;
            } else if (!in_item_ || key_.size() != klen || memcmp(key_.data(), it + 3, klen) != 0) {
                return xgm_set_error(XGM_E_INVALID, "%s: tag components out of sequence", path_.c_str());
            }
            tag_.append((const char*)it + cd, size - cd);
            if (last) {
                in_item_ = false;
                if (key_.empty()) continue;          /* the null-key item that opens every tree (glass_table.cc: "dummy" first item) */
                int rc = emit(key_, tag_);
                if (rc) return rc;
            }
        }
        return XGM_OK;
    }

    RootInfo root_;
    std::string path_;
    int fd_ = -1;
    const uint8_t* map_ = nullptr;
    size_t size_ = 0;
    std::string key_, tag_;
    bool in_item_ = false, skipping_ = false, after_skip_ = false;
};

/* unpack_string_preserving_sort on a key prefix: term bytes up to the unescaped \0 (or the end of the key);
 * *rest points after the terminator, or is NULL when the key ends with the term (first chunk of a list) */
void split_term_key(const std::string& key, std::string* term, const uint8_t** rest) {
    term->clear();
    const uint8_t* p = (const uint8_t*)key.data();
    const uint8_t* end = p + key.size();
    *rest = nullptr;
    while (p < end) {
        const uint8_t ch = *p++;
        if (ch == 0) {
            if (p < end && *p == 0xFF) { ++p; term->push_back('\0'); continue; }
            *rest = p;
            return;
        }
        term->push_back((char)ch);
    }
}

/* BitReader::read_bits / decode / decode_interpolative (common/bitstream.cc:174-255), recursive form */
struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0;
    int n_bits = 0;
    bool ok = true;

    uint64_t read_bits(int count) {
        if (count > 32) { const uint64_t lo = read_bits(32); return lo | (read_bits(count - 32) << 32); }
        while (n_bits < count) {
            if (p >= end) { ok = false; return 0; }
            acc |= (uint64_t)*p++ << n_bits;
            n_bits += 8;
        }
        const uint64_t r = acc & ((1ull << count) - 1ull);
        acc >>= count;
        n_bits -= count;
        return r;
    }
    uint32_t decode(uint32_t outof) {
        int bits = 0;
        for (uint32_t m = outof - 1u; m; m >>= 1) ++bits;                        /* highest_order_bit(outof - 1) */
        const uint32_t spare = (bits >= 32 ? 0u : (1u << bits)) - outof;
        const uint32_t mid_start = (outof - spare) / 2u;
        uint32_t pos;
        if (spare) {
            pos = (uint32_t)read_bits(bits - 1);
            if (pos < mid_start && read_bits(1)) pos += mid_start + spare;
        } else {
            pos = (uint32_t)read_bits(bits);
        }
        return pos;
    }
    void interpolative(uint32_t* pos, int j, int k) {
        while (j + 1 < k && ok) {
            const int mid = j + (k - j) / 2;
            const uint32_t outof = pos[k] - pos[j] - (uint32_t)(k - j) + 1u;
            pos[mid] = decode(outof) + pos[j] + (uint32_t)(mid - j);
            interpolative(pos, j, mid);
            j = mid;
        }
    }
};

struct Export {
    GlassVersion ver;
    std::vector<std::string> terms;
    std::vector<uint32_t> df, did, wdf, doclen, pos;
    std::vector<uint64_t> pos_off;
    bool has_positions = false;
    uint64_t doccount = 0, total_length = 0;
};

/* a chunk body: bool is_last, varint(last - first), wdf of the first entry, then (did increase - 1, wdf) pairs */
template <class F>
int read_chunk_body(const uint8_t* p, const uint8_t* end, uint64_t first_did, F&& entry, uint64_t min_did = 0) {
    if (p >= end || (uint8_t)(*p - '0') > 1) return xgm_set_error(XGM_E_INVALID, "glass postlist chunk: bad header");
    ++p;
    uint64_t span, w;
    if (!get_varint(&p, end, &span) || !get_varint(&p, end, &w)) return xgm_set_error(XGM_E_INVALID, "glass postlist chunk: truncated header");
    if (first_did + span < min_did) return XGM_OK;                 /* the whole chunk lies below the floor: its header says so, nothing is decoded */
    uint64_t d = first_did;
    entry(d, w);
    while (p < end) {
        uint64_t inc;
        if (!get_varint(&p, end, &inc) || !get_varint(&p, end, &w)) return xgm_set_error(XGM_E_INVALID, "glass postlist chunk: truncated entry");
        d += inc + 1;
        entry(d, w);
    }
    if (d != first_did + span) return xgm_set_error(XGM_E_INVALID, "glass postlist chunk: last docid mismatch");
    return XGM_OK;
}

/* min_did > 0: postings and positions of documents below it are skipped (whole chunks by their header, position lists by
 * their key) — the incremental refresh takes those from the previous segment.  Document lengths are always read in full. */
int read_glass(const char* glass_dir, Export* ex, uint64_t min_did_all = 0, const std::set<std::string>* full_terms = nullptr) {
    const std::string dir(glass_dir);
    uint64_t min_did = min_did_all;                 /* of the term at hand: 0 for the terms wanted in full */
    int rc = read_version(dir, &ex->ver);
    if (rc) return rc;
    if (ex->ver.last_docid > 0xFFFFFFFEull) return xgm_set_error(XGM_E_INVALID, "docids beyond 32 bits");
    ex->doclen.assign((size_t)ex->ver.last_docid + 1, 0);

    Table post;
    if ((rc = post.open(dir + "/postlist.glass", ex->ver.root[kTablePostlist]))) return rc;
    std::string term;
    bool have_term = false;
    rc = post.walk([&](const std::string& key, const std::string& tag) -> int {
        const uint8_t* p = (const uint8_t*)tag.data();
        const uint8_t* end = p + tag.size();
        if (key.size() >= 2 && key[0] == '\0') {
            const uint8_t k1 = (uint8_t)key[1];
            if (k1 == 0xE0) {                                                   /* document lengths: wdf field = length */
                uint64_t first;
                if (key.size() == 2) {
                    uint64_t n, cf;
                    if (!get_varint(&p, end, &n) || !get_varint(&p, end, &cf) || !get_varint(&p, end, &first)) return xgm_set_error(XGM_E_INVALID, "doclen list: truncated");
                    ++first;
                } else {
                    const uint8_t* kp = (const uint8_t*)key.data() + 2;
                    if (!get_sortable_uint(&kp, (const uint8_t*)key.data() + key.size(), &first)) return xgm_set_error(XGM_E_INVALID, "doclen list: bad chunk key");
                }
                return read_chunk_body(p, end, first, [&](uint64_t d, uint64_t len) {
                    if (d < ex->doclen.size()) ex->doclen[d] = (uint32_t)len;
                    ++ex->doccount;
                    ex->total_length += len;
                });
            }
            if (k1 == 0xC0 || k1 == 0xD0 || k1 == 0xD8) return XGM_OK;         /* user metadata, value statistics, value chunks */
        }
        std::string t;
        const uint8_t* rest;
        split_term_key(key, &t, &rest);
        uint64_t first;
        if (!rest) {                                                            /* first chunk of a term */
            uint64_t n, cf;
            if (!get_varint(&p, end, &n) || !get_varint(&p, end, &cf) || !get_varint(&p, end, &first)) return xgm_set_error(XGM_E_INVALID, "postlist: truncated first chunk");
            ++first;
            ex->terms.push_back(t);
            ex->df.push_back(0);
            term = t;
            have_term = true;
            min_did = (full_terms && full_terms->count(t)) ? 0 : min_did_all;
        } else {
            if (!have_term || t != term) return xgm_set_error(XGM_E_INVALID, "postlist: continuation chunk without a first chunk");
            if (!get_sortable_uint(&rest, (const uint8_t*)key.data() + key.size(), &first)) return xgm_set_error(XGM_E_INVALID, "postlist: bad chunk key");
        }
        return read_chunk_body(p, end, first, [&](uint64_t d, uint64_t w) {
            if (d < min_did) return;
            ex->did.push_back((uint32_t)d);
            ex->wdf.push_back((uint32_t)w);
            ++ex->df.back();
        }, min_did);
    });
    if (rc) return rc;

    Table posn;
    if ((rc = posn.open(dir + "/position.glass", ex->ver.root[kTablePosition]))) return rc;
    ex->has_positions = !posn.empty();
    if (ex->has_positions) {
        /* the table is ordered by (term, docid) like the postings: merge in lockstep */
        ex->pos_off.assign(1, 0);
        std::string pos_term;                /* (floor) the term of the last entry below it and whether it is wanted in full */
        bool pos_term_full = false;
        size_t ti = 0;                       /* current term of the posting cursor */
        uint64_t pi = 0, t_end = ex->terms.empty() ? 0 : ex->df[0];             /* posting ordinal, end of term ti */
        std::vector<uint32_t> tmp;
        /* under a floor most of the table is not wanted: entries are judged on their raw key (no strings built), and a leaf block
         * that begins and ends in the same term with its last docid below the floor is not looked into at all */
        auto raw_term_end = [](const uint8_t* k, uint32_t n) -> uint32_t {       /* offset of the terminator of the escaped term, n if none */
            for (uint32_t i = 0; i < n; ++i) if (k[i] == 0) { if (i + 1 < n && k[i + 1] == 0xFF) { ++i; continue; } return i; }
            return n;
        };
        auto raw_below_floor = [&](const uint8_t* k, uint32_t n, uint32_t* term_len) -> bool {
            const uint32_t e = raw_term_end(k, n);
            *term_len = e;
            if (e >= n) return false;
            const uint8_t* r = k + e + 1;
            uint64_t d;
            return get_sortable_uint(&r, k + n, &d) && d < min_did_all;
        };
        auto raw_term_wanted_in_full = [&](const uint8_t* k, uint32_t term_len) -> bool {
            if (!full_terms || full_terms->empty()) return false;
            std::string t;
            for (uint32_t i = 0; i < term_len; ++i) { t.push_back((char)k[i]); if (k[i] == 0) ++i; }
            return full_terms->count(t) != 0;
        };
        /* the floor in the keys' own encoding (pack_uint_preserving_sort, common/pack.h:183-219): a separator's docid may be cut short */
        uint8_t floor_enc[9];
        uint32_t floor_len = 0;
        if (min_did_all) {
            uint64_t v = min_did_all;
            if (v < 0x8000) { floor_enc[0] = (uint8_t)(v >> 8); floor_enc[1] = (uint8_t)v; floor_len = 2; }
            else {
                uint32_t len = 3;
                for (uint64_t x = v >> 22; x; x >>= 7) ++len;
                const unsigned mask = 0xFFu << (10 - len);
                for (uint32_t i = 1; i < len; ++i) { floor_enc[len - i] = (uint8_t)v; v >>= 8; }
                floor_enc[0] = (uint8_t)(v | mask);
                floor_len = len;
            }
            const uint8_t* chk = floor_enc;
            uint64_t back = 0;
            if (!get_sortable_uint(&chk, floor_enc + floor_len, &back) || back != min_did_all || chk != floor_enc + floor_len) floor_len = 0;   /* (never: then no block is skipped unseen) */
        }
        auto skip_child = [&](const uint8_t* a, uint32_t an, const uint8_t* z, uint32_t zn) -> bool {
            if (!floor_len) return false;
            const uint32_t ta = raw_term_end(a, an);
            if (ta >= an || zn < ta + 1 || memcmp(a, z, ta + 1) != 0) return false;       /* both separators: the same whole term */
            const uint8_t* r = z + ta + 1;
            const uint32_t rn = zn - ta - 1, m = rn < floor_len ? rn : floor_len;
            const int c = memcmp(r, floor_enc, m);
            /* every key of the block sorts below (term, floor): STRICTLY below the next separator is not guaranteed — an item cut
             * into components across the two blocks gives a separator equal to its key — so that separator itself must sort below */
            if (!(c < 0 || (c == 0 && rn < floor_len))) return false;
            return !raw_term_wanted_in_full(a, ta);
        };
        auto skip_block = [&](const uint8_t* a, uint32_t an, const uint8_t* z, uint32_t zn) -> bool {
            if (!min_did_all) return false;
            uint32_t ta, tz;
            if (!raw_below_floor(z, zn, &tz)) return false;
            ta = raw_term_end(a, an);
            if (ta != tz || ta >= an || memcmp(a, z, ta) != 0) return false;
            return !raw_term_wanted_in_full(a, ta);
        };
        auto skip_item = [&](const uint8_t* k, uint32_t n) -> bool {
            if (!min_did_all) return false;
            uint32_t tl;
            if (!raw_below_floor(k, n, &tl)) return false;
            return !raw_term_wanted_in_full(k, tl);
        };
        rc = posn.walk([&](const std::string& key, const std::string& tag) -> int {
            std::string t;
            const uint8_t* rest;
            split_term_key(key, &t, &rest);
            uint64_t d;
            if (!rest || !get_sortable_uint(&rest, (const uint8_t*)key.data() + key.size(), &d)) return xgm_set_error(XGM_E_INVALID, "position table: bad key");
            if (d < min_did_all) {
                if (!full_terms) return XGM_OK;
                if (t != pos_term) { pos_term = t; pos_term_full = full_terms->count(t) != 0; }
                if (!pos_term_full) return XGM_OK;
            }
            /* advance the posting cursor to (t, d); postings passed over have no positions */
            while (true) {
                if (ti >= ex->terms.size()) return xgm_set_error(XGM_E_INVALID, "position table: entry without a posting");
                const int c = ex->terms[ti].compare(t);
                if (c > 0) return xgm_set_error(XGM_E_INVALID, "position table: entry without a posting");
                if (c == 0 && pi < t_end && ex->did[pi] == d) break;
                if (c == 0 && (pi >= t_end || ex->did[pi] > d)) return xgm_set_error(XGM_E_INVALID, "position table: entry without a posting");
                if (pi < t_end) { ++pi; ex->pos_off.push_back(ex->pos.size()); }
                else { ++ti; if (ti < ex->terms.size()) t_end += ex->df[ti]; }
            }
            const uint8_t* p = (const uint8_t*)tag.data();
            const uint8_t* end = p + tag.size();
            uint64_t last;
            if (!get_varint(&p, end, &last)) return xgm_set_error(XGM_E_INVALID, "position list: truncated");
            if (p == end) {
                ex->pos.push_back((uint32_t)last);                               /* single entry */
            } else {
                BitReader rd{p, end};
                const uint32_t first = rd.decode((uint32_t)last);
                const uint32_t n = rd.decode((uint32_t)last - first) + 2u;
                tmp.assign(n, 0);
                tmp[0] = first; tmp[n - 1] = (uint32_t)last;
                rd.interpolative(tmp.data(), 0, (int)n - 1);
                if (!rd.ok) return xgm_set_error(XGM_E_INVALID, "position list: truncated bit stream");
                ex->pos.insert(ex->pos.end(), tmp.begin(), tmp.end());
            }
            ++pi;
            ex->pos_off.push_back(ex->pos.size());
            return XGM_OK;
        }, skip_child, skip_block, skip_item);
        if (rc) return rc;
        while (ex->pos_off.size() < ex->did.size() + 1) ex->pos_off.push_back(ex->pos.size());
    }
    if (ex->doccount != ex->ver.doccount || ex->total_length != ex->ver.total_doclen)
        return xgm_set_error(XGM_E_INVALID, "glass statistics disagree with the document-length list (%llu/%llu docs, %llu/%llu length)",
                             (unsigned long long)ex->doccount, (unsigned long long)ex->ver.doccount,
                             (unsigned long long)ex->total_length, (unsigned long long)ex->ver.total_doclen);
    return XGM_OK;
}

void fill_raw(const Export& ex, std::vector<const char*>* tp, std::vector<uint32_t>* tl, xgm_raw_postings* raw) {
    tp->clear(); tl->clear();
    for (const std::string& t : ex.terms) { tp->push_back(t.data()); tl->push_back((uint32_t)t.size()); }
    memset(raw, 0, sizeof *raw);
    raw->n_terms = (uint32_t)ex.terms.size();
    raw->lastdocid = (uint32_t)ex.ver.last_docid;
    raw->doccount = (uint32_t)ex.ver.doccount;
    /* the bounds as glass keeps them (get_doclength_lower_bound / get_wdf_upper_bound read the version file:
     * glass_database.cc:815-830): on a shard with deleted or replaced documents they are LOOSER than what the live
     * postings give, and BM25Weight::get_maxpart — hence MSet::get_max_possible — uses glass's */
    raw->doclen_lower_bound = (uint32_t)ex.ver.doclen_lbound;
    raw->wdf_upper_bound = (uint32_t)ex.ver.wdf_ubound;
    raw->doclen_upper_bound = (uint32_t)(ex.ver.wdf_ubound + ex.ver.doclen_ubound_delta);
    raw->has_positions = ex.has_positions ? 1u : 0u;
    raw->total_length = ex.ver.total_doclen;
    raw->n_postings = ex.did.size();
    raw->n_positions = ex.pos.size();
    raw->revision = ex.ver.revision;
    raw->doclen = ex.doclen.data();
    raw->terms = tp->data();
    raw->term_len = tl->data();
    raw->df = ex.df.data();
    raw->did = ex.did.data();
    raw->wdf = ex.wdf.data();
    raw->pos_off = ex.has_positions ? ex.pos_off.data() : nullptr;
    raw->pos = ex.has_positions ? ex.pos.data() : nullptr;
}

}  // namespace

extern "C" int xgm_segment_build_from_glass(const char* glass_dir, uint32_t stripe_bits, const char* out_path) {
    if (!glass_dir || !out_path) return xgm_set_error(XGM_E_INVALID, "null argument");
    Export ex;
    int rc = read_glass(glass_dir, &ex);
    if (rc) return rc;
    std::vector<const char*> tp;
    std::vector<uint32_t> tl;
    xgm_raw_postings raw;
    fill_raw(ex, &tp, &tl, &raw);
    return xgm_segment_build(&raw, stripe_bits, out_path);
}

/* ---- incremental refresh ---------------------------------------------------------------------- */

namespace {

inline uint32_t seg_bits(const uint32_t* w, uint64_t idx, uint32_t bw) {
    if (bw == 0) return 0;
    const uint64_t bit = idx * bw;
    const uint64_t v = (uint64_t)w[bit >> 5] | ((uint64_t)w[(bit >> 5) + 1] << 32);
    return (uint32_t)((v >> (bit & 31)) & (bw == 32 ? 0xFFFFFFFFull : ((1ull << bw) - 1)));
}

}  // namespace

int xgm_map_segment_file(const char* path, XgmSegmentBlob* blob);
int xgm_read_segment_file(const char* path, XgmSegmentBlob* blob);

/* Refresh a shard's segment after the shard moved to a newer revision, re-reading from glass only what changed.
 * Contract (the caller's — Xapiand knows the smallest docid its write-ahead log touched since the old revision, reference
 * src/database/wal.cc): every document below first_changed_docid is the same in the old segment's revision and in glass now.
 * Postings and positions of those documents come from the old segment (bit-unpacking at memory speed: no B-tree walk, no varint
 * or interpolative decoding); glass is read for the rest, whole posting chunks and position lists below the floor being skipped
 * by their headers / keys.  The merged postings go through the ordinary builder, so the result is byte for byte what a full
 * xgm_segment_build_from_glass of the new revision writes.  A cheap part of the contract is checked (document lengths below the
 * floor must agree): XGM_E_INVALID then means "do a full export".  Replaces: the reference reopening the shard on a revision
 * change (src/database/handler.cc:1282, 1333 — DatabaseModifiedError / reopen). */
extern "C" int xgm_segment_refresh_from_glass(const char* old_segment_path, const char* glass_dir, uint32_t first_changed_docid,
                                              uint32_t stripe_bits, const char* out_path) {
    if (!old_segment_path || !glass_dir || !out_path) return xgm_set_error(XGM_E_INVALID, "null argument");
    static const bool timing = getenv("XGM_REFRESH_TIMING") != nullptr;      /* diagnostics: where a refresh spends its time */
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "xgm refresh: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    XgmSegmentBlob old;
    /* the old segment is mapped, not copied — unless the new one is to replace that very file */
    struct stat st_old, st_out;
    const bool in_place = stat(old_segment_path, &st_old) == 0 && stat(out_path, &st_out) == 0 && st_old.st_dev == st_out.st_dev && st_old.st_ino == st_out.st_ino;
    int rc = in_place ? xgm_read_segment_file(old_segment_path, &old) : xgm_map_segment_file(old_segment_path, &old);
    if (rc) return rc;
    lap("map + validate old segment");
    const xgm_seg_header* h = old.header();
    const uint64_t X = first_changed_docid;
    if (X > (uint64_t)h->lastdocid + 1) return xgm_set_error(XGM_E_INVALID, "first_changed_docid %u beyond the old segment's last docid %u + 1", first_changed_docid, h->lastdocid);
    if (stripe_bits && stripe_bits != h->stripe_bits) return xgm_set_error(XGM_E_INVALID, "stripe width differs from the old segment's: full export needed");
    const uint32_t SB = h->stripe_bits;
    const uint64_t Xs = (X >> SB) << SB;         /* stripes below X's are unchanged as a whole: their blocks are copied verbatim */

    /* old dictionary + block tables */
    const uint32_t oT = h->n_terms;
    const uint64_t* so = old.section<uint64_t>(XGM_S_STR_OFF);
    const char* sb = old.section<char>(XGM_S_STR_BYTES);
    const uint32_t* oflags = old.section<uint32_t>(XGM_S_TERM_FLAGS);
    const uint32_t* ocf = old.section<uint32_t>(XGM_S_TERM_CF);
    const uint64_t* tb = old.section<uint64_t>(XGM_S_TERM_BLK);
    const uint64_t* tw = old.section<uint64_t>(XGM_S_TERM_WORD);
    const uint64_t* tp = old.section<uint64_t>(XGM_S_TERM_POS);
    const uint32_t* bf = old.section<uint32_t>(XGM_S_BLK_FIRST);
    const uint32_t* bm = old.section<uint32_t>(XGM_S_BLK_META);
    const uint32_t* bwd = old.section<uint32_t>(XGM_S_BLK_WORD);
    const uint32_t* bps = old.section<uint32_t>(XGM_S_BLK_POS);
    const uint32_t* words = old.section<uint32_t>(XGM_S_WORDS);
    const uint8_t* opos = old.section<uint8_t>(XGM_S_POSITIONS);
    auto old_term = [&](uint32_t t) { return std::string(sb + so[t], (size_t)(so[t + 1] - so[t])); };

    /* terms whose positions the old segment dropped (some posting had positions != wdf) cannot be reconstructed from it — and the
     * offending posting may be gone by now: those terms are read from glass in full */
    std::set<std::string> in_full;
    if (h->has_positions)
        for (uint32_t t = 0; t < oT; ++t) if (!(oflags[t] & XGM_TF_POS_OK)) in_full.insert(old_term(t));

    Export ex;
    if ((rc = read_glass(glass_dir, &ex, X, &in_full))) return rc;
    lap("read glass from the floor");
    if (ex.ver.revision < h->revision) return xgm_set_error(XGM_E_INVALID, "glass revision %llu is older than the segment's %llu", (unsigned long long)ex.ver.revision, (unsigned long long)h->revision);
    if ((h->has_positions != 0) != ex.has_positions && X > 1) return xgm_set_error(XGM_E_INVALID, "the shard gained or lost its position table: full export needed");
    const uint32_t* odl = old.section<uint32_t>(XGM_S_DOCLEN);
    for (uint64_t d = 1; d < X; ++d)
        if (d >= ex.doclen.size() || odl[d] != ex.doclen[d]) return xgm_set_error(XGM_E_INVALID, "document %llu below first_changed_docid differs from the old segment: full export needed", (unsigned long long)d);

    const bool has_pos = ex.has_positions;
    xgm_raw_postings raw;
    std::vector<const char*> tpv;
    std::vector<uint32_t> tl;
    {
        Export stub;                                 /* statistics and document lengths of the new revision (no postings) */
        stub.ver = ex.ver; stub.has_positions = has_pos;
        fill_raw(stub, &tpv, &tl, &raw);
        raw.doclen = ex.doclen.data();
    }
    uint32_t wdf_seen = 0, doclen_lb, doclen_ub, wdf_ub_db;
    for (uint32_t w : ex.wdf) wdf_seen = std::max(wdf_seen, w);
    xgm_database_bounds(&raw, std::max(wdf_seen, X > 1 ? h->wdf_upper_bound : 0u), &doclen_lb, &doclen_ub, &wdf_ub_db);
    XgmSegmentWriter w;
    if ((rc = w.begin(SB, has_pos, wdf_ub_db))) return rc;
    if (X > 1) w.reserve_like(old);

    std::vector<uint32_t> t_did, t_wdf, t_pos;     /* the term's postings that are encoded afresh: old ones in [Xs, X), then glass's */
    std::vector<uint64_t> t_poff;
    uint64_t np = 0;                               /* cursor into ex's postings */
    auto take_new = [&](size_t i) {
        const uint32_t n = ex.df[i];
        for (uint32_t k = 0; k < n; ++k) {
            t_did.push_back(ex.did[np + k]); t_wdf.push_back(ex.wdf[np + k]);
            if (has_pos) {
                for (uint64_t q = ex.pos_off[np + k]; q < ex.pos_off[np + k + 1]; ++q) t_pos.push_back(ex.pos[q]);
                t_poff.push_back(t_pos.size());
            }
        }
        np += n;
    };
    uint32_t oi = 0;
    size_t ni = 0;
    while (oi < oT || ni < ex.terms.size()) {
        int c;
        std::string ot;
        if (oi < oT) ot = old_term(oi);
        if (oi >= oT) c = 1;
        else if (ni >= ex.terms.size()) c = -1;
        else c = ot.compare(ex.terms[ni]);
        const bool from_old = c <= 0 && X > 1 && !in_full.count(ot);
        const std::string& name = c <= 0 ? ot : ex.terms[ni];
        t_did.clear(); t_wdf.clear(); t_pos.clear(); t_poff.assign(1, 0);
        uint64_t b_copy = 0, cf_copied = 0;
        uint32_t first_wdf = 0;
        bool copied16 = true;
        if (from_old) {
            const uint32_t t = oi;
            const bool o16 = (oflags[t] & XGM_TF_POS16) != 0;
            /* blocks wholly below Xs are copied; the rest of the old list is decoded: what is below X is kept (and re-encoded),
             * everything decoded is subtracted from the term's collection frequency to get the copied part's */
            uint64_t lo = tb[t], hi = tb[t + 1];
            while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2; if (bf[mid] < Xs) lo = mid + 1; else hi = mid; }
            b_copy = lo;
            uint64_t cf_rest = 0;
            for (uint64_t b = b_copy; b < tb[t + 1]; ++b) {
                const uint32_t cnt = XGM_META_COUNT(bm[b]), bwg = XGM_META_BWG(bm[b]), bww = XGM_META_BWW(bm[b]);
                const uint32_t* gw = words + tw[t] + bwd[b];
                const uint32_t* ww = gw + ((uint64_t)cnt * bwg + 31) / 32;
                uint32_t d = bf[b];
                uint64_t pe = bps[b];
                for (uint32_t j = 0; j < cnt; ++j) {
                    if (j) d += seg_bits(gw, j, bwg) + 1;
                    const uint32_t wd = seg_bits(ww, j, bww);
                    cf_rest += wd;
                    if (d < X) {
                        t_did.push_back(d); t_wdf.push_back(wd);
                        if (has_pos) {
                            for (uint32_t q = 0; q < wd; ++q) {
                                const uint8_t* e = opos + tp[t] + (pe + q) * (o16 ? 2u : 4u);
                                t_pos.push_back(o16 ? (uint32_t)e[0] | ((uint32_t)e[1] << 8) : (uint32_t)e[0] | ((uint32_t)e[1] << 8) | ((uint32_t)e[2] << 16) | ((uint32_t)e[3] << 24));
                            }
                            t_poff.push_back(t_pos.size());
                        }
                    }
                    pe += wd;
                }
            }
            if (b_copy > tb[t]) {
                if (ocf[t] != 0xFFFFFFFFu) cf_copied = (uint64_t)ocf[t] - cf_rest;
                else {                                                  /* a saturated collection frequency: count the copied part */
                    for (uint64_t b = tb[t]; b < b_copy; ++b) {
                        const uint32_t cnt = XGM_META_COUNT(bm[b]), bwg = XGM_META_BWG(bm[b]), bww = XGM_META_BWW(bm[b]);
                        const uint32_t* ww = words + tw[t] + bwd[b] + ((uint64_t)cnt * bwg + 31) / 32;
                        for (uint32_t j = 0; j < cnt; ++j) cf_copied += seg_bits(ww, j, bww);
                    }
                }
                const uint64_t b0 = tb[t];
                first_wdf = seg_bits(words + tw[t] + bwd[b0] + ((uint64_t)XGM_META_COUNT(bm[b0]) * XGM_META_BWG(bm[b0]) + 31) / 32, 0, XGM_META_BWW(bm[b0]));
                if (has_pos && !o16) {
                    /* 4-byte entries: because of a position >= 65536 among the copied postings — or among postings that are gone? */
                    const uint64_t e_end = b_copy < tb[t + 1] ? bps[b_copy] : cf_copied;
                    const uint8_t* src = opos + tp[t];
                    bool small = true;
                    for (uint64_t e = 0; small && e < e_end; ++e) small = src[4 * e + 2] == 0 && src[4 * e + 3] == 0;
                    copied16 = small;
                }
            }
        }
        if (c >= 0) take_new(ni);
        bool pos_ok = has_pos, pos16 = has_pos && copied16;
        if (has_pos && !t_did.empty()) {
            bool ok, p16;
            xgm_positional_form(t_wdf.data(), (uint32_t)t_did.size(), t_poff.data(), t_pos.data(), &ok, &p16);
            pos_ok = ok; pos16 = ok && p16 && copied16;
        }
        const bool any = (from_old && b_copy > tb[oi]) || !t_did.empty();
        if (any) {
            if ((rc = w.begin_term(name.data(), (uint32_t)name.size(), pos_ok, pos16))) return rc;
            if (from_old && b_copy > tb[oi] && (rc = w.copy_blocks(old, oi, b_copy, cf_copied, first_wdf))) return rc;
            if (!t_did.empty() && (rc = w.add_postings(t_did.data(), t_wdf.data(), (uint32_t)t_did.size(), has_pos ? t_poff.data() : nullptr, t_pos.data()))) return rc;
            if ((rc = w.end_term())) return rc;
        }
        if (c <= 0) ++oi;
        if (c >= 0) ++ni;
    }
    raw.n_postings = w.n_postings;
    lap("merge terms / copy blocks");
    rc = w.finish(&raw, doclen_lb, doclen_ub, nullptr, out_path);
    lap("write");
    return rc;
}

/* ---- value slots (widening row (f).3: the columns a value sort / collapse on the device will read) ---------------------- */

/* One value slot of a committed glass shard as a COLUMN file: per document the rank of its value among the slot's distinct
 * values in bytewise order — what Xapian's value sorts compare (matcher/msetcmp.cc:64-101) and what set_collapse_key groups
 * by — plus the distinct values themselves (the host needs the strings for MSet items and for merging shards).
 *   "XGMCOL1\0", u32 slot, u32 lastdocid, u32 n_distinct, u32 0,
 *   u32 ord[lastdocid + 1]        0 = the document has no value in the slot (the empty string: sorts first), else 1 + rank
 *   u64 off[n_distinct + 1], bytes the distinct values, ascending
 * Value chunks live in the postlist table under "\0\xd8" pack_uint(slot) pack_uint_preserving_sort(first docid); a chunk is
 * pack_string(value of the first docid) { pack_uint(docid increase - 1) pack_string(value) }* (reference
 * backends/glass/glass_values.h:41-47, glass_values.cc:72-95).  Replaces: ValueStreamDocument / Document::get_value per
 * candidate in the matcher (matcher/matcher.cc:509-517). */
extern "C" int xgm_glass_export_column(const char* glass_dir, uint32_t slot, const char* out_path) {
    if (!glass_dir || !out_path) return xgm_set_error(XGM_E_INVALID, "null argument");
    const std::string dir(glass_dir);
    GlassVersion ver;
    int rc = read_version(dir, &ver);
    if (rc) return rc;
    if (ver.last_docid > 0xFFFFFFFEull) return xgm_set_error(XGM_E_INVALID, "docids beyond 32 bits");
    std::vector<std::string> value((size_t)ver.last_docid + 1);
    Table post;
    if ((rc = post.open(dir + "/postlist.glass", ver.root[kTablePostlist]))) return rc;
    rc = post.walk([&](const std::string& key, const std::string& tag) -> int {
        if (key.size() < 3 || key[0] != '\0' || (uint8_t)key[1] != 0xD8) return XGM_OK;
        const uint8_t* kp = (const uint8_t*)key.data() + 2;
        const uint8_t* kend = (const uint8_t*)key.data() + key.size();
        uint64_t s, did;
        if (!get_varint(&kp, kend, &s)) return xgm_set_error(XGM_E_INVALID, "value chunk: bad key");
        if (s != slot) return XGM_OK;
        if (!get_sortable_uint(&kp, kend, &did)) return xgm_set_error(XGM_E_INVALID, "value chunk: bad key");
        const uint8_t* p = (const uint8_t*)tag.data();
        const uint8_t* end = p + tag.size();
        bool first = true;
        while (first || p < end) {
            uint64_t len;
            if (!first) {
                uint64_t inc;
                if (!get_varint(&p, end, &inc)) return xgm_set_error(XGM_E_INVALID, "value chunk: truncated");
                did += inc + 1;
            }
            first = false;
            if (!get_varint(&p, end, &len) || len > (uint64_t)(end - p)) return xgm_set_error(XGM_E_INVALID, "value chunk: truncated value");
            if (did >= value.size()) return xgm_set_error(XGM_E_INVALID, "value chunk: docid beyond the last one");
            value[did].assign((const char*)p, (size_t)len);
            p += len;
        }
        return XGM_OK;
    });
    if (rc) return rc;
    std::set<std::string> distinct;
    for (const std::string& v : value) if (!v.empty()) distinct.insert(v);
    std::vector<const std::string*> sorted;
    for (const std::string& v : distinct) sorted.push_back(&v);
    std::vector<uint32_t> ord(value.size(), 0u);
    for (size_t d = 0; d < value.size(); ++d) {
        if (value[d].empty()) continue;
        const auto it = std::lower_bound(sorted.begin(), sorted.end(), value[d], [](const std::string* a, const std::string& b) { return *a < b; });
        ord[d] = (uint32_t)(it - sorted.begin()) + 1u;
    }
    FILE* f = fopen(out_path, "wb");
    if (!f) return xgm_set_error(XGM_E_IO, "cannot create %s: %s", out_path, strerror(errno));
    const uint32_t h32[4] = {slot, (uint32_t)ver.last_docid, (uint32_t)sorted.size(), 0u};
    bool ok = fwrite("XGMCOL1", 1, 8, f) == 8 && fwrite(h32, 4, 4, f) == 4 && fwrite(ord.data(), 4, ord.size(), f) == ord.size();
    uint64_t off = 0;
    for (const std::string* v : sorted) { ok = ok && fwrite(&off, 8, 1, f) == 1; off += v->size(); }
    ok = ok && fwrite(&off, 8, 1, f) == 1;
    for (const std::string* v : sorted) ok = ok && fwrite(v->data(), 1, v->size(), f) == v->size();
    if (fclose(f) != 0 || !ok) return xgm_set_error(XGM_E_IO, "short write on %s", out_path);
    return XGM_OK;
}

extern "C" int xgm_glass_info(const char* glass_dir, uint64_t* revision, uint32_t* doccount, uint32_t* lastdocid, uint64_t* total_length) {
    if (!glass_dir) return xgm_set_error(XGM_E_INVALID, "null argument");
    GlassVersion v;
    int rc = read_version(glass_dir, &v);
    if (rc) return rc;
    if (revision) *revision = v.revision;
    if (doccount) *doccount = (uint32_t)v.doccount;
    if (lastdocid) *lastdocid = (uint32_t)v.last_docid;
    if (total_length) *total_length = v.total_doclen;
    return XGM_OK;
}

extern "C" int xgm_glass_export_raw(const char* glass_dir, const char* raw_path) {
    if (!glass_dir || !raw_path) return xgm_set_error(XGM_E_INVALID, "null argument");
    Export ex;
    int rc = read_glass(glass_dir, &ex);
    if (rc) return rc;
    FILE* f = fopen(raw_path, "wb");
    if (!f) return xgm_set_error(XGM_E_IO, "cannot create %s: %s", raw_path, strerror(errno));
    bool ok = true;
    auto put = [&](const void* p, size_t bytes) {
        static const char zeros[8] = {0};
        ok = ok && fwrite(p, 1, bytes, f) == bytes;
        if (bytes % 8) ok = ok && fwrite(zeros, 1, 8 - bytes % 8, f) == 8 - bytes % 8;
    };
    uint64_t str_total = 0;
    std::vector<uint32_t> lens;
    for (const std::string& t : ex.terms) { lens.push_back((uint32_t)t.size()); str_total += t.size(); }
    const uint32_t h32[4] = {(uint32_t)ex.terms.size(), (uint32_t)ex.ver.last_docid, (uint32_t)ex.ver.doccount, ex.has_positions ? 1u : 0u};
    const uint64_t h64[5] = {ex.ver.total_doclen, ex.did.size(), ex.pos.size(), ex.ver.revision, str_total};
    ok = fwrite("XGMRAW1", 1, 8, f) == 8 && fwrite(h32, 4, 4, f) == 4 && fwrite(h64, 8, 5, f) == 5;
    put(ex.doclen.data(), ex.doclen.size() * 4);
    put(ex.df.data(), ex.df.size() * 4);
    put(ex.did.data(), ex.did.size() * 4);
    put(ex.wdf.data(), ex.wdf.size() * 4);
    if (ex.has_positions) {
        put(ex.pos_off.data(), ex.pos_off.size() * 8);
        put(ex.pos.data(), ex.pos.size() * 4);
    }
    put(lens.data(), lens.size() * 4);
    for (const std::string& t : ex.terms) ok = ok && fwrite(t.data(), 1, t.size(), f) == t.size();
    if (fclose(f) != 0 || !ok) return xgm_set_error(XGM_E_IO, "short write on %s", raw_path);
    return XGM_OK;
}
