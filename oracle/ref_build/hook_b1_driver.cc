/* xapian_hook_b1 — seam B1 compiled and run: the REAL vendored Xapian with
 * integration/matcher_hook.patch applied to its Matcher::get_mset and integration/xgm_matcher_hook.cc linked in.
 * Every query goes through the reference's own Enquire::get_mset TWICE — hook off (CPU matcher) and hook on (libxgm
 * behind the matcher) — on one shard and, with several <dbdir>s, through Xapiand's per-shard protocol
 * (prepare_mset → add_prepared_mset → set_prepared_mset → get_mset → unshard_docids → merge_mset; reference
 * src/database/handler.cc:1250-1343, 1532-1549), and the two MSets must agree: size, firstitem, docid and weight
 * BITS at every rank, percentages (single shard), max_possible, max_attained.  matches_* are exact counts on the
 * device path (documented exception); they are checked against the CPU matcher's bounds: lower <= exact <= upper.
 *
 *   xapian_hook_b1 [--decline-positional] [--stale] <queries.txt> <dbdir> [<dbdir> ...]
 *   xapian_hook_b1 --leg <name>:<mode>:<queries.txt> [--leg ...] - <dbdir> [...]      several query files against ONE export + load of the
 *       shards (bench.py's hook_parity at 10 M documents: the export is the expensive part); mode = plain | exact-bounds |
 *       positional-reference (page + exact figures) | positional-reference-page | positional-intended; one JSON line per leg ("leg": name), the exit code covers all of them
 * Each shard's segment is exported from its glass directory by the native reader (xgm_segment_build_from_glass) and
 * loaded onto device 0.  --stale registers every shard under a wrong revision: every search must then be declined
 * (CPU path) and still answer identically.  Test infrastructure (tests/test_gpu_hook_b1.py); query file format as
 * xapian_ref. */
#define XGM_REF_DRIVER_NO_MAIN
#include "ref_driver.cc"

#include <unistd.h>
#include <sys/stat.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>

#include "../../integration/xgm_matcher_hook.h"
#include "../../integration/xgm_xapiand_glue.h"

#include "../../integration/xgm_aggregation_adapter.h"
bool xapiand_stand_in_lookup(std::string_view field, Xapian::valueno* slot, bool* integral);      /* xapiand_classes.cc: the stand-in for Schema::get_slot_field */

namespace {

bool same_mset(const Xapian::MSet& a, const Xapian::MSet& b, bool percents, std::string* why) {
    if (a.size() != b.size()) { *why = "size"; return false; }
    if (a.get_firstitem() != b.get_firstitem()) { *why = "firstitem"; return false; }
    auto x = a.begin();
    auto y = b.begin();
    for (; x != a.end(); ++x, ++y) {
        const double wa = x.get_weight(), wb = y.get_weight();
        if (*x != *y) { *why = "docid"; return false; }
        if (memcmp(&wa, &wb, sizeof wa) != 0) { *why = "weight bits"; return false; }
        if (percents && x.get_percent() != y.get_percent()) { *why = "percent"; return false; }
        if (x.get_sort_key() != y.get_sort_key()) { *why = "sort key"; return false; }
        if (x.get_collapse_key() != y.get_collapse_key()) { *why = "collapse key"; return false; }
        if (x.get_collapse_count() != y.get_collapse_count()) { *why = "collapse count"; return false; }
    }
    const double pa = a.get_max_possible(), pb = b.get_max_possible(), ma = a.get_max_attained(), mb = b.get_max_attained();
    if (memcmp(&pa, &pb, 8) != 0) { *why = "max_possible"; return false; }
    if (a.size() && memcmp(&ma, &mb, 8) != 0) { *why = "max_attained"; return false; }
    return true;
}

/* The body of Xapiand's search response as HttpClient::search_view forms it from the MSet (reference src/server/http_client.cc:2544-2599,
 * field names src/response.h:25-43): "aggregations" (the spy's result, when the request had _aggs), "count" = mset.size(), "total" =
 * mset.get_matches_estimated(), and per hit — with ?comments — "#docid", "#shard" = (docid - 1) % shards + 1, "#rank", "#weight",
 * "#percent".  The stored document (its _id and data) is read from the document store after the match and is not part of this path.
 * A restatement of those 50 lines over the REAL MSet: the server itself (cmake, 376 k lines, its own event loop) is not built here. */
std::string http_body(const Xapian::MSet& m, size_t n_shards, const std::string& aggregation) {
    std::string body = "{";
    if (!aggregation.empty()) body += "\"aggregations\":" + aggregation.substr(0, aggregation.find('|')) + ",";
    body += "\"hits\":[";
    char buf[256];
    bool firsth = true;
    for (auto it = m.begin(); it != m.end(); ++it) {
        const Xapian::docid did = *it;
        snprintf(buf, sizeof buf, "%s{\"#docid\":%u,\"#shard\":%zu,\"#rank\":%u,\"#weight\":%.17g,\"#percent\":%d}", firsth ? "" : ",", did,
                 (size_t)((did - 1) % n_shards) + 1, it.get_rank(), it.get_weight(), it.get_percent());
        body += buf;
        firsth = false;
    }
    snprintf(buf, sizeof buf, "],\"count\":%u,\"total\":%u}", m.size(), m.get_matches_estimated());
    body += buf;
    return body;
}

}  // namespace

int main(int argc, char** argv) {
    int a = 1;
    bool stale = false, exact_bounds_on = false, replay_on = false, positional_reference_on = false, commit_glue = false, commit_during = false;
    unsigned n_threads = 0, thread_repeat = 1;
    const char* bodies_dir = nullptr;
    struct Leg { std::string name, mode, file; };
    std::vector<Leg> legs;
    xgm_hook::set_positional_mode(xgm_hook::POSITIONAL_INTENDED);        /* the deployment's choice; the tests pick per run */
    {   /* the driver's own spy class reaches the device through an adapter (how Xapiand binds AggregationMatchSpy) */
        xgm_hook::SpyAdapter ad;
        ad.slot_of = [](const Xapian::MatchSpy& s, Xapian::valueno* slot) {
            const DriverCountSpy* d = dynamic_cast<const DriverCountSpy*>(&s);
            if (!d) return false;
            *slot = d->slot;
            return true;
        };
        ad.feed = [](Xapian::MatchSpy& s, Xapian::doccount total, const std::vector<std::pair<std::string, Xapian::doccount>>& counts) {
            DriverCountSpy& d = static_cast<DriverCountSpy&>(s);
            d.total += total;
            for (const auto& kv : counts) d.values[kv.first] += kv.second;
        };
        xgm_hook::register_spy_adapter("DriverCountSpy", ad);
    }
    /* Xapiand's own AggregationMatchSpy, compiled from the reference: the PRODUCT adapter (integration/xgm_aggregation_adapter.cc), registered as a Xapiand
     * build would — with this harness's stand-in for the Schema lookup */
    xgm_xapiand::register_aggregation_adapter(xapiand_stand_in_lookup);
    for (; a < argc && argv[a][0] == '-'; ++a) {
        if (!strcmp(argv[a], "--decline-positional")) xgm_hook::set_positional_mode(xgm_hook::POSITIONAL_DECLINE);
        else if (!strcmp(argv[a], "--positional-reference")) { xgm_hook::set_positional_mode(xgm_hook::POSITIONAL_REFERENCE); positional_reference_on = true; }
        else if (!strcmp(argv[a], "--exact-bounds")) { xgm_hook::set_exact_bounds(true); exact_bounds_on = true; }
        else if (!strcmp(argv[a], "--collapse-intended")) xgm_hook::set_collapse_mode(xgm_hook::COLLAPSE_INTENDED);
        else if (!strcmp(argv[a], "--collapse-reference")) { xgm_hook::set_collapse_mode(xgm_hook::COLLAPSE_REFERENCE); replay_on = true; }
        else if (!strcmp(argv[a], "--replay")) { xgm_hook::set_replay(true); replay_on = true; }
        else if (!strcmp(argv[a], "--stale")) stale = true;
        else if (!strcmp(argv[a], "--threads") && a + 1 < argc) n_threads = (unsigned)atoi(argv[++a]);      /* Xapiand's load shape: T threads, each its own Database handles, one get_mset at a time through the patched matcher (src/manager.cc:161, src/database/handler.cc:1338) */
        else if (!strcmp(argv[a], "--thread-repeat") && a + 1 < argc) thread_repeat = (unsigned)std::max(1, atoi(argv[++a]));
        else if (!strcmp(argv[a], "--commit-during")) commit_during = true;   /* ... while the writer commits and the glue replaces the registered revision under them (needs --commit-glue) */
        else if (!strcmp(argv[a], "--commit-glue")) commit_glue = true;       /* shards reach the device through integration/xgm_xapiand_glue.cc (what xapiand_shard_hook.patch calls) */
        else if (!strcmp(argv[a], "--http-bodies") && a + 1 < argc) bodies_dir = argv[++a];      /* write every response body (hook off / on) there */
        else if (!strcmp(argv[a], "--leg") && a + 1 < argc) {
            const std::string spec = argv[++a];
            const size_t c1 = spec.find(':'), c2 = c1 == std::string::npos ? c1 : spec.find(':', c1 + 1);
            if (c2 == std::string::npos) { fprintf(stderr, "--leg wants <name>:<mode>:<queries.txt>\n"); return 2; }
            legs.push_back(Leg{spec.substr(0, c1), spec.substr(c1 + 1, c2 - c1 - 1), spec.substr(c2 + 1)});
        }
        else if (!strcmp(argv[a], "-")) { ++a; break; }
        else if (!strcmp(argv[a], "--near-colocated")) xgm_hook::set_near_colocated_terms(true);     /* the indexer may put several terms at one position (nearpostlist.cc:106-140) */
    }
    if (!legs.empty()) --a;                                  /* (no query file argument: the shard directories follow) */
    if (argc - a < 2) { fprintf(stderr, "usage: xapian_hook_b1 [--decline-positional] [--stale] <queries.txt> <dbdir> [<dbdir> ...]\n"); return 2; }
    try {
        auto queries = legs.empty() ? read_queries(argv[a]) : std::vector<QuerySpec>();
        std::vector<Xapian::Database> dbs;
        std::vector<xgm_index*> idx;
        std::vector<std::string> seg_files;
        double export_s = 0.0, open_s = 0.0;
        unsigned long long segment_bytes = 0;
        auto now_s = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        for (int i = a + 1; i < argc; ++i) {
            dbs.emplace_back(argv[i]);
            if (commit_glue) {
                /* Xapiand's way in: Shard::commit → xgm_xapiand::on_commit (export, load, register); the first time in full */
                const double t_e0 = now_s();
                /* (queued: the writer does not wait for the export — this driver does, before it searches) */
                if (!xgm_xapiand::on_commit(argv[i], dbs.back(), 0)) { fprintf(stderr, "on_commit(%s) failed\n", argv[i]); return 1; }
                xgm_xapiand::wait_idle();
                if (xgm_xapiand::stats().failures) { fprintf(stderr, "the export of %s failed\n", argv[i]); return 1; }
                export_s += now_s() - t_e0;
                idx.push_back(nullptr);
                seg_files.push_back("");
                continue;
            }
            char seg[64];
            snprintf(seg, sizeof seg, "/tmp/xgm_b1_%d_%d.seg", (int)getpid(), i);
            const double t_e0 = now_s();
            if (xgm_segment_build_from_glass(argv[i], 0, seg) != XGM_OK) { fprintf(stderr, "export %s: %s\n", argv[i], xgm_last_error()); return 1; }
            const double t_e1 = now_s();
            export_s += t_e1 - t_e0;
            { struct stat st; if (stat(seg, &st) == 0) segment_bytes += (unsigned long long)st.st_size; }
            xgm_index* h = nullptr;
            const uint64_t rev = dbs.back().get_revision();
            if (xgm_index_open(seg, 0, rev, &h) != XGM_OK) { fprintf(stderr, "xgm_index_open: %s\n", xgm_last_error()); return 1; }
            open_s += now_s() - t_e1;
            seg_files.push_back(seg);                 /* kept: the incremental refresh starts from it */
            idx.push_back(h);
            xgm_hook::register_shard(dbs.back(), h);
        }
        unsigned refreshed = 0;
        /* the shard moved on: refresh its segment INCREMENTALLY — documents below first_changed come from the old segment, only
         * the rest is read from glass (xgm_segment_refresh_from_glass; byte-identical to a full export, tests/test_glass.py) */
        auto export_and_register = [&](size_t i, const char* dir, uint32_t first_changed) -> int {
            if (commit_glue) {                       /* (an incremental refresh from the registered segment: the floor as Shard's mutators report it, xgm_xapiand::on_touch) */
                xgm_xapiand::on_touch(dbs[i], first_changed);
                const bool ok = xgm_xapiand::on_commit(dir, dbs[i]);
                xgm_xapiand::wait_idle();
                return ok && !xgm_xapiand::stats().failures ? 0 : 1;
            }
            char seg[64];
            snprintf(seg, sizeof seg, "/tmp/xgm_b1_%d_%zu_r.seg", (int)getpid(), i);
            if (xgm_segment_refresh_from_glass(seg_files[i].c_str(), dir, first_changed, 0, seg) != XGM_OK) { fprintf(stderr, "refresh %s: %s\n", dir, xgm_last_error()); return 1; }
            xgm_index* h = nullptr;
            if (xgm_index_open(seg, 0, dbs[i].get_revision(), &h) != XGM_OK) { fprintf(stderr, "xgm_index_open: %s\n", xgm_last_error()); return 1; }
            unlink(seg_files[i].c_str());
            seg_files[i] = seg;
            if (idx[i]) xgm_index_close(idx[i]);
            idx[i] = h;
            xgm_hook::register_shard(dbs[i], h);
            return 0;
        };
        if (stale) {
            /* The shards move on to a new revision behind the registered segments (one more document each, committed
             * through the reference's WritableDatabase); the re-opened Database handles then carry a revision the
             * registry does not know: every search must be DECLINED (CPU matcher) ... */
            std::vector<uint32_t> first_changed;
            for (size_t i = 0; i < dbs.size(); ++i) {
                first_changed.push_back(dbs[i].get_lastdocid() + 1);            /* documents are only appended below */
                {
                    Xapian::WritableDatabase w(argv[a + 1 + i], Xapian::DB_OPEN);
                    Xapian::Document doc;
                    doc.add_posting("t1", 1); doc.add_posting("t2", 2); doc.add_posting("t3", 3);
                    w.add_document(doc);
                    w.commit();
                }
                dbs[i] = Xapian::Database(argv[a + 1 + i]);
            }
            const xgm_hook::Counters before = xgm_hook::counters();
            for (size_t qi = 0; qi < queries.size() && qi < 8; ++qi) (void)run_query(dbs, make_query(queries[qi]), queries[qi].first, queries[qi].maxitems);
            const xgm_hook::Counters after = xgm_hook::counters();
            if (after.answered != before.answered || after.declined_revision == before.declined_revision) {
                printf("STALE: searches on a moved-on revision were not declined (answered %llu -> %llu, declined_revision %llu -> %llu)\n",
                       (unsigned long long)before.answered, (unsigned long long)after.answered, (unsigned long long)before.declined_revision,
                       (unsigned long long)after.declined_revision);
                return 1;
            }
            /* ... until the refreshed segments (keyed by the new revision) are registered */
            for (size_t i = 0; i < dbs.size(); ++i) { if (export_and_register(i, argv[a + 1 + i], first_changed[i])) return 1; ++refreshed; }
        }
        unsigned bad_total = 0;
        if (legs.empty()) legs.push_back(Leg{"", "", ""});
        for (const Leg& leg : legs) {
        if (!leg.file.empty()) {
            queries = read_queries(leg.file.c_str());
            /* (positional-reference: the reference's page AND its match-count figures; positional-reference-page: the page alone — the faster mode) */
            exact_bounds_on = leg.mode == "exact-bounds" || leg.mode == "positional-reference";
            positional_reference_on = leg.mode == "positional-reference" || leg.mode == "positional-reference-page";
            xgm_hook::set_exact_bounds(exact_bounds_on);
            xgm_hook::set_positional_mode(positional_reference_on ? xgm_hook::POSITIONAL_REFERENCE : xgm_hook::POSITIONAL_INTENDED);
        }
        const xgm_hook::Counters c0 = leg.file.empty() ? xgm_hook::Counters{} : xgm_hook::counters();     /* (the classic single run reports the process's totals: --stale counts declines before the loop) */
        unsigned bad = 0, bounds_bad = 0, http_total_equal = 0, http_bodies_equal = 0;
        double cpu_s = 0.0, hook_s = 0.0;
        const bool percents = dbs.size() == 1;
        std::vector<Xapian::MSet> wants;                             /* the CPU matcher's answers, for the threaded leg */
        std::vector<SpyResult> spy_wants;
        for (size_t qi = 0; qi < queries.size(); ++qi) {
            const QuerySpec& q = queries[qi];
            const Xapian::Query query = make_query(q);
            SpyResult spy_want, spy_got;
            xgm_hook::set_enabled(false);
            const double t_q0 = now_s();
            Xapian::MSet want = run_query(dbs, query, q.first, q.maxitems, &q, &spy_want);
            const double t_q1 = now_s();
            xgm_hook::set_enabled(true);
            Xapian::MSet got = run_query(dbs, query, q.first, q.maxitems, &q, &spy_got);
            cpu_s += t_q1 - t_q0; hook_s += now_s() - t_q1;
            std::string why;
            if (n_threads) { wants.push_back(want); spy_wants.push_back(spy_want); }
            if (!same_mset(want, got, percents, &why)) {
                ++bad;
                printf("MISMATCH query %zu (%s): %s; cpu %u hits, hook %u hits\n", qi, q.op.c_str(), why.c_str(), want.size(), got.size());
            } else if (spy_want.aggregation != spy_got.aggregation) {
                ++bad;
                printf("MISMATCH query %zu (%s): Xapiand's aggregation: cpu %s\n   hook %s\n", qi, q.op.c_str(), spy_want.aggregation.substr(0, 300).c_str(), spy_got.aggregation.substr(0, 300).c_str());
            } else if (spy_want.total != spy_got.total || spy_want.values != spy_got.values) {
                ++bad;
                printf("MISMATCH query %zu (%s): spy: cpu saw %u documents / %zu values, hook %u / %zu\n", qi, q.op.c_str(), spy_want.total, spy_want.values.size(),
                       spy_got.total, spy_got.values.size());
            }
            /* Xapiand's HTTP response: "total" = mset.get_matches_estimated(), "_percent" = get_percent() of every hit (reference
             * src/server/http_client.cc:2553-2554, 2598) — the percentages are part of same_mset above */
            if (want.get_matches_estimated() == got.get_matches_estimated()) ++http_total_equal;
            {
                const std::string bw = http_body(want, dbs.size(), spy_want.aggregation), bg = http_body(got, dbs.size(), spy_got.aggregation);
                if (bw == bg) ++http_bodies_equal;
                if (bodies_dir) {
                    char fn[512];
                    snprintf(fn, sizeof fn, "%s/q%04zu.cpu.json", bodies_dir, qi);
                    if (FILE* f = fopen(fn, "w")) { fputs(bw.c_str(), f); fputc('\n', f); fclose(f); }
                    snprintf(fn, sizeof fn, "%s/q%04zu.hook.json", bodies_dir, qi);
                    if (FILE* f = fopen(fn, "w")) { fputs(bg.c_str(), f); fputc('\n', f); fclose(f); }
                }
            }
            /* the upper bound is a static property of the postlist tree: identical; the lower bound may be looser than the CPU
             * matcher's (which counts the documents it happened to weigh) but never above it or the estimate.  Where the value leads
             * the sort the matcher shows ProtoMSet every document: all three figures must be the reference's. */
            /* ... and, with --exact-bounds, by relevance for EVERY operator and a match of ANY size (round 4: the device hands back the
             * whole match in docid order, xgm_search_all; rounds 1-3 required it only when the match fit one device page): a term, AND,
             * FILTER, AND_NOT through xgm_known_matching_docs, OR / AND_MAYBE / trees through the replay of the reference's own loop;
             * positional queries in --positional-reference mode (the frozen-weight replay counts like ProtoMSet does); the reference
             * also knows the exact count whenever its three figures coincide — then the hook must report the same */
            const bool positional = q.op == "PHRASE" || q.op == "NEAR";
            const bool exact_bounds = !q.collapse_max && (q.sort_mode == "V" || q.sort_mode == "VR" || q.sort_mode == "K" || q.sort_mode == "KR" ||
                                                          (exact_bounds_on && !positional && q.sort_mode.empty()) ||
                                                          (positional_reference_on && exact_bounds_on && positional && q.sort_mode.empty()) ||
                                                          (want.get_matches_lower_bound() == want.get_matches_upper_bound()));
            if (dbs.size() == 1 && replay_on && (q.collapse_max || q.cut_percent || q.cut_weight != 0.0 || q.spy_slot >= 0)) {
                /* replayed through the reference's own collation: every figure is the CPU matcher's */
                if (want.get_matches_lower_bound() != got.get_matches_lower_bound() || want.get_matches_estimated() != got.get_matches_estimated() ||
                    want.get_matches_upper_bound() != got.get_matches_upper_bound() ||
                    want.get_uncollapsed_matches_lower_bound() != got.get_uncollapsed_matches_lower_bound() ||
                    want.get_uncollapsed_matches_estimated() != got.get_uncollapsed_matches_estimated() ||
                    want.get_uncollapsed_matches_upper_bound() != got.get_uncollapsed_matches_upper_bound()) {
                    ++bounds_bad;
                    printf("BOUNDS (replay) query %zu: hook [%u, %u, %u] vs CPU matcher [%u, %u, %u]\n", qi, got.get_matches_lower_bound(), got.get_matches_estimated(),
                           got.get_matches_upper_bound(), want.get_matches_lower_bound(), want.get_matches_estimated(), want.get_matches_upper_bound());
                }
            } else if (dbs.size() == 1 && !q.collapse_max &&
                (want.get_matches_upper_bound() != got.get_matches_upper_bound() || got.get_matches_lower_bound() > want.get_matches_lower_bound() ||
                 got.get_matches_lower_bound() > got.get_matches_estimated() || got.get_matches_estimated() > got.get_matches_upper_bound() ||
                 (exact_bounds && (want.get_matches_lower_bound() != got.get_matches_lower_bound() || want.get_matches_estimated() != got.get_matches_estimated())))) {
                ++bounds_bad;
                printf("BOUNDS query %zu: hook [%u, %u, %u] vs CPU matcher [%u, %u, %u]\n", qi, got.get_matches_lower_bound(), got.get_matches_estimated(),
                       got.get_matches_upper_bound(), want.get_matches_lower_bound(), want.get_matches_estimated(), want.get_matches_upper_bound());
            }
        }
        /* ---- the hook under Xapiand's load shape: T threads, each with its OWN Database handles (a Xapian::Database is not thread-safe: the pool
         * hands every HTTP worker its own Shard), one Enquire::get_mset at a time through the patched matcher; the hook's single-query calls meet in
         * the index's dispatcher (xgm_index_set_batching).  Every answer is compared with the CPU matcher's.  --commit-during: meanwhile the writer
         * commits and the glue registers the new revision — the readers' revision is replaced UNDER them (they hold the old index through their
         * calls, then are declined: their handles still name the old revision) ---- */
        unsigned thr_bad = 0;
        unsigned long long thr_done = 0, thr_device = 0;
        double thr_s = 0.0;
        if (n_threads && !queries.empty()) {
            std::vector<std::string> paths;
            for (int i = a + 1; i < argc; ++i) paths.push_back(argv[i]);
            std::atomic<size_t> next{0};
            std::atomic<unsigned> bad_t{0};
            std::vector<std::mutex> want_mu(queries.size());
            const size_t total = queries.size() * thread_repeat;
            const xgm_hook::Counters t0c = xgm_hook::counters();
            xgm_hook::set_enabled(true);
            /* (handles opened before the clock starts; every thread checks its own revision against the registry through the hook) */
            std::vector<std::vector<Xapian::Database>> handles(n_threads);
            for (unsigned t = 0; t < n_threads; ++t) for (const std::string& p : paths) handles[t].emplace_back(p);
            const double t_b = now_s();
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < n_threads; ++t)
                pool.emplace_back([&, t]() {
                    try {
                        for (size_t i = next.fetch_add(1); i < total; i = next.fetch_add(1)) {
                            const size_t qi = i % queries.size();
                            const QuerySpec& q = queries[qi];
                            SpyResult spy;
                            Xapian::MSet got = run_query(handles[t], make_query(q), q.first, q.maxitems, &q, &spy);
                            std::string why;
                            bool same;
                            {   /* (a Xapian::MSet is a handle on a reference-counted body whose count is NOT atomic, and iterating one copies the handle:
                                 *  two threads comparing against the same expected MSet at once — searches that shared a launch return together — would
                                 *  corrupt the count and free the body under each other) */
                                std::lock_guard<std::mutex> lk(want_mu[qi]);
                                same = same_mset(wants[qi], got, percents, &why);
                            }
                            if (!same) { if (bad_t.fetch_add(1) < 5) printf("MISMATCH (thread %u) query %zu: %s\n", t, qi, why.c_str()); }
                            else if (spy.total != spy_wants[qi].total || spy.values != spy_wants[qi].values || spy.aggregation != spy_wants[qi].aggregation) {
                                if (bad_t.fetch_add(1) < 5) printf("MISMATCH (thread %u) query %zu: spy: cpu saw %u documents / %zu values, hook %u / %zu\n", t, qi,
                                                                   spy_wants[qi].total, spy_wants[qi].values.size(), spy.total, spy.values.size());
                            }
                        }
                    } catch (const Xapian::Error& e) {
                        bad_t.fetch_add(1);
                        printf("MISMATCH (thread %u): %s\n", t, e.get_description().c_str());
                    }
                });
            if (commit_during && commit_glue) {
                /* the writer: one more document and a commit per shard while the readers search; Shard::commit's patch line follows */
                for (size_t i = 0; i < paths.size(); ++i) {
                    Xapian::WritableDatabase w(paths[i], Xapian::DB_OPEN);
                    Xapian::Document doc;
                    doc.add_posting("t1", 1); doc.add_posting("t2", 2);
                    const Xapian::docid did = w.add_document(doc).did;
                    w.commit();
                    xgm_xapiand::on_touch(w, did);
                    xgm_xapiand::on_commit(paths[i], w);
                }
                xgm_xapiand::wait_idle();
            }
            for (auto& th : pool) th.join();
            thr_s = now_s() - t_b;
            thr_bad = bad_t.load();
            thr_done = total;
            thr_device = xgm_hook::counters().answered - t0c.answered;
            bad += thr_bad;
        }
        xgm_hook::Counters c = xgm_hook::counters();
        c.answered -= c0.answered; c.declined_shape -= c0.declined_shape; c.declined_unregistered -= c0.declined_unregistered; c.declined_revision -= c0.declined_revision;
        c.declined_device -= c0.declined_device; c.answered_sorted -= c0.answered_sorted; c.answered_spied -= c0.answered_spied; c.answered_collapsed -= c0.answered_collapsed;
        c.replayed -= c0.replayed; c.combined -= c0.combined; c.combined_launches -= c0.combined_launches;
        if (!leg.name.empty()) printf("{\"leg\": \"%s\", \"mode\": \"%s\", ", leg.name.c_str(), leg.mode.c_str());
        printf("%s\"queries\": %zu, \"shards\": %zu,", leg.name.empty() ? "{" : "", queries.size(), dbs.size());
        printf(" \"mismatches\": %u, \"bounds_violations\": %u, \"answered_on_device\": %llu, \"declined_shape\": %llu, "
               "\"declined_unregistered\": %llu, \"declined_revision\": %llu, \"declined_by_planner\": %llu, \"refreshed_shards\": %u, "
               "\"answered_sorted\": %llu, \"answered_spied\": %llu, \"answered_collapsed\": %llu, \"columns_built\": %llu, \"http_total_equal\": %u, \"replayed\": %llu, "
               "\"docs\": %u, \"export_seconds\": %.3f, \"segment_bytes\": %llu, \"open_seconds\": %.3f, \"cpu_matcher_seconds\": %.3f, \"hook_seconds\": %.3f, "
               "\"http_bodies_equal\": %u, \"glue_full_exports\": %llu, \"glue_refreshes\": %llu, \"glue_failures\": %llu, \"glue_released\": %llu, \"glue_overtaken\": %llu, "
               "\"threads\": %u, \"threaded_queries\": %llu, \"threaded_seconds\": %.4f, \"threaded_mismatches\": %u, \"threaded_answered_on_device\": %llu, \"commit_during\": %s, "
               "\"combined_searches\": %llu, \"combined_launches\": %llu}\n",
               bad, bounds_bad, (unsigned long long)c.answered, (unsigned long long)c.declined_shape,
               (unsigned long long)c.declined_unregistered, (unsigned long long)c.declined_revision, (unsigned long long)c.declined_device, refreshed,
               (unsigned long long)c.answered_sorted, (unsigned long long)c.answered_spied, (unsigned long long)c.answered_collapsed, (unsigned long long)c.columns_built, http_total_equal, (unsigned long long)c.replayed,
               (unsigned)dbs[0].get_doccount(), export_s, segment_bytes, open_s, cpu_s, hook_s, http_bodies_equal,
               (unsigned long long)xgm_xapiand::stats().full_exports, (unsigned long long)xgm_xapiand::stats().refreshes, (unsigned long long)xgm_xapiand::stats().failures,
               (unsigned long long)xgm_xapiand::stats().released, (unsigned long long)xgm_xapiand::stats().overtaken,
               n_threads, thr_done, thr_s, thr_bad, thr_device, (commit_during && commit_glue) ? "true" : "false",
               (unsigned long long)c.combined, (unsigned long long)c.combined_launches);
        fflush(stdout);
        bad_total += bad + bounds_bad;
        }   /* legs */
        if (commit_glue) { xgm_xapiand::wait_idle(); for (auto& d : dbs) xgm_xapiand::on_close(d); }
        else for (auto& d : dbs) xgm_hook::unregister_shard(d);
        for (auto* h : idx) if (h) xgm_index_close(h);
        for (const std::string& f : seg_files) if (!f.empty()) unlink(f.c_str());
        return bad_total ? 1 : 0;
    } catch (const Xapian::Error& e) {
        fprintf(stderr, "Xapian error: %s\n", e.get_description().c_str());
        return 1;
    }
}
