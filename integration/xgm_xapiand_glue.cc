/* See xgm_xapiand_glue.h. */
#include "xgm_xapiand_glue.h"

#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "xgm.h"
#include "xgm_matcher_hook.h"

namespace xgm_xapiand {
namespace {

/* one shard of this process: what the device holds of it and what its writer has committed since */
struct ShardState {
    std::string path;
    std::string segment;                 /* segment file of the registered revision ("" = none yet) */
    uint64_t registered = 0;             /* revision the matcher hook answers (0 = none) */
    uint64_t target = 0;                 /* newest committed revision asked for (> registered: an export is pending or running) */
    uint64_t gave_up = 0;                /* revision whose export FAILED (unreadable tables, no device memory, ...): not tried again — the next commit is */
    bool closed = false;                 /* on_close ran: a finished export is thrown away */
    uint64_t generation = 0;             /* bumped by on_close: an export started before it belongs to a shard that is gone */
    /* smallest docid touched by each commit since the registered revision: (revision, floor); floor 0 = unknown (a delete by term, ...) */
    std::vector<std::pair<uint64_t, uint32_t>> floors;
    uint32_t touching = 0xFFFFFFFFu;     /* smallest docid touched since the last commit (0xFFFFFFFF: nothing yet) */
};
/* (never destroyed: the worker thread is detached and waits on these for the life of the process — a condition variable destroyed under a waiter
 *  at exit blocks in pthread_cond_destroy) */
std::mutex& g_mu = *new std::mutex;
std::condition_variable& g_cv_work = *new std::condition_variable;
std::condition_variable& g_cv_idle = *new std::condition_variable;
std::map<std::string, ShardState>& g_by_uuid = *new std::map<std::string, ShardState>;
std::deque<std::string>& g_queue = *new std::deque<std::string>;         /* uuids with target > registered, oldest request first */
bool g_worker_started = false, g_worker_busy = false;
std::atomic<uint64_t> g_full{0}, g_refresh{0}, g_fail{0}, g_released{0}, g_stale{0};

int device() { static const int d = getenv("XGM_DEVICE") ? atoi(getenv("XGM_DEVICE")) : 0; return d; }
uint32_t batching() { static const uint32_t b = getenv("XGM_BATCHING") ? (uint32_t)atoi(getenv("XGM_BATCHING")) : 256u; return b; }
bool synchronous() { static const bool s = getenv("XGM_EXPORT_SYNC") != nullptr; return s; }

std::string segment_path(const std::string& shard_path, uint64_t revision) {
    const char* dir = getenv("XGM_SEGMENT_DIR");
    std::string base = dir ? std::string(dir) : shard_path + "/.xgm";
    mkdir(base.c_str(), 0755);
    std::string tag = shard_path;
    for (char& c : tag) if (c == '/') c = '_';
    return base + "/" + (dir ? tag + "." : std::string()) + "rev" + std::to_string(revision) + ".seg";
}

/* Export revision `target` of the shard (incrementally from the registered segment when the floor is known), load it, hand it to the matcher
 * hook.  Runs WITHOUT g_mu; the glass tables are read straight from disk (xgm_glass.cc).  Off the writer's thread the writer may commit again
 * meanwhile: glass then re-uses the blocks revision `target` freed, so the export is kept only if `target` is STILL the committed revision when
 * the read is over (the version file is read again) — otherwise it is thrown away and the newer revision exported instead. */
void export_one(const std::string& uuid, const Xapian::Database* db_sync) {
    std::string path, old_segment;
    uint64_t target = 0, registered = 0, generation = 0;
    uint32_t floor = 0xFFFFFFFFu;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_by_uuid.find(uuid);
        if (it == g_by_uuid.end() || it->second.closed || it->second.target <= it->second.registered) return;
        ShardState& st = it->second;
        path = st.path; old_segment = st.segment; target = st.target; registered = st.registered; generation = st.generation;
        for (const auto& f : st.floors) if (f.first > registered && f.first <= target) floor = f.second == 0u ? 0u : (floor == 0u ? 0u : std::min(floor, f.second));
        if (floor == 0xFFFFFFFFu) floor = 0u;                      /* (no record of what the commits touched: full export) */
    }
    uint64_t rev_before = 0, rev_after = 0;
    if (xgm_glass_info(path.c_str(), &rev_before, nullptr, nullptr, nullptr) != XGM_OK || rev_before != target) {
        /* the version file is not (or no longer) at the revision asked for: a newer commit has its own request queued */
        ++g_stale;
        return;
    }
    const std::string seg = segment_path(path, target);
    int rc = XGM_E_INVALID;
    bool refreshed = false;
    if (!old_segment.empty() && floor != 0u) {
        rc = xgm_segment_refresh_from_glass(old_segment.c_str(), path.c_str(), floor, 0, seg.c_str());
        refreshed = rc == XGM_OK;
    }
    if (rc != XGM_OK) rc = xgm_segment_build_from_glass(path.c_str(), 0, seg.c_str());      /* no previous segment, unknown floor, or the floor's contract did not hold */
    if (rc == XGM_OK && (xgm_glass_info(path.c_str(), &rev_after, nullptr, nullptr, nullptr) != XGM_OK || rev_after != target)) {
        ++g_stale;                                                                         /* the writer committed while the tables were read: not a snapshot */
        unlink(seg.c_str());
        return;
    }
    xgm_index* idx = nullptr;
    if (rc == XGM_OK) rc = xgm_index_open(seg.c_str(), device(), target, &idx);
    if (rc != XGM_OK) {
        ++g_fail;
        fprintf(stderr, "xgm: shard %s revision %llu stays on the CPU matcher: %s\n", path.c_str(), (unsigned long long)target, xgm_last_error());
        unlink(seg.c_str());
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_by_uuid.find(uuid);
        if (it != g_by_uuid.end() && it->second.generation == generation) it->second.gave_up = target;
        return;
    }
    if (refreshed) ++g_refresh; else ++g_full;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_by_uuid.find(uuid);
    if (it == g_by_uuid.end() || it->second.closed || it->second.generation != generation) {
        xgm_index_close(idx);                                       /* the shard was closed while its revision was exported */
        unlink(seg.c_str());
        return;
    }
    ShardState& st = it->second;
    /* From here on the hook answers searches on `target` and declines the old revision.  The hook OWNS the index: it is closed — and its
     * segment file removed — when the registry has replaced it AND every search that had picked it up has returned (shared ownership: a
     * reader of the pool still inside the library on the old revision keeps it alive). */
    const std::string seg_copy = seg;
    if (db_sync) xgm_hook::register_shard_owned(*db_sync, idx, batching(), [seg_copy]() { unlink(seg_copy.c_str()); ++g_released; });
    else xgm_hook::register_shard_owned(uuid, target, idx, batching(), [seg_copy]() { unlink(seg_copy.c_str()); ++g_released; });
    st.segment = seg;
    st.registered = target;
    size_t keep = 0;
    for (size_t i = 0; i < st.floors.size(); ++i) if (st.floors[i].first > target) st.floors[keep++] = st.floors[i];
    st.floors.resize(keep);
}

void worker_loop() {
    std::unique_lock<std::mutex> lk(g_mu);
    while (true) {
        g_cv_work.wait(lk, [] { return !g_queue.empty(); });
        const std::string uuid = g_queue.front();
        g_queue.pop_front();
        g_worker_busy = true;
        lk.unlock();
        export_one(uuid, nullptr);
        lk.lock();
        g_worker_busy = false;
        /* a commit that arrived while this one was exported (or an export thrown away as stale) leaves target > registered: again */
        auto it = g_by_uuid.find(uuid);
        if (it != g_by_uuid.end() && !it->second.closed && it->second.target > it->second.registered) {
            uint64_t rev = 0;
            const std::string path = it->second.path;
            lk.unlock();
            const bool readable = xgm_glass_info(path.c_str(), &rev, nullptr, nullptr, nullptr) == XGM_OK;
            lk.lock();
            it = g_by_uuid.find(uuid);
            if (readable && it != g_by_uuid.end() && !it->second.closed && rev > it->second.registered && rev != it->second.gave_up) {
                it->second.target = std::max(it->second.target, rev);
                bool queued = false;
                for (const std::string& u : g_queue) queued = queued || u == uuid;
                if (!queued) g_queue.push_back(uuid);
            }
        }
        if (g_queue.empty()) g_cv_idle.notify_all();
    }
}

}  // namespace

void on_touch(const Xapian::Database& db, uint32_t docid) {
    const std::string uuid = db.get_uuid();
    std::lock_guard<std::mutex> lk(g_mu);
    ShardState& st = g_by_uuid[uuid];
    if (docid == 0u) st.touching = 0u;
    else if (st.touching != 0u) st.touching = std::min(st.touching, docid);
}

bool on_commit(const std::string& shard_path, const Xapian::Database& db, uint32_t first_changed_docid) {
    const std::string uuid = db.get_uuid();
    const uint64_t revision = db.get_revision();
    bool sync = synchronous();
    {
        std::lock_guard<std::mutex> lk(g_mu);
        ShardState& st = g_by_uuid[uuid];
        if (st.closed) { st = ShardState(); }                       /* (a shard opened again under its UUID) */
        st.path = shard_path;
        if (st.registered == revision || st.target == revision) return st.registered == revision;      /* (a commit that did not move the revision) */
        /* what this commit touched: the caller's floor, else what on_touch saw since the last commit (0 = unknown) */
        uint32_t floor = first_changed_docid != kFloorTracked ? first_changed_docid : (st.touching == 0xFFFFFFFFu ? 0u : st.touching);
        st.touching = 0xFFFFFFFFu;
        st.floors.emplace_back(revision, floor);
        st.target = revision;
        if (!sync) {
            bool queued = false;
            for (const std::string& u : g_queue) queued = queued || u == uuid;
            if (!queued) g_queue.push_back(uuid);
            if (!g_worker_started) { g_worker_started = true; std::thread(worker_loop).detach(); }
        }
    }
    if (!sync) { g_cv_work.notify_one(); return true; }
    export_one(uuid, &db);
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_by_uuid.find(uuid);
    return it != g_by_uuid.end() && it->second.registered == revision;
}

bool on_commit(const std::string& shard_path, const Xapian::Database& db) { return on_commit(shard_path, db, kFloorTracked); }

void wait_idle() {
    std::unique_lock<std::mutex> lk(g_mu);
    g_cv_idle.wait(lk, [] { return g_queue.empty() && !g_worker_busy; });
}

void on_close(const Xapian::Database& db) {
    const std::string uuid = db.get_uuid();
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_by_uuid.find(uuid);
        if (it == g_by_uuid.end()) return;
        it->second.closed = true;
        ++it->second.generation;
        it->second.registered = it->second.target = 0;
        it->second.segment.clear();
        it->second.floors.clear();
    }
    /* the registry lets go of the index: it is closed when the last search that holds it returns (xgm_hook::register_shard_owned) */
    xgm_hook::unregister_shard(db);
}

Stats stats() { return Stats{g_full.load(), g_refresh.load(), g_fail.load(), g_released.load(), g_stale.load()}; }

}  // namespace xgm_xapiand
