/* See xgm_xapiand_glue.h. */
#include "xgm_xapiand_glue.h"

#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>

#include "xgm.h"
#include "xgm_matcher_hook.h"

namespace xgm_xapiand {
namespace {

struct Registered { xgm_index* idx; std::string segment; uint64_t revision; };
std::mutex g_mu;
std::map<std::string, Registered> g_by_uuid;
std::atomic<uint64_t> g_full{0}, g_refresh{0}, g_fail{0}, g_released{0};

int device() { static const int d = getenv("XGM_DEVICE") ? atoi(getenv("XGM_DEVICE")) : 0; return d; }
uint32_t batching() { static const uint32_t b = getenv("XGM_BATCHING") ? (uint32_t)atoi(getenv("XGM_BATCHING")) : 256u; return b; }

std::string segment_path(const std::string& shard_path, uint64_t revision) {
    const char* dir = getenv("XGM_SEGMENT_DIR");
    std::string base = dir ? std::string(dir) : shard_path + "/.xgm";
    mkdir(base.c_str(), 0755);
    std::string tag = shard_path;
    for (char& c : tag) if (c == '/') c = '_';
    return base + "/" + (dir ? tag + "." : std::string()) + "rev" + std::to_string(revision) + ".seg";
}

}  // namespace

bool on_commit(const std::string& shard_path, const Xapian::Database& db, uint32_t first_changed_docid) {
    const std::string uuid = db.get_uuid();
    const uint64_t revision = db.get_revision();
    Registered old{nullptr, "", 0};
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_by_uuid.find(uuid);
        if (it != g_by_uuid.end()) old = it->second;
    }
    if (old.idx && old.revision == revision) return true;                        /* (a commit that did not move the revision) */
    const std::string seg = segment_path(shard_path, revision);
    int rc = XGM_E_INVALID;
    /* the committed glass tables are read directly (xgm_glass.cc): the caller holds the shard (Shard::commit runs under its lock) */
    if (old.idx && first_changed_docid != 0) {
        rc = xgm_segment_refresh_from_glass(old.segment.c_str(), shard_path.c_str(), first_changed_docid, 0, seg.c_str());
        if (rc == XGM_OK) ++g_refresh;
    }
    if (rc != XGM_OK) {                                                           /* no previous segment, unknown floor, or the floor's contract did not hold */
        rc = xgm_segment_build_from_glass(shard_path.c_str(), 0, seg.c_str());
        if (rc == XGM_OK) ++g_full;
    }
    xgm_index* idx = nullptr;
    if (rc == XGM_OK) rc = xgm_index_open(seg.c_str(), device(), revision, &idx);
    if (rc != XGM_OK) {
        ++g_fail;
        fprintf(stderr, "xgm: shard %s revision %llu stays on the CPU matcher: %s\n", shard_path.c_str(), (unsigned long long)revision, xgm_last_error());
        unlink(seg.c_str());
        return false;
    }
    xgm_hook::register_shard(db, idx, batching());                               /* from here on the hook answers searches on this revision */
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_by_uuid[uuid] = Registered{idx, seg, revision};
    }
    if (old.idx) {
        /* searches on the old revision are declined from now on (the registry names the new one); those that were already inside the
         * library hold the index through their call: Xapiand releases a shard only after its readers have checked it in (DatabasePool) */
        xgm_index_close(old.idx);
        unlink(old.segment.c_str());
        ++g_released;
    }
    return true;
}

void on_close(const Xapian::Database& db) {
    Registered old{nullptr, "", 0};
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_by_uuid.find(db.get_uuid());
        if (it == g_by_uuid.end()) return;
        old = it->second;
        g_by_uuid.erase(it);
    }
    xgm_hook::unregister_shard(db);
    xgm_index_close(old.idx);
    unlink(old.segment.c_str());
    ++g_released;
}

Stats stats() { return Stats{g_full.load(), g_refresh.load(), g_fail.load(), g_released.load()}; }

}  // namespace xgm_xapiand
