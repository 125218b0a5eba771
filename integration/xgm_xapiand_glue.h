/* xgm_xapiand_glue — the Xapiand side of seam B1: what keeps the device's segment of a shard in step with the shard's committed
 * revision.  Three calls, placed by integration/xapiand_shard_hook.patch in the reference's src/database/shard.cc:
 *
 *   xgm_xapiand::on_touch(db, docid)           wherever the writer adds, replaces or deletes a document (shard.cc:917, 1028, 1227-1229,
 *       1367-1369, 1589-1591): the smallest docid touched since the last commit — what the write-ahead log would tell (src/database/wal.cc) —
 *       so that the next export can be INCREMENTAL; docid 0 = unknown (a delete / replace by term).
 *   xgm_xapiand::on_commit(path, db)           after a local Shard::commit moved the revision on (shard.cc:752-760, where
 *       endpoint.set_revision(current_revision) is called): the revision is QUEUED for export and the call returns — the writer does not
 *       wait (round 5 exported under the shard's lock: 50 s for 10 M documents).  A worker thread reads the shard's glass tables —
 *       incrementally from the previous segment when every commit since knew its floor, else in full —, checks that the revision it read is
 *       STILL the committed one (glass re-uses the blocks of a revision two commits later: an export overtaken by the writer is thrown away and
 *       the newer revision exported), loads the segment onto the device and registers it with the matcher hook under (UUID, revision).  Until
 *       then searches on the new revision are DECLINED by the hook (it checks the registry's revision) and answered by the CPU matcher: never a
 *       stale answer.  XGM_EXPORT_SYNC=1: export inside the call (tests; a deployment that prefers a blocked writer to CPU answers).
 *   xgm_xapiand::on_close(db)                  Shard::do_close OF THE WRITABLE LOCAL SHARD (the patch guards it: readers of the pool share the
 *       UUID): the shard leaves the registry.
 * Index lifetime: the matcher hook owns the device index (xgm_hook::register_shard_owned): replacing or unregistering a revision drops the
 * registry's reference, the index is closed and its segment file removed when the LAST search that had picked it up has returned.
 *
 * Nothing else of Xapiand is touched: the HTTP layer, the query DSL, DocMatcher and Enquire stay as they are and reach the device
 * through Matcher::get_mset (integration/matcher_hook.patch).  Configuration (environment, read once): XGM_DEVICE (HIP device
 * ordinal, default 0), XGM_SEGMENT_DIR (where segments are written; default: next to the shard, <path>/.xgm), XGM_BATCHING
 * (max batch of the index's micro-batching queue for single-query callers, default 256; 0 = off), XGM_EXPORT_SYNC.
 * Run by oracle/ref_build/hook_b1_driver.cc --commit-glue (tests/test_gpu_hook_b1.py::test_commit_glue_and_http_bodies). */
#ifndef XGM_XAPIAND_GLUE_H
#define XGM_XAPIAND_GLUE_H

#include <cstdint>
#include <string>

#include "xapian.h"

namespace xgm_xapiand {

void on_touch(const Xapian::Database& db, uint32_t docid);
/* Queues (XGM_EXPORT_SYNC: performs) the export of db's committed revision; the floor is what on_touch saw since the last commit.  Returns
 * true when the revision is queued — or, synchronously, registered on the device; false leaves the shard on the CPU matcher (the reason is
 * logged to stderr). */
bool on_commit(const std::string& shard_path, const Xapian::Database& db);
/* ... with the floor given by the caller: the smallest docid modified since the previous commit (0 = unknown: full export) */
constexpr uint32_t kFloorTracked = 0xFFFFFFFFu;
bool on_commit(const std::string& shard_path, const Xapian::Database& db, uint32_t first_changed_docid);
void on_close(const Xapian::Database& db);
/* blocks until every queued export has been registered or given up (tests, orderly shutdown) */
void wait_idle();

struct Stats { uint64_t full_exports, refreshes, failures, released, overtaken; };
Stats stats();

}  // namespace xgm_xapiand

#endif
