/* xgm_xapiand_glue — the Xapiand side of seam B1: what keeps the device's segment of a shard in step with the shard's committed
 * revision.  Two calls, placed by integration/xapiand_shard_hook.patch in the reference's src/database/shard.cc:
 *
 *   xgm_xapiand::on_commit(path, db, first_changed_docid)    after a local Shard::commit moved the revision on (shard.cc:752-760,
 *       where endpoint.set_revision(current_revision) is called): the shard's glass directory is exported — incrementally from the
 *       previous segment when the write-ahead log knows the smallest docid it touched since (src/database/wal.cc), else in full —,
 *       loaded onto the device and registered with the matcher hook under (UUID, revision); the previous revision's index is
 *       released once no search can reach it.  Until this returns, searches on the new revision are DECLINED by the hook (it
 *       checks the registry's revision) and answered by the CPU matcher: never a stale answer.
 *   xgm_xapiand::on_close(db)                                Shard::do_close: the shard leaves the registry, its index is released.
 *
 * Nothing else of Xapiand is touched: the HTTP layer, the query DSL, DocMatcher and Enquire stay as they are and reach the device
 * through Matcher::get_mset (integration/matcher_hook.patch).  Configuration (environment, read once): XGM_DEVICE (HIP device
 * ordinal, default 0), XGM_SEGMENT_DIR (where segments are written; default: next to the shard, <path>/.xgm), XGM_BATCHING
 * (max batch of the index's micro-batching queue for single-query callers, default 256; 0 = off).
 * Run by oracle/ref_build/hook_b1_driver.cc --commit-glue (tests/test_gpu_hook_b1.py::test_commit_glue_and_http_bodies). */
#ifndef XGM_XAPIAND_GLUE_H
#define XGM_XAPIAND_GLUE_H

#include <cstdint>
#include <string>

#include "xapian.h"

namespace xgm_xapiand {

/* first_changed_docid: the smallest docid modified since the previously registered revision (0 = unknown: full export).  Returns
 * true when the new revision is registered on the device; false leaves the shard on the CPU matcher (the reason is logged to stderr
 * once per shard). */
bool on_commit(const std::string& shard_path, const Xapian::Database& db, uint32_t first_changed_docid);
void on_close(const Xapian::Database& db);

struct Stats { uint64_t full_exports, refreshes, failures, released; };
Stats stats();

}  // namespace xgm_xapiand

#endif
