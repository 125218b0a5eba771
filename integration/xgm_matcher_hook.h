/* Seam B1: the matcher hook.  Compiled INTO the reference's Xapian library (it sees Xapian's
 * internal headers) together with integration/matcher_hook.patch, which adds one call to
 * Matcher::get_mset (reference src/xapian/matcher/matcher.cc:543-609) in front of get_local_mset (:346-542):
 *
 *     if (!xgm_hook::try_get_mset(...,  local_mset))  local_mset = get_local_mset(...);
 *
 * For an eligible search (SURVEY §8(b) predicate: one local shard, BM25Weight with k2 = 0, relevance order, no
 * collapse / cut-offs / decider / spies / RSet / time limit, a query shape xgm_plan_query accepts) the hook lowers the
 * Xapian::Query to an xgm_query_desc, takes the MERGED statistics out of the Weight::Internal the matcher was
 * handed (Xapiand's add_prepared_mset / set_prepared_mset protocol, src/database/handler.cc:1532-1549, needs nothing
 * else), runs xgm_get_mset_batch on the shard's device-resident segment and builds the MSet exactly like
 * ProtoMSet::finalise does (src/xapian/matcher/protomset.h:672-682).  Anything else — or an xgm return > 0 — leaves
 * the CPU matcher to run, untouched.  A hard failure (< 0) throws Xapian::DatabaseError so that Xapiand's retry
 * logic (src/database/handler.cc:1348-1368) applies.
 *
 * Shards are registered by the embedding server (Xapiand: where it opens / reopens a shard) under the glass
 * database's UUID together with the revision the segment was exported from; a search on a Database whose revision
 * moved on is declined (CPU path) until the refreshed segment is registered — the revision key of SURVEY §8(f).1.
 */
#ifndef XGM_MATCHER_HOOK_H
#define XGM_MATCHER_HOOK_H

#include <cstdint>
#include <vector>

#include "xapian/database.h"
#include "xapian/enquire.h"
#include "xapian/mset.h"
#include "xapian/query.h"
#include "xapian/weight.h"

#include "xgm.h"

namespace xgm_hook {

/* registry (thread-safe) */
void register_shard(const Xapian::Database& db, xgm_index* idx);
void unregister_shard(const Xapian::Database& db);

/* switches (process-wide) */
void set_enabled(bool on);                 /* default: on */
/* PHRASE / NEAR with maxitems < matches: the reference's SelectPostList serves a stale cached weight
 * (src/xapian/matcher/selectpostlist.cc:28-55), the device returns the intended top-k.  A deployment
 * that needs byte-compatibility with the CPU matcher on such searches declines them here. */
void set_decline_positional(bool on);      /* default: off */

struct Counters { uint64_t answered, declined_shape, declined_unregistered, declined_revision, declined_device; };
Counters counters();

/* The call the patch adds.  sort_by_rel: Enquire::Internal::sort_by == REL.  Returns true and fills `out` when the
 * search ran on the device. */
bool try_get_mset(const Xapian::Database& db, const Xapian::Query& query, const Xapian::Weight::Internal& stats,
                  const Xapian::Weight& wtscheme, bool full_db_has_positions, Xapian::doccount first,
                  Xapian::doccount maxitems, Xapian::doccount check_at_least, const Xapian::MatchDecider* mdecider,
                  const Xapian::KeyMaker* sorter, Xapian::doccount collapse_max, int percent_threshold,
                  double weight_threshold, Xapian::Enquire::docid_order order, bool sort_by_rel, double time_limit,
                  size_t n_matchspies, Xapian::MSet& out);

}  // namespace xgm_hook

#endif
