/* Seam B1: the matcher hook.  Compiled INTO the reference's Xapian library (it sees Xapian's
 * internal headers) together with integration/matcher_hook.patch, which adds one call to
 * Matcher::get_mset (reference src/xapian/matcher/matcher.cc:543-609) in front of get_local_mset (:346-542):
 *
 *     if (!xgm_hook::try_get_mset(...,  local_mset))  local_mset = get_local_mset(...);
 *
 * For an eligible search (SURVEY §8(b) predicate: one local shard, BM25Weight with k2 = 0, ascending docid order, no cut-offs /
 * decider / RSet / time limit, a query shape xgm_plan_query accepts) the hook lowers the Xapian::Query to an xgm_query_desc, takes
 * the MERGED statistics out of the Weight::Internal the matcher was handed (Xapiand's add_prepared_mset / set_prepared_mset
 * protocol, src/database/handler.cc:1532-1549, needs nothing else), runs the search on the shard's device-resident segment and
 * builds the MSet exactly like ProtoMSet::finalise does (src/xapian/matcher/protomset.h:672-682).  Anything else — or an xgm
 * return > 0 — leaves the CPU matcher to run, untouched.  A hard failure (< 0) throws Xapian::DatabaseError so that Xapiand's
 * retry logic (src/database/handler.cc:1348-1368) applies.
 *
 * Row (f).3 of SURVEY §8 — what DocMatcher::prepare_mset sets on every Enquire (src/database/handler.cc:1263-1270):
 *   * value sorts (Enquire::set_sort_by_value / _then_relevance / relevance_then_value) and KEY sorts
 *     (set_sort_by_key_then_relevance(Multi_MultiValueKeyMaker), which is how Xapiand sorts): the hook keeps, per shard
 *     revision, one device COLUMN per value slot (read once through the shard's ValueIterator) or per KeyMaker (identified by
 *     name() + serialise(); its key for every document, made once) — ordinals of the distinct strings — and ranks on those;
 *     MSet items carry the sort key strings (MSetIterator::get_sort_key).
 *   * MatchSpies: Xapian::ValueCountMatchSpy natively (counted on the device in the same pass, delivered through its own
 *     merge_results), other spy classes through a registered SpyAdapter (Xapiand: AggregationMatchSpy for its value-count
 *     aggregations, src/aggregations/aggregations.h:107).  Taken only where the reference's matcher shows a spy EVERY matching
 *     document — a sort the value leads (protomset.h:249-283), or check_at_least covering the match — because only then are a spy's
 *     counts a property of the query rather than of the CPU matcher's traversal; otherwise the search is left to the CPU matcher.
 *   * Enquire::set_collapse_key: opt-in (set_collapse_mode), see below.
 *
 * Shards are registered by the embedding server (Xapiand: where it opens / reopens a shard) under the glass
 * database's UUID together with the revision the segment was exported from; a search on a Database whose revision
 * moved on is declined (CPU path) until the refreshed segment is registered — the revision key of SURVEY §8(f).1.
 * Registering a shard switches the index's micro-batching queue on (xgm_index_set_batching): Xapiand's worker threads issue one
 * get_mset each (src/manager.cc:161), and single-query calls of concurrent threads then share launches.
 */
#ifndef XGM_MATCHER_HOOK_H
#define XGM_MATCHER_HOOK_H

#include <cstdint>
#include <functional>
#include <string>
#include <utility>
#include <vector>

#include "xapian/database.h"
#include "xapian/enquire.h"
#include "xapian/intrusive_ptr.h"
#include "xapian/keymaker.h"
#include "xapian/matchspy.h"
#include "xapian/mset.h"
#include "xapian/query.h"
#include "xapian/weight.h"

#include "xgm.h"

namespace Xapian { namespace Internal { class PostList; } }

namespace xgm_hook {

/* registry (thread-safe).  batch: the index's micro-batching queue (0 = leave it as it is).
 * register_shard: the caller keeps ownership — it closes the index after unregister_shard, and only when no search can still be inside the
 * library (tests, single-threaded drivers).  register_shard_owned: the HOOK owns the index — every search that picks the shard up shares the
 * ownership for its call, so replacing the registration (a newer revision) or unregistering never frees an index under a reader; when the last
 * holder lets go the index is closed (xgm_index_close) and released() runs (remove the segment file, count).  This is what a server does
 * (xgm_xapiand_glue.cc): Xapiand's readers are Shard objects of the DatabasePool that search while the writer commits. */
void register_shard(const Xapian::Database& db, xgm_index* idx, uint32_t batch = 256);
void register_shard_owned(const Xapian::Database& db, xgm_index* idx, uint32_t batch, std::function<void()> released);
void register_shard_owned(const std::string& uuid, uint64_t revision, xgm_index* idx, uint32_t batch, std::function<void()> released);
void unregister_shard(const Xapian::Database& db);

/* switches (process-wide) */
void set_enabled(bool on);                 /* default: on */

/* PHRASE / NEAR with maxitems < matches: the reference's SelectPostList serves a stale cached weight
 * (src/xapian/matcher/selectpostlist.cc:28-55) and its top-k is then not a prefix of its own full ranking; the device returns
 * the intended top-k.  The choice is the deployment's and has to be made:
 *   POSITIONAL_DECLINE   — positional queries stay on the CPU matcher (byte-compatible by construction);
 *   POSITIONAL_INTENDED  — answered on the device with the intended semantics;
 *   POSITIONAL_REFERENCE — the reference's own page AND on the device (round 6: inside the batch every other search rides in): the plan carries
 *     XGM_REPLAY_BATCH_FROZEN — the query's units list their first matches in docid order, one wave replays ProtoMSet over them: true weights until
 *     min_weight turns positive, then the FROZEN weight: that of the first document of the underlying conjunction after that point (vet() weighs
 *     before test_doc()), served for every later match (xgm_andw_list_kernel + xgm_frozen_finish_kernel; shapes the listing kernel does not take —
 *     more than 4 terms, pages beyond 64 — are replayed one query at a time over the whole match, xgm_search_replay).  Ranks, docids, weights,
 *     max_attained are the reference's; its match-count figures too when set_exact_bounds(true) (XGM_REPLAY_BATCH_COUNT: every document of the
 *     conjunction is then tested — slower), else the looser bounds of every other shape.
 * Until set_positional_mode has been called positional queries are declined. */
enum PositionalMode { POSITIONAL_DECLINE = 0, POSITIONAL_INTENDED = 1, POSITIONAL_REFERENCE = 2 };
void set_positional_mode(PositionalMode m);
inline void set_decline_positional(bool on) { set_positional_mode(on ? POSITIONAL_DECLINE : POSITIONAL_INTENDED); }

/* OP_NEAR on an index that puts DISTINCT terms at the SAME position of a document (prefixed and unprefixed forms of a word, ...): the
 * reference's NearPostList::test_doc wants one distinct position per term and steps over duplicates (matcher/nearpostlist.cc:106-140);
 * the device's wave-parallel predicate assumes heads never coincide.  A deployment whose indexer co-locates terms says so here and every
 * registered shard (now and later) runs the reference's procedure in full for OP_NEAR (xgm_index_set_near_colocated, round 4), each
 * document from QUERY order.  The reference itself carries the order of its term vector from one tested document to the next (it sorts
 * the member in place), so with coinciding heads its answer depends on what was tested before — NEAR(a b) and NEAR(b a) differ, a top-k
 * search can differ from the full one (the oracle reproduces it and is pinned on it).  Hence: POSITIONAL_INTENDED answers
 * such queries on the device (the stateless reading), POSITIONAL_REFERENCE — the byte-compatible mode — keeps them on the CPU matcher
 * (rounds 2-3: always).  OP_PHRASE is unaffected: its terms sit at increasing positions by definition.  Default: false. */
void set_near_colocated_terms(bool may_exist);

/* Enquire::set_collapse_key.  The reference snapshot's collapser does not keep the best collapse_max documents of a key when
 * the proto-MSet overflows or collapse_max > 1 (matcher/collapser.cc:59-76, protomset.h:310-317); the device does.
 *   COLLAPSE_DECLINE (default) — collapsed searches stay on the CPU matcher;
 *   COLLAPSE_INTENDED — answered on the device: per key the best collapse_max documents under the ranking in force. */
enum CollapseMode { COLLAPSE_DECLINE = 0, COLLAPSE_INTENDED = 1, COLLAPSE_REFERENCE = 2 };
void set_collapse_mode(CollapseMode m);

/* REPLAY — the reference's own collation over the device's match.  Dynamic pruning in the posting-list tree only ever skips documents
 * whose weight is below ProtoMSet's min_weight, and the matcher's loop drops exactly those too (`if (weight < min_weight) continue`,
 * matcher.cc:500-505): the documents that reach ProtoMSet — and with them the collapser, the spies, the cut-offs, known_matching_docs —
 * are the same whether the loop walks the tree or the plain list of ALL matching documents in docid order with their weights.  So for
 * a search the device path cannot answer with the reference's exact semantics (COLLAPSE_REFERENCE: the snapshot's collapser, bugs and
 * all; with set_replay(true) also percentage / weight cut-offs and spies by relevance whose match exceeds check_at_least) the hook
 * fetches the whole match from the device in docid order (xgm_search_all: a match of ANY size since round 4; rounds 1-3 stopped at
 * one device page of XGM_MAX_K documents) and hands Matcher::get_local_mset a PostList that replays it (the second hunk of matcher_hook.patch): ProtoMSet, Collapser,
 * SpyMaster, the sorter run natively on it.  Static figures (termfreq bounds, max weight) are taken from the tree it replaces.
 * Not for positional queries unless POSITIONAL_INTENDED (the frozen weight lives in the tree), nor with a MatchDecider (its
 * counters see the tree's traversal). */
void set_replay(bool on);                  /* default: off */
/* ... for matches of up to this many documents (16 bytes each on the host; the replay's first device search only counts): a larger
 * match leaves the search to the CPU matcher.  Default: 4 000 000. */
void set_replay_limit(uint64_t max_matches);
/* Columns (value slots, KeyMaker keys) are built on the first sorted / spied / collapsed search of a shard revision, on that search's
 * thread, under the shard's column mutex (two passes over the value stream; 4 bytes per document + the distinct strings).  Shards
 * with more documents than this are declined (CPU matcher) instead of stalling their searches behind the scan; a server that wants
 * them on the device builds the columns when it registers the shard (run one sorted search then) and raises the limit.  A build that
 * fails is remembered: later searches decline at once.  Default: 50 000 000. */
void set_column_build_limit(uint32_t max_documents);
Xapian::Internal::PostList* maybe_replay(const Xapian::Database& db, const Xapian::Query& query, Xapian::Internal::PostList* pl);

/* MSet::get_matches_lower_bound / _estimated by relevance.  The reference derives them from known_matching_docs — how many
 * documents reached ProtoMSet::add, i.e. passed `weight >= min_weight` in the matcher's loop (matcher.cc:500-505), where
 * min_weight is the k-th best weight so far once check_at_least documents have been seen.  For the operators that visit every
 * match whatever min_weight is (a term, AND, FILTER, AND_NOT, PHRASE, NEAR: MultiAndPostList / SelectPostList / AndNotPostList
 * ignore w_min) that number is a function of the match in docid order (xgm_known_matching_docs); OR and AND_MAYBE skip
 * documents by weight inside the posting-list tree (orpostlist.cc:35-204)
 * — but only documents the matcher's loop would drop anyway, so there too the number is that function of the whole match.
 * With exact bounds ON the hook, for a full page, lets the device count as ProtoMSet would over the whole match in docid order
 * (a second device search, xgm_search_replay with XGM_REPLAY_COUNT: the match never leaves HBM, one number comes back) and reports
 * the reference's own figures for every shape (rounds 3-4: a download of the match + the host function / the reference's loop over a
 * ReplayPostList); when OFF (the default) the number of documents returned stands in (valid bounds, possibly looser). */
void set_exact_bounds(bool on);

/* A MatchSpy class the hook does not know natively: the server tells it which value slot the spy counts and how to hand it the
 * counts of a finished search (total = matching documents, counts = (value, documents) in ascending value order — exactly what the
 * spy would have tallied had it been shown every matching document).  Xapian::ValueCountMatchSpy needs no adapter. */
struct SpyAdapter {
    std::function<bool(const Xapian::MatchSpy&, Xapian::valueno* slot)> slot_of;
    std::function<void(Xapian::MatchSpy&, Xapian::doccount total, const std::vector<std::pair<std::string, Xapian::doccount>>& counts)> feed;
};
void register_spy_adapter(const std::string& spy_class_name, SpyAdapter adapter);

struct Counters { uint64_t answered, declined_shape, declined_unregistered, declined_revision, declined_device, answered_sorted, answered_spied,
                  answered_collapsed, columns_built, replayed,
                  combined,            /* sorted / spied / collapsed searches that went out in a launch shared with other threads' */
                  combined_launches; };
Counters counters();

/* The call the patch adds.  sort_by: Enquire::Internal::sort_setting as an int.  Returns true and fills `out` when the
 * search ran on the device. */
bool try_get_mset(const Xapian::Database& db, const Xapian::Query& query, const Xapian::Weight::Internal& stats,
                  const Xapian::Weight& wtscheme, bool full_db_has_positions, Xapian::doccount first,
                  Xapian::doccount maxitems, Xapian::doccount check_at_least, const Xapian::MatchDecider* mdecider,
                  const Xapian::KeyMaker* sorter, Xapian::valueno collapse_key, Xapian::doccount collapse_max, int percent_threshold,
                  double weight_threshold, Xapian::Enquire::docid_order order, Xapian::valueno sort_key, int sort_by,
                  bool sort_val_reverse, double time_limit,
                  const std::vector<Xapian::Internal::opt_intrusive_ptr<Xapian::MatchSpy>>& matchspies, Xapian::MSet& out);

}  // namespace xgm_hook

#endif
