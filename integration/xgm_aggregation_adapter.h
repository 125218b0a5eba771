/* xgm_aggregation_adapter — how Xapiand's OWN match spy, AggregationMatchSpy (reference src/aggregations/aggregations.h:107-157: what DocMatcher attaches for a
 * request's `_aggs`, src/database/handler.cc:1283), rides on the device path.  A PRODUCT component (round 6; round 5 kept a `_values`-only version inside the
 * test harness): compile it into Xapiand next to xgm_matcher_hook.cc and call
 *
 *     xgm_xapiand::register_aggregation_adapter([schema](std::string_view field, Xapian::valueno* slot, bool* integral) {
 *         auto spc = schema->get_slot_field(field);            // database/schema.cc:9460
 *         *slot = spc.slot;
 *         *integral = spc.get_type() != FieldType::floating;   // sums over a floating field depend on the order of addition: left to the CPU matcher
 *         return spc.slot != Xapian::BAD_VALUENO;
 *     });
 *
 * once at start-up.  What it takes: a request whose whole `_aggs` tree reads ONE value slot — any nesting of the aggregations that are functions of the slot's
 * value(s): `_values`, `_histogram`, `_range` (buckets, aggregations.cc:57-85 / bucket.h) and `_count`, `_sum`, `_avg`, `_min`, `_max`, `_variance`,
 * `_std_deviation`, `_stats`, `_extended_stats` (metrics.h), sub-aggregations included as long as they name the same field.  The device counts the matching documents per
 * DISTINCT value of the slot (xgm_search_sorted_spy: one pass); per distinct value the reference's own class is shown ONE document carrying it, and that
 * one-document result is merged count times by DOUBLING with the class's own merge_results (aggregations.h:152-153: what Xapiand merges its shards'
 * aggregations with) — O(distinct values x log count) merges instead of a virtual call per matching document, and every bucket / metric rule stays the
 * reference's.  Anything else — `_filter`, `_terms` / `_term`, `_median` / `_mode` (they keep every value), aggregations over two fields — is declined:
 * the search stays on the CPU matcher. */
#ifndef XGM_AGGREGATION_ADAPTER_H
#define XGM_AGGREGATION_ADAPTER_H

#include <functional>
#include <string_view>
#include <utility>
#include <vector>

#include "xapian.h"

namespace xgm_xapiand {

/* the server's schema lookup: field name → value slot; *integral = the field's values are integers (sums are then exact whatever the order) */
typedef std::function<bool(std::string_view field, Xapian::valueno* slot, bool* integral)> FieldLookup;

/* registers the adapter with the matcher hook (xgm_hook::register_spy_adapter("AggregationMatchSpy", ...)) */
void register_aggregation_adapter(FieldLookup lookup);

/* the two halves, exposed for tests: which slot the spy's whole `_aggs` tree reads (false: not a shape the adapter takes), and feeding the device's counts */
bool aggregation_slot_of(const Xapian::MatchSpy& spy, const FieldLookup& lookup, Xapian::valueno* slot);
void aggregation_feed(Xapian::MatchSpy& spy, Xapian::valueno slot, Xapian::doccount total, const std::vector<std::pair<std::string, Xapian::doccount>>& counts);

}  // namespace xgm_xapiand

#endif
