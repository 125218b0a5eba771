/* See xgm_aggregation_adapter.h. */
#include "xgm_aggregation_adapter.h"

#include <memory>
#include <stdexcept>
#include <string>

#include "aggregations/aggregations.h"      /* Xapiand's: AggregationMatchSpy */
#include "msgpack.h"
#include "reserved/aggregations.h"
#include "serialise_list.h"

#include "xgm_matcher_hook.h"

namespace xgm_xapiand {
namespace {

enum Kind { K_BUCKET_OR_METRIC_OK, K_SUM_LIKE, K_DECLINED, K_NOT_A_TYPE };

/* the aggregation types of aggregations.cc:57-85, by what the adapter can do with them */
Kind kind_of(std::string_view key) {
    static const char* const plain[] = {RESERVED_AGGS_COUNT, RESERVED_AGGS_MIN, RESERVED_AGGS_MAX, RESERVED_AGGS_VALUES, RESERVED_AGGS_VALUE, RESERVED_AGGS_HISTOGRAM,
                                        RESERVED_AGGS_RANGE};
    static const char* const sums[] = {RESERVED_AGGS_SUM, RESERVED_AGGS_AVG, RESERVED_AGGS_VARIANCE, RESERVED_AGGS_STD, RESERVED_AGGS_STATS, RESERVED_AGGS_EXT_STATS};
    static const char* const no[] = {RESERVED_AGGS_MEDIAN, RESERVED_AGGS_MODE, RESERVED_AGGS_FILTER, RESERVED_AGGS_TERMS, RESERVED_AGGS_TERM, RESERVED_AGGS_CARDINALITY,
                                     RESERVED_AGGS_GEO_BOUNDS, RESERVED_AGGS_GEO_CENTROID, RESERVED_AGGS_PERCENTILES, RESERVED_AGGS_PERCENTILES_RANK,
                                     RESERVED_AGGS_SCRIPTED_METRIC, RESERVED_AGGS_DATE_HISTOGRAM, RESERVED_AGGS_DATE_RANGE, RESERVED_AGGS_GEO_DISTANCE,
                                     RESERVED_AGGS_GEO_TRIXELS, RESERVED_AGGS_MISSING, RESERVED_AGGS_IP_RANGE, RESERVED_AGGS_GEO_IP};
    for (const char* k : plain) if (key == k) return K_BUCKET_OR_METRIC_OK;
    for (const char* k : sums) if (key == k) return K_SUM_LIKE;
    for (const char* k : no) if (key == k) return K_DECLINED;
    return K_NOT_A_TYPE;
}

/* Walks the object under `_aggs` / `_aggregations`: every named aggregation is an object holding its type (→ its configuration with `_field`) and,
 * beside it, optional sub-aggregations.  Collects the ONE field everything reads; false = a shape the adapter does not take. */
bool walk(const MsgPack& aggs, std::string* field, bool* sum_like, int depth) {
    if (!aggs.is_map() || depth > 8) return false;
    const auto it_end = aggs.end();
    for (auto it = aggs.begin(); it != it_end; ++it) {
        const MsgPack& agg = it.value();
        if (!agg.is_map()) return false;
        bool typed = false;
        const auto jt_end = agg.end();
        for (auto jt = agg.begin(); jt != jt_end; ++jt) {
            const auto key = jt->str_view();
            if (key == RESERVED_AGGS_AGGS || key == RESERVED_AGGS_AGGREGATIONS) {
                if (!walk(jt.value(), field, sum_like, depth + 1)) return false;
                continue;
            }
            const Kind k = kind_of(key);
            if (k == K_DECLINED) return false;
            if (k == K_NOT_A_TYPE) continue;                      /* (a parameter beside the type: `_sort`, `_limit`, ...) */
            if (typed) return false;                               /* (one type per named aggregation) */
            typed = true;
            if (k == K_SUM_LIKE) *sum_like = true;
            const MsgPack& conf = jt.value();
            if (!conf.is_map()) return false;
            const auto ft = conf.find(RESERVED_AGGS_FIELD);
            if (ft == conf.end() || !ft.value().is_string()) return false;
            const std::string f(ft.value().str_view());
            if (field->empty()) *field = f; else if (*field != f) return false;       /* (two fields: their JOINT distribution is not what the device counted) */
        }
        if (!typed) return false;
    }
    return true;
}

}  // namespace

bool aggregation_slot_of(const Xapian::MatchSpy& spy, const FieldLookup& lookup, Xapian::valueno* slot) {
    const auto* a = dynamic_cast<const AggregationMatchSpy*>(&spy);
    if (!a || !lookup) return false;
    try {
        const std::string ser = a->serialise();                   /* StringList{_aggs as msgpack, the schema} (aggregations.cc: AggregationMatchSpy::serialise) */
        StringList data(ser);
        if (data.size() != 2) return false;
        const MsgPack conf = MsgPack::unserialise(*data.begin());
        auto it = conf.find(RESERVED_AGGS_AGGS);
        if (it == conf.end()) it = conf.find(RESERVED_AGGS_AGGREGATIONS);
        if (it == conf.end()) return false;
        std::string field;
        bool sum_like = false;
        if (!walk(it.value(), &field, &sum_like, 0) || field.empty()) return false;
        bool integral = false;
        if (!lookup(field, slot, &integral)) return false;
        return integral || !sum_like;                              /* (floating sums: the order of addition shows in the last bits — the CPU matcher's order is the reference) */
    } catch (...) {
        return false;
    }
}

void aggregation_feed(Xapian::MatchSpy& spy, Xapian::valueno slot, Xapian::doccount total, const std::vector<std::pair<std::string, Xapian::doccount>>& counts) {
    auto* a = dynamic_cast<AggregationMatchSpy*>(&spy);
    if (!a) throw std::logic_error("xgm_aggregation_adapter: not an AggregationMatchSpy");
    /* the result of `count` documents that all carry `value` (empty: none): one document through the reference's own operator(), merged by doubling */
    auto add = [&](const std::string* value, Xapian::doccount count) {
        std::unique_ptr<AggregationMatchSpy> pow(static_cast<AggregationMatchSpy*>(a->clone()));
        Xapian::Document doc;
        if (value) doc.add_value(slot, *value);
        (*pow)(doc, 0.0);
        for (Xapian::doccount n = count; n; n >>= 1) {
            if (n & 1u) a->merge_results(*pow);
            if (n > 1u) {
                std::unique_ptr<AggregationMatchSpy> twice(static_cast<AggregationMatchSpy*>(a->clone()));
                twice->merge_results(*pow);
                twice->merge_results(*pow);
                pow = std::move(twice);
            }
        }
    };
    Xapian::doccount with_value = 0;
    for (const auto& vc : counts) { if (vc.second) add(&vc.first, vc.second); with_value += vc.second; }
    if (total > with_value) add(nullptr, total - with_value);      /* matching documents without a value: counted, in no bucket */
}

void register_aggregation_adapter(FieldLookup lookup) {
    xgm_hook::SpyAdapter ad;
    ad.slot_of = [lookup](const Xapian::MatchSpy& s, Xapian::valueno* slot) { return aggregation_slot_of(s, lookup, slot); };
    ad.feed = [lookup](Xapian::MatchSpy& s, Xapian::doccount total, const std::vector<std::pair<std::string, Xapian::doccount>>& counts) {
        Xapian::valueno slot = 0;
        if (!aggregation_slot_of(s, lookup, &slot)) throw std::logic_error("xgm_aggregation_adapter: not an aggregation the adapter takes");
        aggregation_feed(s, slot, total, counts);
    };
    xgm_hook::register_spy_adapter("AggregationMatchSpy", ad);
}

}  // namespace xgm_xapiand
