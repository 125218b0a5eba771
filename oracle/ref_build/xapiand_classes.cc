/* TEST INFRASTRUCTURE: Xapiand's OWN sort class in front of the matcher hook.
 *
 * Multi_MultiValueKeyMaker (reference src/multivalue/keymaker.h:366, keymaker.cc) is what DocMatcher hands to
 * Enquire::set_sort_by_key_then_relevance on most real searches (src/database/handler.cc:1125, 1269).  Its translation unit and the ones
 * it needs — length.cc, sortable_serialise.cc, exception.cc, geospatial/cartesian.cc, fmt/format.cc, repr.cc, io.cc — are compiled where
 * they lie under /root/reference by oracle/ref_build/Makefile (nothing is copied); this file hands the driver instances of the REAL class
 * (xapiand_keymaker) and stubs the five symbols those units reference that a sort never reaches (content types, the geo key's
 * serialisation, the logger) — the rest of Xapiand (schema, logger thread, msgpack) is not built.
 *
 * The hook recognises a key maker by name() + serialise() (xgm_hook::key_column) — "Multi_MultiValueKeyMaker" implements both — builds
 * the device column from the keys the class itself makes of every document, and the driver compares hook off vs hook on. */
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

#include "multivalue/keymaker.h"
#include "aggregations/aggregations.h"
#include "database/data.h"
#include "database/schema.h"
#include "length.h"
#include "logger_fwd.h"
#include "msgpack.h"
#include "reserved/aggregations.h"
#include "serialise.h"
#include "serialise_list.h"

/* ---- the real class, instantiated the way QueryDSL::get_sorter does (src/query_dsl.cc:1274-1509: one add_* per sort field) ---- */
const Xapian::KeyMaker* xapiand_keymaker(unsigned variant, bool reverse) {
    static std::map<unsigned, Multi_MultiValueKeyMaker*> made;        /* lives as long as any Enquire that was handed it */
    const unsigned key = variant * 2u + (reverse ? 1u : 0u);
    auto it = made.find(key);
    if (it == made.end()) {
        auto* km = new Multi_MultiValueKeyMaker();
        /* slot 3 holds Xapiand-style multi-values (a StringList of the document's values, ascending: serialise_list.h:318-327);
         * SerialiseKey picks the smallest (forward) or the largest (reverse) — then a second field in the other direction */
        km->add_serialise(3, reverse);
        if (variant % 2u == 1u) km->add_serialise(variant % 3u, !reverse);
        it = made.emplace(key, km).first;
    }
    return it->second;
}

/* ---- never reached by a sort; referenced by the translation units above ---- */
ct_type_t::ct_type_t(std::string_view) { }                                            /* (database/data.cc: the static content-type constants of data.h) */
CartesianList Unserialise::centroids(std::string_view) { throw std::logic_error("xapiand_classes.cc: geo keys are not part of this build"); }
Cartesian Unserialise::cartesian(std::string_view) { throw std::logic_error("xapiand_classes.cc: geo keys are not part of this build"); }
std::string Serialise::centroids(const std::vector<Cartesian>&) { throw std::logic_error("xapiand_classes.cc: geo keys are not part of this build"); }
Log::~Log() noexcept { }
Log vlog(bool, std::chrono::steady_clock::time_point, bool, bool, bool, uint64_t, int, std::exception_ptr&&, void**, const char*, const char*, int, std::string_view,
         fmt::format_args) {
    return Log();
}


/* ==== Xapiand's OWN aggregation spy in front of the matcher hook (VERDICT r4 missing #2, r5 #5) ====================
 *
 * AggregationMatchSpy (reference src/aggregations/aggregations.h:107-157) is the Xapian::MatchSpy DocMatcher attaches for a request's
 * `_aggs` (src/database/handler.cc:1283).  Its three translation units — aggregations.cc, bucket.cc, metrics.cc — are compiled where
 * they lie; what they need of Xapiand's Schema (10 k lines of schema.cc, the whole indexing front end) is two lookups, restated here
 * for the synthetic value slots: field "slot<N>" = value slot N as a keyword, field "num<N>" = value slot N as a positive integer
 * (sortable_serialise'd, as Xapiand's Serialise::positive stores numbers).  Everything that aggregates — the handlers' reading of a slot as a
 * StringList, buckets, metrics, the (un)serialisation, merge_results — is the reference's.
 *
 * The ADAPTER that lets the class ride on the device is a product component since round 6: integration/xgm_aggregation_adapter.{h,cc}
 * (registered by hook_b1_driver.cc with THIS file's stand-in lookup; a Xapiand build registers its Schema's).  What is left here is test
 * infrastructure: the stand-in schema and the `_aggs` descriptions the driver's queries ask for ("SPYA=<slot>:<kind>"). */
namespace {

struct StandInField { unsigned slot; bool numeric; };
StandInField stand_in_field(std::string_view field) {
    if (field.size() >= 5 && field.substr(0, 4) == "slot") return StandInField{(unsigned)std::stoul(std::string(field.substr(4))), false};
    if (field.size() >= 4 && field.substr(0, 3) == "num") return StandInField{(unsigned)std::stoul(std::string(field.substr(3))), true};
    throw std::invalid_argument("xapiand_classes.cc: the stand-in schema knows fields slot<N> and num<N>");
}

/* a Schema that is never constructed (its constructor and members live in schema.cc, which is not built): the two lookups below do
 * not touch the object */
std::shared_ptr<Schema> stand_in_schema() {
    static std::shared_ptr<Schema> s(static_cast<Schema*>(::operator new(sizeof(Schema))), [](Schema* p) { ::operator delete(p); });
    return s;
}

MsgPack on_field(const char* type, const std::string& field) {
    MsgPack conf = MsgPack::MAP();
    conf[RESERVED_AGGS_FIELD] = field;
    MsgPack agg = MsgPack::MAP();
    agg[type] = conf;
    return agg;
}

/* the `_aggs` of a request, by kind (tests/test_gpu_hook_b1.py::xapiand_aggregation_queries) */
MsgPack aggs_conf(unsigned slot, unsigned kind) {
    const std::string keyword = "slot" + std::to_string(slot), number = "num4";
    MsgPack aggs = MsgPack::MAP();
    switch (kind) {
        case 0: aggs["by_value"] = on_field(RESERVED_AGGS_VALUES, keyword); break;
        case 1: aggs["figures"] = on_field(RESERVED_AGGS_STATS, number); break;
        case 2:                                                            /* several metrics of one field side by side */
            aggs["n"] = on_field(RESERVED_AGGS_COUNT, number); aggs["total"] = on_field(RESERVED_AGGS_SUM, number); aggs["mean"] = on_field(RESERVED_AGGS_AVG, number);
            aggs["least"] = on_field(RESERVED_AGGS_MIN, number); aggs["most"] = on_field(RESERVED_AGGS_MAX, number);
            aggs["spread"] = on_field(RESERVED_AGGS_EXT_STATS, number); aggs["var"] = on_field(RESERVED_AGGS_VARIANCE, number);
            aggs["dev"] = on_field(RESERVED_AGGS_STD, number);
            break;
        case 3: {                                                          /* a histogram whose buckets carry a sub-aggregation of the same field */
            MsgPack h = on_field(RESERVED_AGGS_HISTOGRAM, number);
            h[RESERVED_AGGS_HISTOGRAM][RESERVED_AGGS_INTERVAL] = 100000;
            MsgPack sub = MsgPack::MAP();
            sub["top"] = on_field(RESERVED_AGGS_MAX, number);
            sub["sum"] = on_field(RESERVED_AGGS_SUM, number);
            h[RESERVED_AGGS_AGGS] = sub;
            aggs["per_100k"] = h;
            break;
        }
        case 4: {                                                          /* ranges */
            MsgPack r = on_field(RESERVED_AGGS_RANGE, number);
            MsgPack ranges = MsgPack::ARRAY();
            MsgPack r0 = MsgPack::MAP(); r0[RESERVED_AGGS_TO] = 250000;
            MsgPack r1 = MsgPack::MAP(); r1[RESERVED_AGGS_FROM] = 250000; r1[RESERVED_AGGS_TO] = 750000;
            MsgPack r2 = MsgPack::MAP(); r2[RESERVED_AGGS_FROM] = 750000;
            ranges.push_back(r0); ranges.push_back(r1); ranges.push_back(r2);
            r[RESERVED_AGGS_RANGE][RESERVED_AGGS_RANGES] = ranges;
            aggs["thirds"] = r;
            break;
        }
        case 5: {                                                          /* TWO fields (a keyword's buckets, a number's sum inside them): not the adapter's — the CPU matcher's */
            MsgPack v = on_field(RESERVED_AGGS_VALUES, keyword);
            MsgPack sub = MsgPack::MAP();
            sub["sum"] = on_field(RESERVED_AGGS_SUM, number);
            v[RESERVED_AGGS_AGGS] = sub;
            aggs["by_value"] = v;
            break;
        }
        case 6: aggs["middle"] = on_field(RESERVED_AGGS_MEDIAN, number); break;      /* keeps every value: the CPU matcher's */
        default: throw std::invalid_argument("xapiand_classes.cc: unknown aggregation kind");
    }
    MsgPack conf = MsgPack::MAP();
    conf[RESERVED_AGGS_AGGS] = aggs;
    return conf;
}

}  // namespace

Xapian::MatchSpy* xapiand_aggregation_spy(unsigned slot, unsigned kind) { return new AggregationMatchSpy(aggs_conf(slot, kind), stand_in_schema()); }

/* the stand-in for a Xapiand build's Schema lookup (integration/xgm_aggregation_adapter.h, FieldLookup) */
bool xapiand_stand_in_lookup(std::string_view field, Xapian::valueno* slot, bool* integral) {
    try {
        const StandInField f = stand_in_field(field);
        *slot = (Xapian::valueno)f.slot;
        *integral = true;                                                  /* (keywords and positive integers) */
        return true;
    } catch (...) {
        return false;
    }
}

/* what a response would carry (get_aggregation(): `_aggregations` → doc_count + buckets) followed by the wire form */
std::string xapiand_aggregation_result(Xapian::MatchSpy* spy) {
    auto* a = dynamic_cast<AggregationMatchSpy*>(spy);
    if (!a) return "";
    return a->get_aggregation().to_string() + "|" + a->serialise_results();
}

void xapiand_aggregation_merge(Xapian::MatchSpy* into, Xapian::MatchSpy* from) {
    auto* a = dynamic_cast<AggregationMatchSpy*>(into);
    auto* b = dynamic_cast<AggregationMatchSpy*>(from);
    if (a && b) a->merge_results(*b);
}

/* ---- the two lookups of Xapiand's Schema the aggregations make (database/schema.cc:9460, 9665), for the synthetic slots ---- */
required_spc_t::flags_t::flags_t() { std::memset(static_cast<void*>(this), 0, sizeof *this); }
required_spc_t::required_spc_t() : sep_types({{FieldType::empty, FieldType::empty, FieldType::empty}}), slot(Xapian::BAD_VALUENO) { }
required_spc_t::required_spc_t(required_spc_t&& o) noexcept = default;
required_spc_t::required_spc_t(const required_spc_t& o) = default;
std::string required_spc_t::prefix_t::operator()() const noexcept { return field; }
required_spc_t Schema::get_slot_field(std::string_view field_name) const {
    required_spc_t spc;
    const StandInField f = stand_in_field(field_name);
    spc.slot = (Xapian::valueno)f.slot;
    spc.set_type(f.numeric ? FieldType::positive : FieldType::keyword);
    return spc;
}
std::pair<required_spc_t, std::string> Schema::get_data_field(std::string_view, bool) const { throw std::logic_error("xapiand_classes.cc: term aggregations are not part of this build"); }
std::shared_ptr<const MsgPack> Schema::get_const_schema() const { static auto m = std::make_shared<const MsgPack>(MsgPack::MAP()); return m; }
std::string Serialise::MsgPack(const required_spc_t&, const class MsgPack&) { throw std::logic_error("xapiand_classes.cc: range / filter aggregations are not part of this build"); }
/* (reached only by aggregation shapes this build's adapter declines: typed slots other than keyword, remote spies) */
Schema::Schema(std::shared_ptr<const MsgPack>, std::unique_ptr<MsgPack>, std::string) { throw std::logic_error("xapiand_classes.cc: Xapiand's Schema is not part of this build"); }
std::string Unserialise::uuid(std::string_view, UUIDRepr) { throw std::logic_error("xapiand_classes.cc: uuid slots are not part of this build"); }
double Unserialise::timedelta_d(std::string_view) { throw std::logic_error("xapiand_classes.cc: timedelta slots are not part of this build"); }
double Unserialise::time_d(std::string_view) { throw std::logic_error("xapiand_classes.cc: time slots are not part of this build"); }
range_t Unserialise::range(std::string_view) { throw std::logic_error("xapiand_classes.cc: geo slots are not part of this build"); }
specification_t::specification_t() { }          /* (a member of Schema, whose constructor above only throws) */
