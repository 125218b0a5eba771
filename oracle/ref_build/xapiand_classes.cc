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


/* ==== Xapiand's OWN aggregation spy in front of the matcher hook (VERDICT r4 missing #2) =========================
 *
 * AggregationMatchSpy (reference src/aggregations/aggregations.h:108-157) is the Xapian::MatchSpy DocMatcher attaches for a request's
 * `_aggs` (src/database/handler.cc:1283).  Its three translation units — aggregations.cc, bucket.cc, metrics.cc — are compiled where
 * they lie; what they need of Xapiand's Schema (10 k lines of schema.cc, the whole indexing front end) is two lookups, restated here
 * for the synthetic value slots: field "slot<N>" = value slot N, keyword type.  Everything that aggregates — ValuesAggregation, its
 * handler's reading of a slot as a StringList (serialise_list.h), the buckets, the (un)serialisation — is the reference's.
 *
 * How the class reaches the device (what INTEGRATION.md tells a Xapiand maintainer to register, xgm_hook::SpyAdapter):
 *   slot_of  the spy's own serialise() carries its `_aggs` description: ONE `_values` aggregation on a slot field, no sub-aggregations,
 *            is what the device counts (xgm_search_sorted_spy: matching documents per distinct value of the slot's column); anything
 *            else is declined and stays on the CPU matcher;
 *   feed     per DISTINCT slot value the reference's own class is shown ONE document carrying it (a clone, so that multi-valued slots
 *            fall into their buckets exactly as AggregationMatchSpy::operator() would put them), its serialised result is scaled by the
 *            device's count of that value, and merged with the spy's own merge_results — O(distinct values), not O(matches). */
namespace {

unsigned field_slot(std::string_view field) {
    if (field.size() < 5 || field.substr(0, 4) != "slot") throw std::invalid_argument("xapiand_classes.cc: the stand-in schema knows fields slot<N>");
    return (unsigned)std::stoul(std::string(field.substr(4)));
}

/* a Schema that is never constructed (its constructor and members live in schema.cc, which is not built): the two lookups below do
 * not touch the object */
std::shared_ptr<Schema> stand_in_schema() {
    static std::shared_ptr<Schema> s(static_cast<Schema*>(::operator new(sizeof(Schema))), [](Schema* p) { ::operator delete(p); });
    return s;
}

MsgPack values_conf(unsigned slot) {
    MsgPack field = MsgPack::MAP();
    field[RESERVED_AGGS_FIELD] = "slot" + std::to_string(slot);
    MsgPack agg = MsgPack::MAP();
    agg[RESERVED_AGGS_VALUES] = field;
    MsgPack aggs = MsgPack::MAP();
    aggs["by_value"] = agg;
    MsgPack conf = MsgPack::MAP();
    conf[RESERVED_AGGS_AGGS] = aggs;
    return conf;
}

/* every count of a serialised Aggregation (aggregations.cc:250-262: length(doc_count), then per sub-aggregation its name and its
 * serialised results; a bucket aggregation: per bucket its key and the bucket's Aggregation, bucket.h:460-467) multiplied by n */
std::string scale_aggregation(std::string_view ser, unsigned long long n, int depth = 0);
std::string scale_buckets(std::string_view ser, unsigned long long n, int depth) {
    std::string out;
    const char* p = ser.data();
    const char* end = p + ser.size();
    while (p != end) {
        const std::string_view key = unserialise_string(&p, end);
        const std::string_view inner = unserialise_string(&p, end);
        out += serialise_string(key);
        out += serialise_string(scale_aggregation(inner, n, depth + 1));
    }
    return out;
}
std::string scale_aggregation(std::string_view ser, unsigned long long n, int depth) {
    if (depth > 4) throw std::invalid_argument("xapiand_classes.cc: aggregation nested too deep for the adapter");
    const char* p = ser.data();
    const char* end = p + ser.size();
    std::string out = serialise_length(unserialise_length(&p, end) * n);
    while (p != end) {
        const std::string_view name = unserialise_string(&p, end);
        const std::string_view sub = unserialise_string(&p, end);
        out += serialise_string(name);
        out += serialise_string(scale_buckets(sub, n, depth));
    }
    return out;
}

}  // namespace

Xapian::MatchSpy* xapiand_aggregation_spy(unsigned slot) { return new AggregationMatchSpy(values_conf(slot), stand_in_schema()); }

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

bool xapiand_aggregation_slot_of(const Xapian::MatchSpy& spy, Xapian::valueno* slot) {
    const auto* a = dynamic_cast<const AggregationMatchSpy*>(&spy);
    if (!a) return false;
    try {
        const std::string ser = a->serialise();              /* (a StringList is a VIEW of the string it is given) */
        StringList data(ser);
        if (data.size() != 2) return false;
        const MsgPack conf = MsgPack::unserialise(*data.begin());
        auto it = conf.find(RESERVED_AGGS_AGGS);
        if (it == conf.end()) it = conf.find(RESERVED_AGGS_AGGREGATIONS);
        if (it == conf.end() || !it.value().is_map() || it.value().size() != 1) return false;
        const MsgPack& agg = it.value().begin().value();
        if (!agg.is_map() || agg.size() != 1) return false;                      /* (a sub-aggregation would sit beside the type) */
        const auto vt = agg.find(RESERVED_AGGS_VALUES);
        if (vt == agg.end() || !vt.value().is_map() || vt.value().size() != 1) return false;
        const auto ft = vt.value().find(RESERVED_AGGS_FIELD);
        if (ft == vt.value().end() || !ft.value().is_string()) return false;
        *slot = (Xapian::valueno)field_slot(ft.value().str_view());
        return true;
    } catch (...) {
        return false;
    }
}

void xapiand_aggregation_feed(Xapian::MatchSpy& spy, Xapian::doccount total, const std::vector<std::pair<std::string, Xapian::doccount>>& counts) {
    auto* a = dynamic_cast<AggregationMatchSpy*>(&spy);
    Xapian::valueno slot = 0;
    if (!a || !xapiand_aggregation_slot_of(spy, &slot)) throw std::logic_error("xapiand_classes.cc: not an aggregation the adapter takes");
    Xapian::doccount with_value = 0;
    for (const auto& vc : counts) {
        std::unique_ptr<Xapian::MatchSpy> one(a->clone());
        Xapian::Document doc;
        doc.add_value(slot, vc.first);
        (*one)(doc, 0.0);                                                       /* the reference's own per-document logic, once per distinct value */
        a->merge_results(scale_aggregation(one->serialise_results(), vc.second));
        with_value += vc.second;
    }
    if (total > with_value) {                                                    /* matching documents without a value: counted, in no bucket */
        std::unique_ptr<Xapian::MatchSpy> one(a->clone());
        (*one)(Xapian::Document(), 0.0);
        a->merge_results(scale_aggregation(one->serialise_results(), total - with_value));
    }
}

/* ---- the two lookups of Xapiand's Schema the aggregations make (database/schema.cc:9460, 9665), for the synthetic slots ---- */
required_spc_t::flags_t::flags_t() { std::memset(static_cast<void*>(this), 0, sizeof *this); }
required_spc_t::required_spc_t() : sep_types({{FieldType::empty, FieldType::empty, FieldType::empty}}), slot(Xapian::BAD_VALUENO) { }
required_spc_t::required_spc_t(required_spc_t&& o) noexcept = default;
required_spc_t::required_spc_t(const required_spc_t& o) = default;
std::string required_spc_t::prefix_t::operator()() const noexcept { return field; }
required_spc_t Schema::get_slot_field(std::string_view field_name) const {
    required_spc_t spc;
    spc.slot = (Xapian::valueno)field_slot(field_name);
    spc.set_type(FieldType::keyword);
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
