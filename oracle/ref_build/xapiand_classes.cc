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
#include <map>
#include <stdexcept>
#include <string>
#include <string_view>

#include "multivalue/keymaker.h"
#include "database/data.h"
#include "logger_fwd.h"
#include "serialise.h"

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
