/* Link-time stand-ins for the three symbols of the vendored Xapian that live in source files we do
 * not compile into the oracle because they depend on generated code (lemon query parser, snowball
 * stemmers — SURVEY.md §8(c)).  None of them is on the match/rank path.  Test infrastructure only. */
#include "config.h"
#include <xapian.h>
#include "xapian/api/msetinternal.h"

namespace Xapian {

/* reference: src/xapian/queryparser/queryparser.cc:59 (empty destructor). */
RangeProcessor::~RangeProcessor() {}

/* reference: src/xapian/queryparser/termgenerator_internal.cc:745 — snippets need the stemmers. */
std::string MSet::Internal::snippet(const std::string&, size_t, const Xapian::Stem&, unsigned,
                                    const std::string&, const std::string&, const std::string&) const {
    throw Xapian::UnimplementedError("snippet() is not built into the oracle");
}

}  // namespace Xapian
