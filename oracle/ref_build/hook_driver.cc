/* xapian_hook — drop-in demonstration and parity check of seam B2: the REAL
 * vendored Xapian of the reference runs every query twice through its own Enquire::get_mset —
 * once with its CPU matcher, once with the query replaced by integration/GpuTopKPostingSource, i.e.
 * with libxgm.so behind the reference's PostingSource plug-in API — and the two MSets must agree
 * in docid and weight (bit pattern) at every rank.  Test infrastructure: built only where
 * /root/reference exists, into oracle/_ref/, and run by tests/test_gpu_hook.py on the GPU box.
 *
 *   xapian_hook <glass dbdir> <segment file> <queries.txt>
 * The segment must have been exported from that glass DB (xapian_ref export + xgm_segment_build_from_file).
 * Query file format: as xapian_ref (oracle/ref_build/ref_driver.cc).
 */
#include <xapian.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../integration/xgm_posting_source.h"

namespace {

struct QuerySpec {
    std::string op;
    unsigned first = 0, maxitems = 10, window = 0;
    std::vector<std::string> terms;
};

Xapian::Query cpu_query(const QuerySpec& q) {
    std::vector<Xapian::Query> subs;
    unsigned pos = 1;
    for (auto& t : q.terms) subs.emplace_back(t, 1, q.op == "PHRASE" ? pos++ : 0);
    if (q.op == "AND") return Xapian::Query(Xapian::Query::OP_AND, subs.begin(), subs.end());
    if (q.op == "OR") return Xapian::Query(Xapian::Query::OP_OR, subs.begin(), subs.end());
    return Xapian::Query(Xapian::Query::OP_PHRASE, subs.begin(), subs.end(), q.window ? q.window : (unsigned)q.terms.size());
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: xapian_hook <dbdir> <segment> <queries.txt>\n"); return 2; }
    try {
        Xapian::Database db(argv[1]);
        xgm_index* idx = nullptr;
        if (xgm_index_open(argv[2], 0, UINT64_MAX, &idx) != XGM_OK) { fprintf(stderr, "xgm_index_open: %s\n", xgm_last_error()); return 1; }
        std::ifstream in(argv[3]);
        std::string line;
        unsigned n = 0, bad = 0, declined = 0;
        while (std::getline(in, line)) {
            if (line.empty() || line[0] == '#') continue;
            std::istringstream ss(line);
            QuerySpec q;
            ss >> q.op >> q.first >> q.maxitems >> q.window;
            std::string t;
            while (ss >> t) q.terms.push_back(t);
            Xapian::Enquire cpu(db);
            cpu.set_query(cpu_query(q));
            Xapian::MSet want = cpu.get_mset(q.first, q.maxitems);

            const uint32_t op = q.op == "AND" ? XGM_OP_AND : q.op == "OR" ? XGM_OP_OR : XGM_OP_PHRASE;
            /* eligibility first: a declined shape keeps the original Xapian::Query (and the CPU matcher) */
            auto* src = GpuTopKPostingSource::create(idx, op, q.terms, q.first + q.maxitems, q.window);
            Xapian::Enquire gpu(db);
            if (src) gpu.set_query(Xapian::Query(src->release())); else { gpu.set_query(cpu_query(q)); ++declined; }
            Xapian::MSet got = gpu.get_mset(q.first, q.maxitems);
            ++n;
            bool ok = want.size() == got.size();
            auto a = want.begin();
            auto b = got.begin();
            for (; ok && a != want.end(); ++a, ++b) {
                const double wa = a.get_weight(), wb = b.get_weight();
                ok = *a == *b && memcmp(&wa, &wb, sizeof wa) == 0;
            }
            if (!ok) {
                ++bad;
                printf("MISMATCH %s: cpu %u hits, gpu %u hits\n", line.c_str(), want.size(), got.size());
            }
        }
        printf("{\"queries\": %u, \"mismatches\": %u, \"declined\": %u}\n", n, bad, declined);
        xgm_index_close(idx);
        return bad ? 1 : 0;
    } catch (const Xapian::Error& e) {
        fprintf(stderr, "Xapian error: %s\n", e.get_description().c_str());
        return 1;
    }
}
