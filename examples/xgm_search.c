/* Minimal C client of the C ABI (include/xgm.h): open a segment, plan one query against the shard's own
 * statistics, run it on the device, print the MSet.  Plain C99 — this file is also the proof that the header
 * is a C header (tests/test_abi.py compiles it with `gcc -std=c99 -pedantic`).
 *
 *   cc -std=c99 -Iinclude examples/xgm_search.c -Lxapiand_amd/csrc -lxgm -Wl,-rpath,$PWD/xapiand_amd/csrc -o xgm_search
 *   ./xgm_search shard.seg AND t3 t17 t120          (or OR / PHRASE; AND_NOT:2 t3 t17 t9 = (t3 AND t17) AND_NOT t9)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "xgm.h"

int main(int argc, char** argv) {
    xgm_index* idx = NULL;
    xgm_query_desc d;
    xgm_query q;
    xgm_hit hits[10];
    xgm_result_hdr h;
    uint32_t i;
    int rc;
    char op[32];
    char* colon;

    if (argc < 4) { fprintf(stderr, "usage: %s <segment> <AND|OR|PHRASE|AND_NOT:n|AND_MAYBE:n|FILTER:n> term...\n", argv[0]); return 2; }
    memset(&d, 0, sizeof d);
    strncpy(op, argv[2], sizeof op - 1);
    op[sizeof op - 1] = '\0';
    colon = strchr(op, ':');
    if (colon) { d.n_required = (uint32_t)strtoul(colon + 1, NULL, 10); *colon = '\0'; }
    d.op = !strcmp(op, "AND") ? XGM_OP_AND : !strcmp(op, "OR") ? XGM_OP_OR : !strcmp(op, "PHRASE") ? XGM_OP_PHRASE :
           !strcmp(op, "AND_NOT") ? XGM_OP_AND_NOT : !strcmp(op, "AND_MAYBE") ? XGM_OP_AND_MAYBE : !strcmp(op, "FILTER") ? XGM_OP_FILTER : 0;
    d.n_terms = (uint32_t)(argc - 3);
    if (!d.op || d.n_terms > XGM_MAX_TERMS) { fprintf(stderr, "bad operator or too many terms\n"); return 2; }
    for (i = 0; i < d.n_terms; ++i) { d.terms[i] = argv[3 + i]; d.term_len[i] = (uint32_t)strlen(argv[3 + i]); }
    d.first = 0; d.maxitems = 10; d.check_at_least = 0;
    d.k1 = 1; d.k2 = 0; d.k3 = 1; d.b = 0.5; d.min_normlen = 0.5;          /* Xapian::BM25Weight defaults */

    rc = xgm_index_open(argv[1], 0, UINT64_MAX, &idx);
    if (rc) { fprintf(stderr, "xgm_index_open: %s\n", xgm_last_error()); return 1; }
    rc = xgm_plan_query(idx, &d, NULL, &q);
    if (rc > 0) { fprintf(stderr, "query shape not handled by the device path: use the CPU matcher\n"); xgm_index_close(idx); return 3; }
    if (rc == 0) rc = xgm_search(idx, &q, hits, &h);
    if (rc) { fprintf(stderr, "search failed: %s\n", xgm_last_error()); xgm_index_close(idx); return 1; }
    printf("%llu matches%s, max_possible %.17g\n", (unsigned long long)XGM_MATCHES_COUNT(h.matches_exact),
           (h.matches_exact & XGM_MATCHES_LOWER_BOUND) ? " (at least)" : "", h.max_possible);
    for (i = q.first; i < h.n_hits; ++i) printf("%2u  docid %-10u weight %.17g\n", i, hits[i].docid, hits[i].weight);
    xgm_index_close(idx);
    return 0;
}
