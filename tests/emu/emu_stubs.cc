/* TEST INFRASTRUCTURE (tests/emu): what the emulated build of the library leaves out. */
#include "../../include/xgm.h"

int xgm_set_error(int code, const char* fmt, ...);

/* the device corpus builder sorts with hipCUB: not emulated — emulation tests build their segments from files */
extern "C" int xgm_index_build_synthetic(const xgm_synth_params*, int, xgm_index**) {
    return xgm_set_error(XGM_E_INVALID, "xgm_index_build_synthetic is not part of the emulated build");
}

/* xgm_search_all's ordering step is rocPRIM's device radix sort (xgm_all.hip): here "device" memory is host memory, a host sort stands in */
#include <algorithm>
#include <cstring>
#include <vector>
#include "xgm_launch.h"
size_t xgm_all_sort_temp_bytes(size_t) { return 16; }
int xgm_all_sort_pack(void*, size_t, unsigned long long* keys, unsigned long long*, unsigned long long* vals, unsigned long long*, size_t n,
                      xgm_hit* out, hipStream_t) {
    std::vector<size_t> ix(n);
    for (size_t i = 0; i < n; ++i) ix[i] = i;
    std::stable_sort(ix.begin(), ix.end(), [&](size_t a, size_t b) { return (keys[a] >> 32) < (keys[b] >> 32); });
    for (size_t i = 0; i < n; ++i) {
        out[i].docid = (uint32_t)(keys[ix[i]] >> 32);
        out[i].subqs_matched = (uint32_t)keys[ix[i]];
        memcpy(&out[i].weight, &vals[ix[i]], 8);
    }
    return 0;
}
