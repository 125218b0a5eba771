/* GPU segment builder for the synthetic corpus — placeholder until the builder lands. */
#include "xgm_internal.h"
extern "C" int xgm_index_build_synthetic(const xgm_synth_params*, int, xgm_index** out) {
    if (out) *out = nullptr;
    return xgm_set_error(XGM_E_INVALID, "synthetic builder not built yet");
}
