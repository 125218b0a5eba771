/* TEST INFRASTRUCTURE (tests/emu): what the emulated build of the library leaves out. */
#include "../../include/xgm.h"

int xgm_set_error(int code, const char* fmt, ...);

/* the device corpus builder sorts with hipCUB: not emulated — emulation tests build their segments from files */
extern "C" int xgm_index_build_synthetic(const xgm_synth_params*, int, xgm_index**) {
    return xgm_set_error(XGM_E_INVALID, "xgm_index_build_synthetic is not part of the emulated build");
}
