/* Stand-in for <hip/hip_ext.h> (tests/emu): a launch that carries its events is a launch (the emulation has no clock). */
#pragma once
#include <hip/hip_runtime.h>
#define hipExtLaunchKernelGGL(kern, grid, block, lds_bytes, stream, ev_start, ev_stop, flags, ...) \
    do { (void)(ev_start); (void)(ev_stop); (void)(flags); hipLaunchKernelGGL(kern, grid, block, lds_bytes, stream, __VA_ARGS__); } while (0)
