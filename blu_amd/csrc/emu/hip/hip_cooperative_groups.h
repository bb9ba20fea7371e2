// emu stand-in: cooperative (grid-wide) launches are not emulated; hipLaunchCooperativeKernel reports
// "not supported" and the library falls back to its one-workgroup kernels, so sync() is never reached.
#pragma once
#include "hip_runtime.h"
namespace cooperative_groups {
struct grid_group {
    void sync() const
    {
        fprintf(stderr, "emu: grid sync reached (cooperative launches are not emulated)\n");
        abort();
    }
};
inline grid_group this_grid() { return grid_group(); }
} // namespace cooperative_groups
