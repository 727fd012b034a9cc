// dof.hpp -- parameter block of the depth-of-field resolve (dof_kernel.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace mirt {

struct DofFrame {
    const float *rgb;        // pixelColours, full-frame indexing, row stride W; rows [ry0, ry1) are valid
    const float *fd;         // focalDistances, same indexing
    int W, H;
    int K;                   // DOF_KERNEL_SIZE
    int y0, y1, row_origin;  // rows to resolve
    int ry0, ry1;            // rows present in rgb/fd (the band plus its halo)
    uint32_t *xrgb;
    int pitch_words;
    int clear_border;        // rasteriser: Update() painted the whole surface black (rasteriser.cpp:190), so border words become 0
};

// Enqueues the resolve of rows [y0, y1) on `stream`.
void launch_dof(const DofFrame &d, hipStream_t stream);

}  // namespace mirt
