// dof_kernel.hip -- the depth-of-field branch of CalculateDOF (raytracer.cpp:613-640, rasteriser.cpp:494-513), the
// step AFTER the per-pixel path, shared by both renderers: a KxK blur of pixelColours whose weights come from the
// centre pixel's focalDistances, followed by PutPixelSDL.
//
//   for z, z2 in [ceil(K / -2.0f), ceil(K / 2.0f)):
//       weighting = (z == 0 && z2 == 0) ? 1 - min(|fd|, 1) * ((K*K - 1) / (K*K)) : min(|fd|, 1) * (1 / (K*K))
//       finalColour += pixelColours[(y+z)*stride + (x+z2)] * weighting            (sequential float accumulation)
//
// The tap address is a FLAT index, exactly as in the reference: columns outside the row wrap into the neighbouring
// rows.  Flat indices outside the frame read whatever lies next to the reference's global array (undefined
// behaviour); here such taps contribute nothing (documented divergence).  One thread per pixel;
// neighbouring lanes share their taps through L1/L2, the plane is read once from HBM.
#include "mirt_math.hpp"

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

__global__ __launch_bounds__(256) void k_dof(const DofFrame f)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = f.y0 + blockIdx.y;
    if (x >= f.W) return;
    if (!(x >= 1 && x < f.W - 1 && y >= 1 && y < f.H - 1)) {                  // interior pixels only (:618-620)
        if (f.clear_border) f.xrgb[(size_t)(y - f.row_origin) * f.pitch_words + x] = 0u;
        return;
    }
    const float totalPixels = (float)(f.K * f.K);                             // :615
    const int zlo = (int)ceilf((float)f.K / -2.0f), zhi = (int)ceilf((float)f.K / 2.0f);
    const float fdc = f.fd[(size_t)y * f.W + x];
    const float a = fminf(fabsf(fdc), 1.0f);                                  // min(abs(focalDistances[..]), 1.0f)
    const float w_centre = 1 - (a * ((totalPixels - 1) / totalPixels));       // :629
    const float w_other = a * (1.0f / totalPixels);                           // :631
    const long long npx = (long long)f.W * f.H;
    v3 fin = V3(0.0f, 0.0f, 0.0f);
    for (int z = zlo; z < zhi; z++) {
        for (int z2 = zlo; z2 < zhi; z2++) {
            const float w = (z == 0 && z2 == 0) ? w_centre : w_other;
            const long long idx = (long long)(y + z) * f.W + (x + z2);        // flat index, as in the reference (:634)
            v3 c = V3(0.0f, 0.0f, 0.0f);
            if (idx >= 0 && idx < npx) {
                const int row = (int)(idx / f.W);                             // within the rendered band + halo by construction
                if (row >= f.ry0 && row < f.ry1) c = ld3(f.rgb + 3 * idx);
            }
            fin = add3(fin, scale3(c, w));                                    // finalColour += colour * weighting
        }
    }
    f.xrgb[(size_t)(y - f.row_origin) * f.pitch_words + x] = pack_xrgb(fin);  // PutPixelSDL (:646)
}

}  // namespace mirt
