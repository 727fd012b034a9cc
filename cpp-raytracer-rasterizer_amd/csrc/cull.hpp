// cull.hpp -- the cull step of the rasteriser's Update() (rasteriser.cpp:385-447, InCuboid :451-458), shared by the host
// entry point mirt_cull (scene_host.cpp) and the device kernel behind mirt_cull_device (raster_kernels.hip): the same
// float operations in the same order on both sides.
#pragma once

#include "mirt_math.hpp"
#include "../../include/mirt.h"

namespace mirt {

struct CullParams {
    float cam[3];
    float rot[9];
    float tr[16];        // the reference's `transform` (glm::mat4, column-major: tr[c*4 + r]), :397-402
    int flags;           // bit0 = BACKFACE_CULLING_ENABLED, bit1 = FRUSTUM_CULLING_ENABLED
};

void cull_setup(const mirt_view *view, int flags, CullParams *cp);

// isCulled of one triangle (:404-447)
MIRT_HD uint8_t cull_one(const float *p, const CullParams &cp)
{
    const v3 cam = ld3(cp.cam);
    int c = 0;
    if (cp.flags & 1)
        if (dot3(sub3(ld3(p), cam), ld3(p + 9)) > 0.0f) c = 1;                      // :408-414
    if ((cp.flags & 2) && !c) {
        bool inside[3];
        for (int k = 0; k < 3; k++) {
            const v3 q = vec_mul_mat3(sub3(ld3(p + 3 * k), cam), cp.rot);          // :423-425
            const float v[4] = { q.x, q.y, q.z, 1.0f };
            float o[4];
            for (int j = 0; j < 4; j++)      // vec4 * mat4, raytracer/glm/detail/type_mat4x4.inl:664-675
                o[j] = cp.tr[j * 4 + 0] * v[0] + cp.tr[j * 4 + 1] * v[1] + cp.tr[j * 4 + 2] * v[2] + cp.tr[j * 4 + 3] * v[3];
            const float X = o[0] / o[3], Y = o[1] / o[3], Z = o[2] / o[3];          // :435-437
            inside[k] = X >= -1.0f && X <= 1.0f && Y >= -1.0f && Y <= 1.0f && Z >= 0.0f && Z <= 1.0f;
        }
        if (!inside[0] && !inside[1] && !inside[2]) c = 1;                          // :444-445
    }
    return (uint8_t)c;
}

}  // namespace mirt
