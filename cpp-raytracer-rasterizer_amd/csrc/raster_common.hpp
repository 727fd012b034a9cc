// raster_common.hpp -- parameter blocks and scratch buffers of the rasteriser kernels
// (rasteriser/Source/rasteriser.cpp:461-482, 532-768).
#pragma once

#include "mirt_math.hpp"
#include "../../include/mirt.h"

namespace mirt {

// Screen coordinates of a projected vertex beyond this are outside the contract: the reference itself
// would size std::vectors from them (UB / bad_alloc).  Such triangles are skipped (oracle does the same).
constexpr int RASTER_COORD_LIMIT = 1 << 20;

// What VertexShader leaves per triangle, plus the row bookkeeping derived from it.
struct TriSetup {
    int x[3], y[3];          // Pixel::x, Pixel::y of the three projected vertices (rasteriser.cpp:544-545)
    float zinv[3];           // Pixel::zinv (:541)
    float p[3][3];           // Pixel::pos3d (:538)
    int minY, maxY;          // ComputePolygonRows :679-680
    int r0;                  // first row of this triangle inside the band [y0, y1)
    int rows;                // rows inside the band (0 when culled / outside / out of contract)
};

// One edge's sample at one row, as Interpolate emits it (rasteriser.cpp:624-636): 5 fields per slot.
// Slots of a row are stored field-major: slot[(row*3 + edge)*5 + field], field = {x, zinv, p.x, p.y, p.z}.
constexpr int SLOT_FIELDS = 5;

// The span DrawLineSDL/Bresenham walk for one (triangle,row) (rasteriser.cpp:592-612, 639-672).
struct Span {
    int ax, dx;              // a.x and b.x - a.x; pixels drawn are x = ax+1 .. ax+dx
    float azinv, zstep;      // a.zinv, (b.zinv - a.zinv)/float(dx)          (:648)
    float ap[3], pstep[3];   // a.pos3d, (b.pos3d - a.pos3d)/float(dx)       (:649)
    int tri, y;
};
static_assert(sizeof(Span) == 48, "span record must be 48 bytes");

// A span as the small-scene kernel (k_raster_small) keeps it in LDS: the depth-test and shading constants of one (triangle, row).
constexpr int SMALL_MAX_TRIS = 64;
struct SmallSpan {
    int ax, dx;                                     // Span::ax, Span::dx
    float azinv, zstep;
    float ap[3], pstep[3];
    float nrm[3], col[3];                           // triangles[tri].normal (NOT normalised by the rasteriser, :578) and .color
    int tri;
    int safe;                                       // every pixel of the span has pos3d and zinv inside the range of div3p_sel (mirt_math2.hpp)
    int pad[2];
};
static_assert(sizeof(SmallSpan) == 80, "row-list record must be 80 bytes (16-byte aligned)");

struct RasterScratch {
    TriSetup *setup = nullptr;       // n
    uint32_t *row_base = nullptr;    // n + 1 (exclusive scan of TriSetup::rows), then block sums
    uint32_t *block_sums = nullptr;
    float *slots = nullptr;          // cap_rows * 3 * SLOT_FIELDS
    Span *spans = nullptr;           // cap_rows
    unsigned long long *keys = nullptr;   // band pixels
    uint32_t *counters = nullptr;    // [0] total rows, [1] overflow flag, [2] rows of the tallest triangle
    uint32_t max_rows = 0;           // counters[2] of the last sizing pass
    int cap_tris = 0;
    size_t cap_rows = 0;             // rows the slot table holds (and the span table, at least)
    size_t cap_spans = 0;            // rows the span table holds (>= cap_rows; the worst case n x band rows for small scenes)
    size_t cap_px = 0;
    size_t keys_zero_px = 0;         // keys[0 .. keys_zero_px) are known to be zero (k_raster_resolve re-zeroes what a frame used)
    // sizing cache: the total row count of the previous frame with the same inputs
    uint64_t sizing_key = 0;
    bool sizing_valid = false;
};

struct RasterFrame {
    const float *tris15;
    const uint8_t *culled;
    int n;
    float cam[3];
    float rot[9];
    float invrot[9];
    float focal;
    int W, H;
    int nlights;
    float lpos[MIRT_MAX_LIGHTS][3];
    float lcol[MIRT_MAX_LIGHTS][3];
    int lights_in_range;     // every lcol component passes light_colour_in_range (mirt_math.hpp)
    float indirect[3];
    int y0, y1, row_origin;
    uint32_t *xrgb;
    int pitch_words;
    float *rgb;
    float *zinv;
    int32_t *index;
    float *fd;               // nullable: focalDistances = distance(pPos3d, cameraPos) - FOCAL_LENGTH of the owner fragment (rasteriser.cpp:563-565), 0 where nothing was drawn
    float focal_plane;       // FOCAL_LENGTH (:31)
    int edge_segments;       // k_raster_edges_lds: the edge chains as 64 predicted-and-checked segments each (one frame at a time: latency) or as one walk
    RasterScratch scratch;
};

// `cur` after n more of the reference's `current += step` (:632-635) WITHOUT doing them one by one -- a prediction; the caller
// checks it against the additions themselves (see k_raster_edges_lds).  While the sum stays inside one binade its values lie on
// that binade's grid (spacing u), and adding the same `step` to a grid point x rounds to x + q with one and the same multiple q
// of u: round-to-nearest moves x + step to the grid point nearest to it, which does not depend on x -- except when step falls
// exactly half-way between two grid points, where ties-to-even makes the FIRST such addition depend on x's parity and all later
// ones (x even by then) agree.  So: three real additions (all inside one binade: the last two's mantissas differ by q in units
// of u), then as many steps as keep the mantissa at least one unit inside the binade's ends at once, then real
// additions again across the boundary.  Zeros, subnormals, infinities and NaNs only ever see real additions.
__device__ __forceinline__ float edge_advance(float cur, float step, int n)
{
    while (n > 0) {
        const float c1 = cur + step;
        if (--n == 0) return c1;
        const float c2 = c1 + step;
        if (--n == 0) return c2;
        const float c3 = c2 + step;
        --n;
        cur = c3;
        // (all three on the binade's grid: c2 = c1 + step with c1 OFF the grid -- the addition that entered the binade -- may be odd,
        // and then a half-way step moves it by one unit more or less than it moves the even sums that follow)
        const uint32_t b1 = __float_as_uint(c1), b2 = __float_as_uint(c2), b3 = __float_as_uint(c3), ex = b3 & 0x7F800000u;
        if (n > 0 && (((b1 ^ b3) | (b2 ^ b3)) & 0xFF800000u) == 0u && ex != 0u && ex != 0x7F800000u) {
            const int m3 = (int)(b3 & 0x7FFFFFu), d = m3 - (int)(b2 & 0x7FFFFFu);
            int i = n;                                                  // (d == 0: the sum no longer moves)
            if (d > 0) i = (0x7FFFFE - m3) / d;
            else if (d < 0) i = (m3 - 1) / -d;
            i = max(min(i, n), 0);
            cur = __uint_as_float(b3 + (uint32_t)(i * d));
            n -= i;
        }
    }
    return cur;
}

int raster_scratch_ensure(RasterScratch &s, int n, int W, int band_rows);
void raster_scratch_free(RasterScratch &s);

}  // namespace mirt
