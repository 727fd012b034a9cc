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
    RasterScratch scratch;
};

int raster_scratch_ensure(RasterScratch &s, int n, int W, int band_rows);
void raster_scratch_free(RasterScratch &s);

}  // namespace mirt
