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
// behaviour); here such taps contribute nothing (documented divergence).
//
// The weights belong to the OUTPUT pixel and the sum is sequential, so partial sums cannot be shared between pixels
// (a running box sum would round differently): every output needs its own K*K multiply-adds.  What can be shared is
// the data: k_dof_tile stages a (32+K-1) x (64+K-1) tile of pixelColours in LDS; a thread owns eight
// vertically adjacent outputs and walks the tile rows once, feeding each value it reads to every output whose window
// holds that row -- each output still sees its taps in the reference's (z, z2) order.  HBM traffic is the
// algorithmic 16 B read + 4 B written per pixel: 136 MB fetched per 4K frame by the counters once the tiles of an XCD are
// neighbours (303 MB with tiles dealt row-major over the XCDs: every tile's halo came over the fabric); the K*K*6 VALU
// operations per pixel (no FMA: the reference multiplies, then adds) are 1472 packed + 145 scalar instructions per thread of
// 1912, 42 us at four cycles each.  69 us measured (74 before the XCD order, the incremental staging addresses and the
// straight-line stores of round 4): 0.185 vector instructions per clock and SIMD, which is what this instruction mix reaches
// here whatever the structure -- 4 or 8 outputs per thread, 4 or 8 waves per SIMD (68.6-69.5 us with DOF_WAVES = 8,
// DOF_PY = 4 at 63 VGPRs), 24 or 8 LDS reads per tap row, or every wave streaming its own rows through an LDS ring with no
// barrier at all (a variant built, measured and removed); tools/ubench.hip's packed multiply-add chains reach 0.207 at four
// waves per SIMD and 0.23 at eight.  Two more A/B builds of round 4: the tap rows never read from LDS (wrong pictures) 64 us --
// the reads cost 5 us --, and the same sums with one-lane v_mul_f32 / v_add_f32 (two cycles each when every operand is a vector
// register, profiles/r04_ubench.txt) 70 us for 55 M instructions instead of 31 M: packed or not, the pipes deliver ~51 T
// multiplies-or-adds per second here of the ~70 T the probes reach.
#include "dof.hpp"

#include <utility>
#include "mirt_math.hpp"

namespace mirt {

namespace {

constexpr int DOF_TX = 64;            // outputs per tile row = one wavefront
#ifndef MIRT_DOF_WAVES
#define MIRT_DOF_WAVES 4
#define MIRT_DOF_PY 8
#endif
constexpr int DOF_WAVES = MIRT_DOF_WAVES;
constexpr int DOF_PY = MIRT_DOF_PY;   // vertically adjacent outputs per thread
constexpr int DOF_TY = DOF_WAVES * DOF_PY;
constexpr int DOF_MAX_TILE_K = 16;
typedef float f2 __attribute__((ext_vector_type(2)));
static_assert(DOF_PY % 4 == 0, "outputs are paired for the packed arithmetic and fenced four at a time");    // larger kernels take k_dof_direct

__device__ __forceinline__ int dof_zlo(int K) { return (int)ceilf((float)K / -2.0f); }
__device__ __forceinline__ int dof_zhi(int K) { return (int)ceilf((float)K / 2.0f); }

// Scheduling fence: everything that feeds the accumulators happens before it, no LDS read moves above it.
__device__ __forceinline__ void dof_fence(v3 *fin, const float *row)
{
#pragma unroll
    for (int p = 0; p < DOF_PY; p += 2)
        asm volatile("" : "+v"(fin[p].x), "+v"(fin[p].y), "+v"(fin[p].z), "+v"(fin[p + 1].x), "+v"(fin[p + 1].y), "+v"(fin[p + 1].z)
                     : "v"(row) : "memory");
}

__device__ __forceinline__ void dof_fence2(f2 *fxy, f2 *fz, const float *row)
{
#pragma unroll
    for (int p = 0; p < DOF_PY; p += 4)
        asm volatile("" : "+v"(fxy[p]), "+v"(fxy[p + 1]), "+v"(fxy[p + 2]), "+v"(fxy[p + 3]), "+v"(fz[p / 2]), "+v"(fz[p / 2 + 1])
                     : "v"(row) : "memory");
}

// One tile row: z = zlo + (RR - p) for output p.  Everything that depends on (RR, c, p) is resolved at compile time.
// fxy: (r, g) of each output; fz: b of outputs (2q, 2q+1); wo2: their off-centre weights; cxy / cz: the K taps of the current
// tile row ((r, g) pairs and blues); nxy / nz: where the next row's taps are read to (the two register sets swap roles from row
// to row: copying next into current cost 24 moves per row, a sixth of the kernel's instructions).  The tile is planar per row
// -- TC (r, g) pairs, then TC blues -- so a thread's K taps are K consecutive 8-byte words and K consecutive floats: 4
// ds_read2_b64 + 4 ds_read2_b32 per row instead of 24 ds_read_b32, landing in the register pairs the packed arithmetic wants.
template <int RR, int KT>
__device__ __forceinline__ void dof_row(f2 (&fxy)[DOF_PY], f2 (&fz)[DOF_PY / 2], const f2 (&wo2)[DOF_PY / 2], const float (&wc)[DOF_PY],
                                        const float (&wo)[DOF_PY], f2 (&cxy)[KT], float (&cz)[KT], f2 (&nxy)[KT], float (&nz)[KT],
                                        const float *t0, int pitch, int boff)
{
    constexpr int ZC = KT / 2;                        // -ceil(KT / -2.0f): index of the centre tap
    if constexpr (RR + 1 < DOF_PY + KT - 1) {
        // the reads of row RR+1 are issued before the arithmetic of row RR (the fences on both sides keep them here)
        // (one address register per row and plane, the taps at small offsets from it: left alone, the compiler derives every
        // read's address from t0 with an add of its own once the offset no longer fits the instruction)
        typedef const __attribute__((address_space(3))) float lds_float;
        typedef const __attribute__((address_space(3))) f2 lds_f2;
        lds_float *tn = (lds_float *)(t0 + (RR + 1) * pitch), *tb = tn + boff;
        asm volatile("" : "+v"(tn), "+v"(tb));
#pragma unroll
        for (int c = 0; c < KT; c++) nxy[c] = *(lds_f2 *)(tn + 2 * c);
#pragma unroll
        for (int c = 0; c < KT; c++) nz[c] = tb[c];
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int c = 0; c < KT; c++) {
        const f2 vxy = cxy[c];
        const float vz = cz[c];
#pragma unroll
        for (int p = 0; p < DOF_PY; p++) {
            const int zi = RR - p;
            if (zi < 0 || zi >= KT) continue;
            const float w = (zi == ZC && c == ZC) ? wc[p] : wo[p];
            fxy[p] = fxy[p] + vxy * (f2){ w, w };                         // finalColour += colour * weighting (r, g)
        }
#pragma unroll
        for (int q = 0; q < DOF_PY / 2; q++) {
            const int z0 = RR - 2 * q, z1 = z0 - 1;
            const bool a0 = z0 >= 0 && z0 < KT, a1 = z1 >= 0 && z1 < KT;
            const bool c0 = z0 == ZC && c == ZC, c1 = z1 == ZC && c == ZC;
            if (a0 && a1) {
                const f2 w = (c0 || c1) ? (f2){ c0 ? wc[2 * q] : wo[2 * q], c1 ? wc[2 * q + 1] : wo[2 * q + 1] } : wo2[q];
                fz[q] = fz[q] + (f2){ vz, vz } * w;                       // b of outputs 2q and 2q+1
            } else if (a0) {
                fz[q].x = fz[q].x + vz * (c0 ? wc[2 * q] : wo[2 * q]);
            } else if (a1) {
                fz[q].y = fz[q].y + vz * (c1 ? wc[2 * q + 1] : wo[2 * q + 1]);
            }
        }
    }
    dof_fence2(fxy, fz, t0);
}

template <int KT, int... RR>
__device__ __forceinline__ void dof_rows(f2 (&fxy)[DOF_PY], f2 (&fz)[DOF_PY / 2], const f2 (&wo2)[DOF_PY / 2], const float (&wc)[DOF_PY],
                                         const float (&wo)[DOF_PY], f2 (&exy)[KT], float (&ez)[KT], f2 (&oxy)[KT], float (&oz)[KT],
                                         const float *t0, int pitch, int boff, std::integer_sequence<int, RR...>)
{
    (dof_row<RR, KT>(fxy, fz, wo2, wc, wo, (RR & 1) ? oxy : exy, (RR & 1) ? oz : ez, (RR & 1) ? exy : oxy, (RR & 1) ? ez : oz, t0, pitch, boff), ...);
}

// One pixelColours element by flat index, 0 outside the frame or outside the rows this call rendered.
__device__ __forceinline__ float dof_fetch(const DofFrame &f, long long flat_px, int ch)
{
    if (flat_px < 0 || flat_px >= (long long)f.W * f.H) return 0.0f;
    const int row = (int)(flat_px / f.W);
    if (row < f.ry0 || row >= f.ry1) return 0.0f;     // cannot happen for a halo of reach rows; kept as a guard
    return f.rgb[3 * flat_px + ch];
}

// KT > 0: kernel size known at compile time (loops unrolled); KT == 0: f.K at run time (<= DOF_MAX_TILE_K).
template <int KT>
__global__ __launch_bounds__(DOF_TX * DOF_WAVES) void k_dof_tile(const DofFrame f)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];   // KT == 0: [TR][TC][3], pixelColours as it lies in memory; KT > 0: planar rows, see below
    const int K = KT > 0 ? KT : f.K;
    const int zlo = dof_zlo(K);
    const int TR = DOF_TY + K - 1;                    // tile rows
    const int TC = DOF_TX + K - 1;                    // tile columns
    const int pitch = KT > 0 ? ((TC * 3 + 1) & ~1) : TC * 3;   // floats per tile row (even for the planar layout: its (r, g) pairs are read as 8-byte words)
    // Workgroup -> tile.  Consecutive workgroup ids go to consecutive XCDs, each with an L2 of its own: with tiles dealt in
    // row-major order every neighbour of a tile ran on another XCD and each tile's halo (7 of 39 rows, 7 of 71 columns, rounded
    // out to whole cache lines) came over the fabric once per tile.  So XCD g (= id % 8) takes the g-th eighth of the tiles in
    // row-major order, id / 8 counting through it: neighbours share an L2 and the halo is fetched once per XCD.
    int bx, by;
    {
        const int tiles_x = (f.W + DOF_TX - 1) / DOF_TX, tiles = tiles_x * ((f.y1 - f.y0 + DOF_TY - 1) / DOF_TY);
#ifndef MIRT_DOF_ROW_MAJOR
        const int per = (tiles + 7) >> 3, t = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
        if (t >= tiles) return;                          // (the grid is 8 * per workgroups; a whole workgroup leaves together)
#else
        const int t = blockIdx.x;
        if (t >= tiles) return;
#endif
        by = t / tiles_x; bx = t - by * tiles_x;
    }
    const int x0 = bx * DOF_TX, ty0 = f.y0 + by * DOF_TY;

    // stage: tile(r, c) = pixelColours[(ty0 + zlo + r) * W + (x0 + zlo + c)] by FLAT index, as the reference addresses
    // it (a column outside the row wraps into the neighbouring row); 0 outside the frame / the rows rendered.
    // Every tile row is one contiguous run of pixels, one wave per row.
    const long long lo = 3LL * max(0, f.ry0) * f.W, hi = 3LL * min(f.H, f.ry1) * f.W;
    if constexpr (KT > 0) {
        // The frame's rendered rows [lo, hi) as a range-checked buffer: a load whose byte offset falls outside it returns 0 --
        // the very rule the taps follow -- so staging needs no comparison and no branch per element (the per-element tests
        // were 64-bit compares under exec masks: 600 scalar and 500 vector instructions per wave, a fifth of the kernel).
        // Offsets are 32-bit: an element before `lo` wraps to ~4e9 and is out of range like one beyond `hi` (launch_dof sends
        // frames of 2^31 bytes or more to k_dof_tile<0>).  A lane loads whole pixels (three floats, each range-checked by itself)
        // and files them planar: row r holds TC (r, g) pairs, then TC blues.
        // All loads of the thread first, then all LDS stores: one round trip to memory instead of one per element.
        constexpr int ROWS = (DOF_TY + KT - 1 + DOF_WAVES - 1) / DOF_WAVES, COLS = (DOF_TX + KT - 1 + DOF_TX - 1) / DOF_TX;
        typedef float f3 __attribute__((ext_vector_type(3)));
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(f.rgb + lo), 0, (int)((hi - lo) * 4), 0x00020000);
        f3 v[ROWS][COLS];
        // (one 64-bit product for the thread's first row; the rows below it are DOF_WAVES frame rows further on, the column groups 64
        // pixels: 32-bit additions of a uniform stride and a constant -- the sums wrap exactly as the offsets themselves would)
        const uint32_t off0 = (uint32_t)((3 * ((long long)(ty0 + zlo + (int)threadIdx.y) * f.W + (x0 + zlo)) - lo) * 4) + threadIdx.x * 12u;
        const uint32_t row_step = (uint32_t)(DOF_WAVES * 12) * (uint32_t)f.W;
#pragma unroll
        for (int j = 0; j < ROWS; j++) {
            const uint32_t row_off = off0 + (uint32_t)j * row_step;
#pragma unroll
            for (int k = 0; k < COLS; k++)
                v[j][k] = __builtin_bit_cast(f3, __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)(row_off + (uint32_t)(k * DOF_TX * 12)), 0, 0));
        }
        // (every load is issued here, before the first store: a load sunk into a guarded store would wait for memory alone)
#pragma unroll
        for (int j = 0; j < ROWS; j++)
#pragma unroll
            for (int k = 0; k < COLS; k++) asm volatile("" : "+v"(v[j][k]));
        // the thread's first (r, g) pair and first blue; everything else sits at compile-time distances from them
        float *const pxy = tile + threadIdx.y * pitch + 2 * threadIdx.x, *const pz = tile + threadIdx.y * pitch + 2 * TC + threadIdx.x;
#pragma unroll
        for (int j = 0; j < ROWS; j++) {
            const int r = threadIdx.y + j * DOF_WAVES;
#pragma unroll
            for (int k = 0; k < COLS; k++) {
                const int c = threadIdx.x + k * DOF_TX;
                // (only the last row group and the last column group can fall outside the tile: the rest is unconditional)
                if (((j + 1) * DOF_WAVES <= DOF_TY + KT - 1 || r < TR) && ((k + 1) * DOF_TX <= DOF_TX + KT - 1 || c < TC)) {
                    *reinterpret_cast<f2 *>(pxy + j * DOF_WAVES * pitch + 2 * k * DOF_TX) = (f2){ v[j][k].x, v[j][k].y };
                    pz[j * DOF_WAVES * pitch + k * DOF_TX] = v[j][k].z;
                }
            }
        }
    } else {
        for (int r = threadIdx.y; r < TR; r += DOF_WAVES) {
            const long long src = 3 * ((long long)(ty0 + zlo + r) * f.W + (x0 + zlo));
            for (int i = threadIdx.x; i < pitch; i += DOF_TX) {
                const long long s = src + i;
                tile[r * pitch + i] = (s >= lo && s < hi) ? f.rgb[s] : 0.0f;
            }
        }
    }
    __syncthreads();

    const int x = x0 + threadIdx.x;
    const int yb = ty0 + threadIdx.y * DOF_PY;        // first of this thread's rows
    if (x >= f.W) return;
    const float totalPixels = (float)(K * K);                                 // :615
    float wc[DOF_PY], wo[DOF_PY];
    v3 fin[DOF_PY];
#pragma unroll
    for (int p = 0; p < DOF_PY; p++) {
        const int y = yb + p;
        float fdc = 0.0f;
        if constexpr (KT > 0) {
            // (range-checked like the colours: rows beyond the band read 0)
            const __amdgpu_buffer_rsrc_t fdsrc = __builtin_amdgcn_make_buffer_rsrc((void *)f.fd, 0, (int)((long long)f.y1 * f.W * 4), 0x00020000);
            fdc = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(fdsrc, (y * f.W + x) * 4, 0, 0));
        } else if (y < f.y1) fdc = f.fd[(size_t)y * f.W + x];
        const float a = fminf(fabsf(fdc), 1.0f);                              // min(abs(focalDistances[..]), 1.0f)
        wc[p] = 1 - (a * ((totalPixels - 1) / totalPixels));                  // :629
        wo[p] = a * (1.0f / totalPixels);                                     // :631
        fin[p] = V3(0.0f, 0.0f, 0.0f);
    }
    // KT == 0: lanes read with a stride of 3 floats: odd, so the 64 lanes of a read fall into 64 different banks
    const float *t0 = tile + (threadIdx.y * DOF_PY) * pitch + threadIdx.x * (KT > 0 ? 2 : 3);
    if constexpr (KT > 0) {
        // Software pipeline over tile rows: the reads of row rr+1 are issued (volatile: they stay where they are
        // written) before the arithmetic of row rr, which covers the LDS latency.  Without this the scheduler either
        // hoists every read (280 VGPRs, 1 wave per SIMD) or sinks each read to just before its use and stalls on it.
        // The arithmetic is packed FP32 (v_pk_mul_f32 / v_pk_add_f32: two IEEE multiplies or adds per issue slot, the
        // same roundings as the scalar forms): red and green of one output share an instruction, and the blues of two
        // vertically adjacent outputs share one.
        f2 fxy[DOF_PY], fz[DOF_PY / 2], wo2[DOF_PY / 2];
#pragma unroll
        for (int p = 0; p < DOF_PY; p++) fxy[p] = (f2){ 0.0f, 0.0f };
#pragma unroll
        for (int q = 0; q < DOF_PY / 2; q++) { fz[q] = (f2){ 0.0f, 0.0f }; wo2[q] = (f2){ wo[2 * q], wo[2 * q + 1] }; }
        f2 exy[KT], oxy[KT];                              // the taps of even / odd tile rows
        float ez[KT], oz[KT];
        const int boff = 2 * TC - threadIdx.x;            // from the lane's first (r, g) pair to its first blue
#pragma unroll
        for (int c = 0; c < KT; c++) { exy[c] = *reinterpret_cast<const f2 *>(t0 + 2 * c); ez[c] = t0[boff + c]; }
        dof_rows<KT>(fxy, fz, wo2, wc, wo, exy, ez, oxy, oz, t0, pitch, boff, std::make_integer_sequence<int, DOF_PY + KT - 1>());
#pragma unroll
        for (int p = 0; p < DOF_PY; p++) fin[p] = V3(fxy[p].x, fxy[p].y, (p & 1) ? fz[p / 2].y : fz[p / 2].x);
    } else {
        for (int rr = 0; rr < DOF_PY + K - 1; rr++) {
            const float *tr = t0 + rr * pitch;
            for (int c = 0; c < K; c++) {
                const v3 v = V3(tr[3 * c], tr[3 * c + 1], tr[3 * c + 2]);
#pragma unroll
                for (int p = 0; p < DOF_PY; p++) {
                    const int zi = rr - p;
                    if (zi < 0 || zi >= K) continue;
                    const float w = (zi == -zlo && c == -zlo) ? wc[p] : wo[p];
                    fin[p] = add3(fin[p], scale3(v, w));
                }
            }
        }
    }
    if constexpr (KT > 0) {
        // Straight-line stores: the surface's rows up to y1 as a range-checked buffer -- a store whose offset lies beyond it is dropped --,
        // and a pixel that is not to be written (beyond the band; a border pixel of a surface that keeps its border) gets such an
        // offset.  (With a branch per output the eight stores cost 300 scalar instructions of exec-mask bookkeeping per wave.)
        const long long span = (long long)(f.y1 - f.row_origin) * f.pitch_words * 4;      // < 2^31 (launch_dof)
        const __amdgpu_buffer_rsrc_t osrc = __builtin_amdgcn_make_buffer_rsrc((void *)f.xrgb, 0, (int)span, 0x00020000);
        const bool xin = x >= 1 && x < f.W - 1;
        const uint32_t off0 = ((uint32_t)(yb - f.row_origin) * (uint32_t)f.pitch_words + (uint32_t)x) * 4u, step = (uint32_t)f.pitch_words * 4u;
#pragma unroll
        for (int p = 0; p < DOF_PY; p++) {
            const int y = yb + p;
            const bool in = xin && y >= 1 && y < f.H - 1;                                 // interior only (:618-620)
            const uint32_t word = in ? pack_xrgb(fin[p]) : 0u;                            // PutPixelSDL (:646); a cleared border otherwise
            const bool put = y < f.y1 && (in || f.clear_border);
            __builtin_amdgcn_raw_buffer_store_b32(word, osrc, (int)(put ? off0 + (uint32_t)p * step : 0xFFFFFFFFu), 0, 0);
        }
    } else {
#pragma unroll
        for (int p = 0; p < DOF_PY; p++) {
            const int y = yb + p;
            if (y >= f.y1) break;
            uint32_t *out = f.xrgb + (size_t)(y - f.row_origin) * f.pitch_words + x;
            if (x >= 1 && x < f.W - 1 && y >= 1 && y < f.H - 1) *out = pack_xrgb(fin[p]);   // interior only (:618-620), PutPixelSDL (:646)
            else if (f.clear_border) *out = 0u;
        }
    }
}

// Any kernel size: one thread per pixel, taps through L1/L2.
__global__ __launch_bounds__(256) void k_dof_direct(const DofFrame f)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = f.y0 + blockIdx.y;
    if (x >= f.W) return;
    if (!(x >= 1 && x < f.W - 1 && y >= 1 && y < f.H - 1)) {                  // interior pixels only (:618-620)
        if (f.clear_border) f.xrgb[(size_t)(y - f.row_origin) * f.pitch_words + x] = 0u;
        return;
    }
    const float totalPixels = (float)(f.K * f.K);                             // :615
    const int zlo = dof_zlo(f.K), zhi = dof_zhi(f.K);
    const float fdc = f.fd[(size_t)y * f.W + x];
    const float a = fminf(fabsf(fdc), 1.0f);
    const float w_centre = 1 - (a * ((totalPixels - 1) / totalPixels));       // :629
    const float w_other = a * (1.0f / totalPixels);                           // :631
    v3 fin = V3(0.0f, 0.0f, 0.0f);
    for (int z = zlo; z < zhi; z++) {
        for (int z2 = zlo; z2 < zhi; z2++) {
            const float w = (z == 0 && z2 == 0) ? w_centre : w_other;
            const long long idx = (long long)(y + z) * f.W + (x + z2);        // flat index, as in the reference (:634)
            const v3 c = V3(dof_fetch(f, idx, 0), dof_fetch(f, idx, 1), dof_fetch(f, idx, 2));
            fin = add3(fin, scale3(c, w));
        }
    }
    f.xrgb[(size_t)(y - f.row_origin) * f.pitch_words + x] = pack_xrgb(fin);  // PutPixelSDL (:646)
}

size_t dof_tile_lds_bytes(int K)
{
    return (size_t)(DOF_TY + K - 1) * (((DOF_TX + K - 1) * 3 + 1) & ~1) * sizeof(float);      // (rows padded to an even number of floats)
}

}  // namespace

void launch_dof(const DofFrame &d, hipStream_t stream)
{
    const int rows = d.y1 - d.y0;
    if (rows <= 0 || d.W <= 0) return;
    if (d.K <= DOF_MAX_TILE_K) {
        const int tiles = ((d.W + DOF_TX - 1) / DOF_TX) * ((rows + DOF_TY - 1) / DOF_TY);
        const dim3 grid(((tiles + 7) / 8) * 8), block(DOF_TX, DOF_WAVES);      // (eight runs of tiles, one per XCD: see k_dof_tile)
        // (k_dof_tile<8> addresses the frame and the surface with 32-bit byte offsets)
        if (d.K == 8 && 12LL * d.W * d.H < (1LL << 31) && 4LL * (d.y1 - d.row_origin) * d.pitch_words < (1LL << 31) && d.row_origin <= d.y0) hipLaunchKernelGGL(k_dof_tile<8>, grid, block, dof_tile_lds_bytes(8), stream, d);
        else hipLaunchKernelGGL(k_dof_tile<0>, grid, block, dof_tile_lds_bytes(d.K), stream, d);
    } else {
        hipLaunchKernelGGL(k_dof_direct, dim3((d.W + 255) / 256, rows), dim3(256), 0, stream, d);
    }
}

}  // namespace mirt
