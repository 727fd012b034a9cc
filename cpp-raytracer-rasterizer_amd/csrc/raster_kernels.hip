// raster_kernels.hip -- hand-written gfx950 kernels for the rasteriser's Draw() loop
// (rasteriser/Source/rasteriser.cpp: Draw :461-482, DrawPolygon :755-768, VertexShader :532-546,
//  ComputePolygonRows :674-735 + Interpolate :615-637, DrawRows :738-753, DrawLineSDL :592-612,
//  Bresenham :639-672, PixelShader :549-589, CalculateDOF :484-529, clear in Update :183-192).
//
// Pipeline per frame (all on one stream, no host round trip once the row count is known):
//   k_raster_vertex   one thread per triangle: VertexShader x3, row range, rows inside the band
//   k_scan_*          exclusive scan of rows-per-triangle -> row_base (where each triangle's rows live)
//                     (small scenes: both in one single-workgroup launch, k_raster_vertex_scan)
//   k_raster_edges    one workgroup per triangle: Interpolate's SEQUENTIAL float accumulation, bit for bit
//                     (a + k*step would round differently), 15 chains staged through LDS into per-row slots; then
//                     per row strict-min / strict-max x over the <=3 edge samples -> the span (a.x, b.x] and its
//                     zinv / pos3d steps
//   k_raster_frag     one wave per span: zinv = a.zinv + zstep*float(i) per fragment and an atomic z-compare:
//                     atomicMax on the u64 key  zinv_bits<<32 | (0xFFFFFFFF - tri)  (zinv > 0 orders as
//                     unsigned bits; the reference's strict `>` with in-order triangles = max zinv, lowest
//                     index among exact ties) -- race-free and deterministic, unlike the reference's OpenMP loop
//   k_raster_resolve  one thread per pixel: decode the winner, rebuild its pos3d, PixelShader, and write the
//                     XRGB word (+ optional float colour / depth / index planes) with coalesced stores
#include "raster_common.hpp"
#include "mirt_math2.hpp"
#include "scan.hpp"

#include <limits.h>
#include <algorithm>
#include <cstdlib>

#include "cull.hpp"

namespace mirt {

// ---- VertexShader (rasteriser.cpp:532-546) ------------------------------------------------------------
__device__ __forceinline__ TriSetup vertex_setup(const RasterFrame &f, int t)
{
    const float *tri = f.tris15 + (size_t)15 * t;
    const v3 cam = ld3(f.cam);
    TriSetup s;
    bool ok = f.culled[t] == 0;                                        // Draw() :470
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const v3 pos = vec_mul_mat3(sub3(ld3(tri + 3 * k), cam), f.rot);   // (v - cameraPos) * cameraRot
        const v3 p3 = div3s(pos, pos.z);                                    // pos / pos.z
        const float zi = 1.0f / pos.z;
        s.p[k][0] = p3.x; s.p[k][1] = p3.y; s.p[k][2] = p3.z;
        s.zinv[k] = zi;
        s.x[k] = f2i_x86((float)f2i_x86(f.focal * (pos.x * zi)) + ((float)f.W / 2.0f));
        s.y[k] = f2i_x86((float)f2i_x86(f.focal * (pos.y * zi)) + ((float)f.H / 2.0f));
        if (s.x[k] <= -RASTER_COORD_LIMIT || s.x[k] >= RASTER_COORD_LIMIT ||
            s.y[k] <= -RASTER_COORD_LIMIT || s.y[k] >= RASTER_COORD_LIMIT) ok = false;
    }
    s.maxY = max(max(s.y[0], s.y[1]), s.y[2]);
    s.minY = min(min(s.y[0], s.y[1]), s.y[2]);
    const int lo = max(s.minY, f.y0), hi = min(s.maxY, f.y1 - 1);
    s.r0 = lo;
    s.rows = (ok && hi >= lo) ? hi - lo + 1 : 0;
    return s;
}

__global__ __launch_bounds__(256) void k_raster_vertex(const RasterFrame f)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= f.n) return;
    const TriSetup s = vertex_setup(f, t);
    f.scratch.setup[t] = s;
    f.scratch.row_base[t] = (uint32_t)s.rows;     // scanned in place by k_scan_*
    if (s.rows > 0) atomicMax(&f.scratch.counters[2], (uint32_t)s.rows);      // the tallest triangle (zeroed by the host per sizing pass)
}

// ---- exclusive scan of row counts (3 passes, 1024 items per block) -----------------------------------
__device__ __forceinline__ uint32_t block_exclusive_scan_256x4(uint32_t v[4], uint32_t *total)
{
    __shared__ uint32_t s_wave[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t mine = v[0] + v[1] + v[2] + v[3];
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t wave_off = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { if (w < wave) wave_off += s_wave[w]; all += s_wave[w]; }
    *total = all;
    return wave_off + incl - mine;
}

__global__ __launch_bounds__(256) void k_scan_block_sums(const uint32_t *__restrict__ in, int n, uint32_t *__restrict__ sums)
{
    const int base = blockIdx.x * SCAN_ITEMS + threadIdx.x * 4;
    uint32_t v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = (base + i < n) ? in[base + i] : 0u;
    uint32_t total;
    (void)block_exclusive_scan_256x4(v, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// single block: exclusive scan of the block sums in place; total -> counters[0]
__global__ __launch_bounds__(256) void k_scan_sums(uint32_t *__restrict__ sums, int nblocks, uint32_t *__restrict__ counters)
{
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += SCAN_ITEMS) {
        const int i0 = base + threadIdx.x * 4;
        uint32_t v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = (i0 + i < nblocks) ? sums[i0 + i] : 0u;
        uint32_t total;
        uint32_t off = block_exclusive_scan_256x4(v, &total) + s_carry;
#pragma unroll
        for (int i = 0; i < 4; i++) { if (i0 + i < nblocks) sums[i0 + i] = off; off += v[i]; }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { counters[0] = s_carry; counters[1] = 0; }    // [2],[3] are left to the caller (debug statistics)
}

__global__ __launch_bounds__(256) void k_scan_apply(uint32_t *__restrict__ data, int n, const uint32_t *__restrict__ sums,
                                                    const uint32_t *__restrict__ counters)
{
    const int base = blockIdx.x * SCAN_ITEMS + threadIdx.x * 4;
    uint32_t v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = (base + i < n) ? data[base + i] : 0u;
    uint32_t total;
    uint32_t off = block_exclusive_scan_256x4(v, &total) + sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 4; i++) { if (base + i < n) data[base + i] = off; off += v[i]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) data[n] = counters[0];
}

// VertexShader + the row-count scan in ONE single-workgroup launch, for scenes of a few thousand triangles (the
// reference's own scene has 30): four tiny dependent launches cost more in launch gaps than in work.
__global__ __launch_bounds__(256) void k_raster_vertex_scan(const RasterFrame f)
{
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) { s_carry = 0; f.scratch.counters[2] = 0u; }
    __syncthreads();
    for (int base = 0; base < f.n; base += SCAN_ITEMS) {
        const int i0 = base + threadIdx.x * 4;
        uint32_t v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            v[i] = 0u;
            if (i0 + i < f.n) {
                const TriSetup s = vertex_setup(f, i0 + i);
                f.scratch.setup[i0 + i] = s;
                v[i] = (uint32_t)s.rows;
                if (s.rows > 0) atomicMax(&f.scratch.counters[2], (uint32_t)s.rows);
            }
        }
        uint32_t total;
        uint32_t off = block_exclusive_scan_256x4(v, &total) + s_carry;
#pragma unroll
        for (int i = 0; i < 4; i++) { if (i0 + i < f.n) f.scratch.row_base[i0 + i] = off; off += v[i]; }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { f.scratch.row_base[f.n] = s_carry; f.scratch.counters[0] = s_carry; f.scratch.counters[1] = 0; }
}

// ---- left/right per row (rasteriser.cpp:716-733) and the span constants (:646-649) --------------------
// get(edge, field) = field {x, zinv, p.x, p.y, p.z} of that edge's sample on this row (global scratch table or the workgroup's LDS copy)
template <class Get>
__device__ __forceinline__ void build_span(const RasterFrame &f, const TriSetup &s, int t, uint32_t r, int y, Get get)
{
    int lx = INT_MAX, rx = -INT_MAX;
    float lz = 0.0f, rz = 0.0f;
    v3 lp = V3(0, 0, 0), rp = V3(0, 0, 0);
#pragma unroll
    for (int e = 0; e < 3; e++) {
        const int ya = s.y[e], yb = s.y[(e + 1) % 3];
        if (y < min(ya, yb) || y > max(ya, yb)) continue;       // this edge has no sample on this row
        const int x = __float_as_int(get(e, 0));
        const float sz = get(e, 1);
        const v3 sp3 = V3(get(e, 2), get(e, 3), get(e, 4));
        if (x < lx) { lx = x; lz = sz; lp = sp3; }   // strict <, first edge wins ties
        if (x > rx) { rx = x; rz = sz; rp = sp3; }   // strict >
    }
    Span sp;
    sp.tri = t; sp.y = y;
    sp.ax = lx;
    const long long d = (long long)rx - (long long)lx;
    sp.dx = (d > 0 && d < (1ll << 22)) ? (int)d : 0;
    const float fdx = (float)sp.dx;
    sp.azinv = lz;
    sp.zstep = (rz - lz) / fdx;                                   // :648 (unused when dx == 0)
    const v3 ps = div3s(sub3(rp, lp), fdx);                       // :649
    sp.ap[0] = lp.x; sp.ap[1] = lp.y; sp.ap[2] = lp.z;
    sp.pstep[0] = ps.x; sp.pstep[1] = ps.y; sp.pstep[2] = ps.z;
    f.scratch.spans[r] = sp;
}

// ---- Interpolate along the three edges (rasteriser.cpp:615-637, called from :706-715) -----------------
// `current += step` is a dependent float chain (a + k*step would round differently), so it cannot be split
// across threads: 15 chains per triangle (3 edges x {x, zinv, pos3d.xyz}) run in 15 lanes of one workgroup.
// The chain lanes only touch LDS (a global store per step made every step cost ~100 cycles); each EDGE_CHUNK
// steps the whole workgroup flushes the chunk to the per-row slots in global memory.
constexpr int EDGE_CHUNK = 512;

struct EdgeWalk { int ystart, dir, skip, cnt; };

__device__ __forceinline__ EdgeWalk edge_walk(const TriSetup &s, int e)
{
    // Samples are y = ya, ya+dir, ... yb.  Only those inside the band rows [r0, r1) are stored, but the float
    // chain has to be walked from the edge's first sample: `skip` pure additions, then `cnt` stored steps.
    const int ya = s.y[e], yb = s.y[(e + 1) % 3];
    const int r0 = s.r0, r1 = s.r0 + s.rows;
    EdgeWalk w;
    w.dir = (yb > ya) - (yb < ya);
    if (w.dir > 0) { w.ystart = max(ya, r0); w.skip = w.ystart - ya; w.cnt = min(yb, r1 - 1) - w.ystart + 1; }
    else if (w.dir < 0) { w.ystart = min(ya, r1 - 1); w.skip = ya - w.ystart; w.cnt = w.ystart - max(yb, r0) + 1; }
    else { w.ystart = ya; w.skip = 0; w.cnt = (ya >= r0 && ya < r1) ? 1 : 0; }
    if (w.cnt < 0) w.cnt = 0;
    return w;
}

__global__ __launch_bounds__(256) void k_raster_edges(const RasterFrame f, int handled_rows)
{
    __shared__ float bufs[2][EDGE_CHUNK * 15];
    const int t = blockIdx.x;
    const TriSetup &s = f.scratch.setup[t];
    const int rows = s.rows;
    if (rows == 0 || rows <= handled_rows) return;      // (k_raster_edges_lds took the triangles of up to handled_rows rows)
    const size_t base = f.scratch.row_base[t];
    if (base + (size_t)rows > f.scratch.cap_rows) { if (threadIdx.x == 0) atomicExch(&f.scratch.counters[1], 1u); return; }

    const EdgeWalk w0 = edge_walk(s, 0), w1 = edge_walk(s, 1), w2 = edge_walk(s, 2);
    const int maxcnt = max(max(w0.cnt, w1.cnt), w2.cnt);

    // Roles.  Chain lanes: wave 0 lanes 0..11 carry the twelve float fields (zinv, pos3d.xyz of the three edges),
    // wave 1 lanes 0..2 the three x chains (which also convert to int) -- a lone wave issues one instruction every
    // ~5 cycles, so the shorter each chain's loop body, the shorter the walk.  Flush lanes: 120 threads of waves 2-3,
    // 8 per channel, copy the PREVIOUS chunk from LDS to the per-row slots while the chains fill the next one.
    const int tid = threadIdx.x;
    int e = -1, fld = 0;
    if (tid < 12) { e = tid >> 2; fld = 1 + (tid & 3); }
    else if (tid >= 64 && tid < 67) { e = tid - 64; fld = 0; }
    float cur = 0.0f, step = 0.0f;
    int mycnt = 0;
    if (e >= 0) {
        const int i = e, j = (e + 1) % 3;
        const int N = abs(s.y[i] - s.y[j]) + 1;                      // :713
        const float div = (float)max(N - 1, 1);                      // :622
        if (fld == 0) { cur = (float)s.x[i]; step = (float)(s.x[j] - s.x[i]) / div; }
        else if (fld == 1) { cur = s.zinv[i]; step = (s.zinv[j] - s.zinv[i]) / div; }
        else { cur = s.p[i][fld - 2]; step = (s.p[j][fld - 2] - s.p[i][fld - 2]) / div; }
        const EdgeWalk w = (e == 0) ? w0 : (e == 1) ? w1 : w2;
        mycnt = w.cnt;
#pragma unroll 8
        for (int k = 0; k < w.skip; k++) cur += step;                // :632-635, sequential on purpose
    }
    const int ft = tid - 128;                                        // flush lane id
    const bool flusher = ft >= 0 && ft < 120;
    const int fg = ft / 15, fch = ft - fg * 15, fe = fch / 5;        // 8 lanes per channel
    const EdgeWalk fw = (fe == 0) ? w0 : (fe == 1) ? w1 : w2;
    float *slots = f.scratch.slots;

    const int nchunks = (maxcnt + EDGE_CHUNK - 1) / EDGE_CHUNK;
    for (int ci = 0; ci <= nchunks; ci++) {
        if (ci < nchunks && e >= 0) {
            const int c0 = ci * EDGE_CHUNK;
            const int kend = min(EDGE_CHUNK, mycnt - c0);
            float *dst = bufs[ci & 1] + e * 5 + fld;
            if (fld == 0) {
                // result[i].x = current.x truncates (:628).  |x| stays below 2^21 for in-contract triangles, so the
                // plain conversion equals the x86 one (f2i_x86) here.
#pragma unroll 8
                for (int k = 0; k < kend; k++) { dst[k * 15] = __int_as_float((int)cur); cur += step; }
            } else {
#pragma unroll 8
                for (int k = 0; k < kend; k++) { dst[k * 15] = cur; cur += step; }
            }
        }
        if (ci > 0 && flusher) {
            const int c0 = (ci - 1) * EDGE_CHUNK;
            const int kend = min(EDGE_CHUNK, fw.cnt - c0);           // steps of this channel's edge in the chunk
            const float *src = bufs[(ci - 1) & 1] + fch;
            // slot of step k: row y = ystart + dir*k  ->  ((base + y - r0)*3 + edge)*5 + field
            float *dst = slots + ((base + (size_t)(fw.ystart + fw.dir * (c0 + fg) - s.r0)) * 3 + fe) * SLOT_FIELDS + (fch - fe * 5);
            const ptrdiff_t dstep = (ptrdiff_t)fw.dir * 8 * 3 * SLOT_FIELDS;
            for (int k = fg; k < kend; k += 8) { *dst = src[k * 15]; dst += dstep; }
        }
        __syncthreads();
    }
    // all edge samples of this triangle are in its slots (written by this workgroup, barrier above): build the spans
    __threadfence_block();
    for (int r = threadIdx.x; r < rows; r += 256) {
        const float *slot = f.scratch.slots + (base + r) * 3 * SLOT_FIELDS;
        build_span(f, s, t, (uint32_t)(base + r), s.r0 + r, [&](int e, int fld) { return slot[e * SLOT_FIELDS + fld]; });
    }
}

// ---- the same with the sample table in LDS ---------------------------------------------------------------------------------
// k_raster_edges stages the samples through LDS into a global slot table in chunks of 512 steps (barrier per chunk, 120
// flush lanes, then the span pass reads the table back): 48 us for the 20 visible triangles of the Cornell box at 4K, a
// third of the frame, on 20 CUs.  A triangle of R rows needs at most R x 15 samples -- 130 KB at R = 2160, which a CU's
// 160 KB of LDS holds.  So triangles of up to `lds_rows` rows keep all their samples in LDS and the spans are built from
// there; taller ones take k_raster_edges.
//
// What bounds the kernel is the dependent float chain, walked by ONE wave that is alone on its SIMD.  Measured
// (tools/edgebench.hip): such a wave issues an instruction every ~8.6 cycles, dependent or not, and every LDS store in the
// stream costs about two more slots, whatever its width.  So the chain loop is stripped to what must be sequential:
//   * all 15 chains (3 edges x {x, zinv, pos3d.xyz}) sit in lanes 0..14 of wave 0 and run the SAME number of steps (the
//     longest edge; a shorter edge computes samples nobody reads), so the loop is wave-uniform: scalar loop control, no
//     exec-mask bookkeeping, unrolled 32 steps deep;
//   * the table is indexed by STEP, not by row (sample k of channel c at s[c*stride + k]): every lane stores to the same
//     offset, four consecutive steps are 16 contiguous bytes = one ds_write_b128 per four v_add_f32; the span pass maps a
//     row to each edge's step (k = (y - ystart) * dir);
//   * the x chains store the float; `result[i].x = current.x` (:628, truncation) is applied by the span pass.
// ~1.3 instructions per step instead of ~4.  The workgroup has 1024 threads: the span pass that follows the barrier (four
// IEEE divides and a 48-byte store per row) is spread over 16 waves instead of 4 lone ones.
__device__ __forceinline__ int edge_lds_stride(int lds_rows) { return ((lds_rows + 3) & ~3) + 4; }

__global__ __launch_bounds__(1024) void k_raster_edges_lds(const RasterFrame f, int lds_rows)
{
    extern __shared__ __attribute__((aligned(16))) float s_steps[];      // 15 channels x stride
    const int t = blockIdx.x;
    const TriSetup &s = f.scratch.setup[t];
    const int rows = s.rows;
    if (rows == 0 || rows > lds_rows) return;
    const size_t base = f.scratch.row_base[t];
    if (base + (size_t)rows > f.scratch.cap_rows) { if (threadIdx.x == 0) atomicExch(&f.scratch.counters[1], 1u); return; }
    const int stride = edge_lds_stride(lds_rows);
    const EdgeWalk w0 = edge_walk(s, 0), w1 = edge_walk(s, 1), w2 = edge_walk(s, 2);

    const int tid = threadIdx.x;
    if (f.edge_segments) {
    // The 15 chains (3 edges x {x, zinv, pos3d.xyz}), one WAVE per chain, each lane a segment of it: lane g starts from the value
    // edge_advance() predicts for its first step, walks its segment with the reference's own additions and hands its last sum to
    // lane g + 1, which compares it -- bit for bit -- with the prediction it started from.  Lane 0 starts from the true first
    // value, so if every comparison holds every stored sample is the one the sequential walk produces; if one fails (a prediction
    // that is wrong is merely slow, never visible) lane 0 walks the whole chain the old way.  A 4K-tall edge is 2160 dependent
    // additions, ~9 cycles apart in a wave alone on its SIMD whatever one does to the loop: 16-25 us of the frame's chain for the
    // twenty triangles of the Cornell box; as 64 segments of 34 it is the predictions' few dozen instructions plus 34 additions:
    // k_raster_edges_lds 18-19 -> 14.4 us, the 4K frame alone 61.1 -> 56.5 us.  Fifteen busy waves per triangle instead of one get in
    // the way of the neighbouring frames' kernels, though (four frames in flight: 28.4 against 27.8 us per frame), so the host asks for
    // segments when at most two frames are in flight (f.edge_segments; MIRT_EDGE_SEGMENTS=0|1 fixes it).
    {
        const int wave = tid >> 6, lane = tid & 63, nwaves = (int)blockDim.x >> 6;
        const int L = ((max(max(w0.cnt, w1.cnt), w2.cnt) + 3) >> 2) << 2;       // samples kept per chain (workgroup-uniform; a shorter edge keeps samples nobody reads)
        for (int c = wave; c < 3 * SLOT_FIELDS; c += nwaves) {
            const int e = c / SLOT_FIELDS, fld = c - e * SLOT_FIELDS;
            const int i = e, j = (e + 1) % 3;
            const int N = abs(s.y[i] - s.y[j]) + 1;                      // :713
            const float div = (float)max(N - 1, 1);                      // :622
            float cur0, step;
            if (fld == 0) { cur0 = (float)s.x[i]; step = (float)(s.x[j] - s.x[i]) / div; }
            else if (fld == 1) { cur0 = s.zinv[i]; step = (s.zinv[j] - s.zinv[i]) / div; }
            else { cur0 = s.p[i][fld - 2]; step = (s.p[j][fld - 2] - s.p[i][fld - 2]) / div; }
            const EdgeWalk w = (e == 0) ? w0 : (e == 1) ? w1 : w2;
            // sample k of the chain is its value after k additions; those before the band (k < skip) are walked, not kept (:632-635)
            const int T = w.skip + L, seg = (T + 63) >> 6, n0 = min(lane * seg, T), n1 = min(n0 + seg, T);
            float *dst = s_steps + c * stride;                           // sample k lives at dst[k - skip]
            float start = edge_advance(cur0, step, n0);
            // (tests: MIRT_EDGE_SEGMENTS=2 spoils one lane's prediction, so that the check fails and the whole-chain walk runs)
            if (f.edge_segments == 2 && lane == 5 && n0 < T) start = __uint_as_float(__float_as_uint(start) ^ 1u);
            float cur = start;
            for (int k = n0; k < n1; k++) { if (k >= w.skip) dst[k - w.skip] = cur; cur += step; }
            const float handed = __shfl_up(cur, 1);                      // the previous lane's sum after ITS last addition = this lane's first value
            const bool ok = lane == 0 || n0 >= T || __float_as_uint(handed) == __float_as_uint(start);
            if (__builtin_amdgcn_ballot_w64(!ok) != 0ull && lane == 0) {
                cur = cur0;
                for (int k = 0; k < T; k++) { if (k >= w.skip) dst[k - w.skip] = cur; cur += step; }
            }
        }
    }
    } else if (tid < 64) {                           // wave 0: lane = channel = edge*5 + field
        const int e = min(tid / SLOT_FIELDS, 2), fld = tid - e * SLOT_FIELDS;
        const int i = e, j = (e + 1) % 3;
        const int N = abs(s.y[i] - s.y[j]) + 1;                      // :713
        const float div = (float)max(N - 1, 1);                      // :622
        float cur, step;
        if (fld == 0) { cur = (float)s.x[i]; step = (float)(s.x[j] - s.x[i]) / div; }
        else if (fld == 1) { cur = s.zinv[i]; step = (s.zinv[j] - s.zinv[i]) / div; }
        else { cur = s.p[i][min(fld - 2, 2)]; step = (s.p[j][min(fld - 2, 2)] - s.p[i][min(fld - 2, 2)]) / div; }
        const EdgeWalk w = (e == 0) ? w0 : (e == 1) ? w1 : w2;
        // samples before the band are walked, not stored: `skip` pure additions (wave-uniform count: the longest)
        const int skipmax = max(max(w0.skip, w1.skip), w2.skip);
        for (int k = 0; k < skipmax; k++) cur = (k < w.skip) ? cur + step : cur;      // :632-635, sequential on purpose
        const int groups = (max(max(w0.cnt, w1.cnt), w2.cnt) + 3) >> 2;          // wave-uniform
        float4 *dst = reinterpret_cast<float4 *>(s_steps + min(tid, 14) * stride);
        if (tid < 15) {
#pragma unroll 8
            for (int g4 = 0; g4 < groups; g4++) {
                float4 v;
                v.x = cur; cur += step;
                v.y = cur; cur += step;
                v.z = cur; cur += step;
                v.w = cur; cur += step;
                dst[g4] = v;
            }
        }
    }
    __syncthreads();
    const int dirs[3] = { w0.dir, w1.dir, w2.dir }, starts[3] = { w0.ystart, w1.ystart, w2.ystart };
    for (int r = tid; r < rows; r += (int)blockDim.x) {
        const int y = s.r0 + r;
        build_span(f, s, t, (uint32_t)(base + r), y, [&](int e2, int f2) {
            const int k = (y - starts[e2]) * dirs[e2];               // this row's sample of edge e2
            const float v = s_steps[(e2 * SLOT_FIELDS + f2) * stride + k];
            return f2 == 0 ? __int_as_float((int)v) : v;             // :628 (|x| < 2^21 in contract: (int) equals the x86 conversion)
        });
    }
}

// ---- fragments with an atomic z-compare (rasteriser.cpp:603-610, 657-669) -----------------------------
__global__ __launch_bounds__(256) void k_raster_frag(const RasterFrame f)
{
    const uint32_t R = min(f.scratch.counters[0], (uint32_t)f.scratch.cap_rows);
    const int lane = threadIdx.x & 63;
    const uint32_t wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t r = wave_id; r < R; r += nwaves) {
        const Span &sp = f.scratch.spans[r];
        const int dx = sp.dx;
        if (dx <= 0) continue;
        const int ax = sp.ax, y = sp.y;
        // fragments are x = ax+1+i, i in [0,dx); those with x outside [0,W) are never produced (:663, E-2)
        const int i0 = max(0, -ax - 1), i1 = min(dx, f.W - ax - 1);
        const float azinv = sp.azinv, zstep = sp.zstep;
        const unsigned long long low = 0xFFFFFFFFull - (unsigned long long)(uint32_t)sp.tri;
        unsigned long long *row = f.scratch.keys + (size_t)(y - f.y0) * f.W;
        for (int i = i0 + lane; i < i1; i += 64) {
            const float zinv = azinv + zstep * (float)i;              // :667
            if (zinv > 0.0f)                                          // depthBuffer starts at 0, strict > (:606)
                atomicMax(row + (ax + 1 + i), ((unsigned long long)__float_as_uint(zinv) << 32) | low);
        }
    }
}

// ---- PixelShader (rasteriser.cpp:549-589) on the winner + CalculateDOF's PutPixelSDL (:491-519) ------
__global__ __launch_bounds__(256) void k_raster_resolve(const RasterFrame f)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = f.y0 + blockIdx.y;
    if (x >= f.W) return;
    unsigned long long *kp = f.scratch.keys + (size_t)(y - f.y0) * f.W + x;
    const unsigned long long key = *kp;
    if (key != 0ull) *kp = 0ull;                 // leave the depth keys cleared for the next frame (Update() :188): no memset per frame
    v3 colour = V3(0.0f, 0.0f, 0.0f);            // Update() cleared pixelColours (:189)
    float zinv = 0.0f;                           // and depthBuffer (:188)
    float fdist = 0.0f;
    int tri = -1;
    if (key != 0ull) {
        tri = (int)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull));
        zinv = __uint_as_float((uint32_t)(key >> 32));
        const TriSetup &s = f.scratch.setup[tri];
        const Span &sp = f.scratch.spans[f.scratch.row_base[tri] + (uint32_t)(y - s.r0)];
        const float fi = (float)(x - sp.ax - 1);
        const v3 p3 = add3(ld3(sp.ap), scale3(ld3(sp.pstep), fi));   // a.pos3d + pos3d*float(i) (:668)
        const float *t15 = f.tris15 + (size_t)15 * tri;
        const v3 normal = ld3(t15 + 9), color = ld3(t15 + 12);
        v3 P = div3s(p3, zinv);                                       // pPos3d /= p.zinv (:557)
        P = vec_mul_mat3(P, f.invrot);                                // * glm::inverse(cameraRot) (:559)
        P = add3(P, ld3(f.cam));                                      // += cameraPos (:560)
        if (f.fd) fdist = distance3(P, ld3(f.cam)) - f.focal_plane;   // focalDistances (:563-565): only the depth-of-field pass reads them
        v3 result = V3(0.0f, 0.0f, 0.0f);
        for (int k = 0; k < f.nlights; k++) {
            const v3 L = ld3(f.lpos[k]);
            const float r = distance3(P, L);                          // :574
            const float A = sphere_area(r);                           // :575
            const v3 rDir = normalize3(sub3(L, P));                   // :577
            const v3 B = div3s(ld3(f.lcol[k]), A);                    // :579
            const float d = dot3(rDir, normal);                       // normal NOT re-normalised here (:578)
            const float m = (d < 0.0f) ? 0.0f : d;                    // std::max (:581)
            result = add3(result, scale3(B, m));
        }
        // currentReflectance(1,1,1) * (result + indirectLightPowerPerArea) * color (:587)
        colour = mul3(mul3(V3(1.0f, 1.0f, 1.0f), add3(result, ld3(f.indirect))), color);
    }
    const size_t px = (size_t)y * f.W + x;
    if (f.rgb) st3(f.rgb + 3 * px, colour);
    if (f.zinv) f.zinv[px] = zinv;
    if (f.fd) f.fd[px] = fdist;
    if (f.index) f.index[px] = tri;
    // Update() paints every pixel black (:190); CalculateDOF then draws the interior only (:491-493)
    const bool interior = x >= 1 && x < f.W - 1 && y >= 1 && y < f.H - 1;
    f.xrgb[(size_t)(y - f.row_origin) * f.pitch_words + x] = interior ? pack_xrgb(colour) : 0u;
}

// ---- scenes of at most 64 triangles: depth test in registers, no key buffer ---------------------------------------------------
// The reference's own scene has 30 triangles, 20 of them visible.  For it the atomic z-buffer is the wrong tool: k_raster_frag
// moves 8 bytes per fragment through the memory-side atomic units (~1.3 TB/s chip-wide: 45 us at 4K) and k_raster_resolve reads
// the keys back and re-zeroes them.  With so few triangles a row holds a handful of spans, so here a workgroup takes 512
// pixels of one row: its first wave collects the spans of that row that reach into those pixels (span constants, plus the
// triangle's normal and colour) into LDS, in triangle order; every thread then owns TWO pixels (x and x + 256), evaluates
//     zinv = a.zinv + zstep * float(x - a.x - 1)                                   (rasteriser.cpp:667)
// for the listed spans that contain each of them, keeps the largest (strict `>` in ascending triangle order = the reference's
// sequential `zinv > depthBuffer[y][x]`, :606: largest zinv, lowest index among exact ties) and shades the winner -- PixelShader
// (:549-589) for both pixels in packed FP32, every operand from LDS.  Same arithmetic on the same operands as k_raster_frag +
// k_raster_resolve, so depthBuffer, pixelColours and the surface come out bit-identical; the frame writes 4 bytes per pixel
// and reads nothing but the spans.
constexpr int SMALL_PX = 512;                       // pixels of a row per work item: four passes of 128 (two per lane)
constexpr int SMALL_ROWS = 4;                       // waves per workgroup: every wave works alone, on its own items

// PERSISTENT waves (round 4).  The first version started a wave per (row, 512 pixels): 17 280 short waves at 4K, each beginning with
// two dependent round trips to memory (the triangle's row bookkeeping, then its span of the row) before its ~760 vector
// instructions -- the kernel ran at half its issue floor with the waves mostly waiting (SQ_WAIT_INST_ANY > SQ_ACTIVE_INST_ANY).
// Now the launch holds as many waves as the chip keeps resident (five per SIMD) and a wave takes a contiguous run of items (row, 512
// pixels), row-major: the per-triangle words (first row, row count, where its spans start, normal, colour) are loaded ONCE per
// wave; the spans of a row are loaded once per row -- the 512-pixel segments of a row share them.  The kernel alone takes 40 us where
// it took 43.5; what the change buys shows with frames in flight: the 4K frame went from 38.4 to 31.5 us (the setup kernels of the
// next frame no longer queue behind 17 000 waves waiting for their first loads).
// (Requesting the NEXT row's spans before the current row is shaded -- twelve more registers, four waves per SIMD instead of five --
// measured no better: 43.7 us alone against 40.4 without, 4K Cornell box; kept as a build variant.)
#ifndef MIRT_SMALL_PREFETCH
#define MIRT_SMALL_PREFETCH 0
#endif
__global__ __launch_bounds__(256) void k_raster_small(const RasterFrame f)
{
    __shared__ __attribute__((aligned(16))) SmallSpan s_list[SMALL_ROWS][SMALL_MAX_TRIS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rr = wave;
    const int xsegs = (f.W + SMALL_PX - 1) / SMALL_PX, band_rows = f.y1 - f.y0;
    const int items = xsegs * band_rows;                    // (at most 64 x 32768)
    const int nwaves = (int)gridDim.x * SMALL_ROWS, gw = (int)blockIdx.x * SMALL_ROWS + wave;
    const int per = (items + nwaves - 1) / nwaves;
    const int it0 = gw * per, it1 = min(items, it0 + per);
    if (it0 >= it1) return;
    // ---- once per wave: lane t holds what does not depend on the row of triangle t ----
    int t_r0 = 0, t_rows = 0;
    size_t t_base = 0;
    if (lane < f.n) {
        const TriSetup &st = f.scratch.setup[lane];
        t_r0 = st.r0; t_rows = st.rows;
        t_base = f.scratch.row_base[lane];
        // (a triangle whose rows do not fit the span table was not walked: the frame is incomplete, and mirt_sync says so --
        // the edge kernels raise the same flag, this one does not rely on them)
        if (t_rows > 0 && t_base + (size_t)t_rows > f.scratch.cap_rows) { atomicExch(&f.scratch.counters[1], 1u); t_rows = 0; }
        const float *t15 = f.tris15 + (size_t)15 * lane;
        float4 *dst = reinterpret_cast<float4 *>(&s_list[rr][lane]);
        // normal (words 10..12), colour (13..15), triangle (16): rows 2.zw, 3, 4.x of the record; the row's part comes below
        reinterpret_cast<float *>(dst)[10] = t15[9]; reinterpret_cast<float *>(dst)[11] = t15[10];
        dst[3] = make_float4(t15[11], t15[12], t15[13], t15[14]);
        reinterpret_cast<float *>(dst)[16] = __int_as_float(lane);
    }
    auto row_has = [&](int y) { return t_rows > 0 && y >= t_r0 && y < t_r0 + t_rows; };
    auto load_row = [&](int y, float4 &a, float4 &b, float4 &c) {
        a = make_float4(0.0f, 0.0f, 0.0f, 0.0f); b = a; c = a;        // {ax, dx = 0, ...}: no span in this row
        if (row_has(y)) {
            const float4 *src = reinterpret_cast<const float4 *>(f.scratch.spans + t_base + (uint32_t)(y - t_r0));
            a = src[0]; b = src[1]; c = src[2];                        // {ax, dx, azinv, zstep | ap.xyz, pstep.x | pstep.yz, tri, y}
        }
    };
    const int y_first = f.y0 + it0 / xsegs, y_last = f.y0 + (it1 - 1) / xsegs;
#if MIRT_SMALL_PREFETCH
    float4 na, nb, nc;                                                 // the spans of the next row this wave will need
    load_row(y_first, na, nb, nc);
#else
    (void)y_first; (void)y_last;
#endif
    const v3p camp = splat3(ld3(f.cam));
    const bool planes = f.rgb || f.zinv || f.fd || f.index;
    int row = f.y0 - 1;
    int sp_first = 0, sp_count = 0;             // pixels x = sp_first + i, 0 <= i < sp_count, clipped to x < W (never produced otherwise: :663, E-2)
    float sp_azinv = 0.0f, sp_zstep = 0.0f;
    bool sp_any = false;
    for (int it = it0; it < it1; it++) {
    const int y = f.y0 + it / xsegs, xbase = (it % xsegs) * SMALL_PX;
    if (y != row) {
        // ---- once per row: the lane's span of row y into registers (depth test) and LDS (shading); the next row's on its way ----
        row = y;
#if MIRT_SMALL_PREFETCH
        const float4 a = na, b = nb, c = nc;
        if (y < y_last) load_row(y + 1, na, nb, nc);
#else
        float4 a, b, c;
        load_row(y, a, b, c);
#endif
        const int ax = __float_as_int(a.x), dx = __float_as_int(a.y);
        // fragments are x = ax+1 .. ax+dx (|ax|, dx <= 2^21: RASTER_COORD_LIMIT); those outside [0, W) are never produced
        sp_first = ax + 1;
        sp_count = min(dx, f.W - 1 - ax);
        sp_azinv = a.z; sp_zstep = a.w;
        sp_any = dx > 0 && sp_count > 0;
        // (wave-private LDS: the passes of the previous row have read their records)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (sp_any) {
            // Both ends of the span's walk in the range of the shared-reciprocal division (mirt_math2.hpp: div3p_sel) and no
            // component changing sign => every pixel between them too: a + step * float(i) is monotone in i, roundings included
            const float fl = (float)(dx - 1);
            const float ze = a.z + a.w * fl, xe = b.x + b.w * fl, ye = b.y + c.x * fl, ze3 = b.z + c.y * fl;
            const bool safe = a.z >= DIV3_LO && a.z < DIV3_HI && ze >= DIV3_LO && ze < DIV3_HI &&
                              div3_mag_in_range(b.x) && div3_mag_in_range(xe) && (b.x < 0.0f) == (xe < 0.0f) &&
                              div3_mag_in_range(b.y) && div3_mag_in_range(ye) && (b.y < 0.0f) == (ye < 0.0f) &&
                              div3_mag_in_range(b.z) && div3_mag_in_range(ze3) && (b.z < 0.0f) == (ze3 < 0.0f);
            float4 *dst = reinterpret_cast<float4 *>(&s_list[rr][lane]);
            dst[0] = a; dst[1] = b;
            reinterpret_cast<float *>(dst)[8] = c.x; reinterpret_cast<float *>(dst)[9] = c.y;     // pstep.yz
            reinterpret_cast<float *>(dst)[17] = __int_as_float(safe ? 1 : 0);
        }
    }
    // The spans of row y that draw into [xbase, xbase + SMALL_PX): seg_mask[s] = the triangles whose span reaches into the 128
    // pixels of pass s (the passes fetch the span constants with v_readlane, triangle by triangle).
    unsigned long long seg_mask[SMALL_PX / 128];
    {
        const bool take = sp_any && sp_first + sp_count > xbase && sp_first < xbase + SMALL_PX;
#pragma unroll
        for (int seg = 0; seg < SMALL_PX / 128; seg++)
            seg_mask[seg] = __builtin_amdgcn_ballot_w64(take && sp_first + sp_count > xbase + seg * 128 && sp_first < xbase + seg * 128 + 128);
        // (wave-private LDS: the wave's own accesses execute in order; this only stops the compiler from moving them)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

#pragma unroll
    for (int seg = 0; seg < SMALL_PX / 128; seg++) {
        // the lane's two pixels are neighbours: both inside or both outside a surface almost always, so a pass whose pixels
        // nothing covers skips the shading whole, and the pair leaves as one 8-byte store
        const int x0 = xbase + seg * 128 + 2 * lane, x1 = x0 + 1;
        if (xbase + seg * 128 >= f.W) break;
        const bool ok0 = x0 < f.W, ok1 = x1 < f.W;
        // ---- depth test (:603-608) for both pixels against the spans of this pass, in ascending triangle order with the
        // reference's strict `zinv > depthBuffer[y][x]` (depthBuffer starts at 0, Update() :188): the largest zinv wins, the lowest
        // triangle among exact ties, NaN never ----
        float bzf0 = 0.0f, bzf1 = 0.0f;
        int bj0 = -1, bj1 = -1;
        for (unsigned long long m = seg_mask[seg]; m; m &= m - 1ull) {
            const int j = __builtin_ctzll(m);
            const int first = __builtin_amdgcn_readlane(sp_first, j), cnt = __builtin_amdgcn_readlane(sp_count, j);
            const float az = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sp_azinv), j));
            const float zs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sp_zstep), j));
            const int i0 = x0 - first, i1 = i0 + 1;                               // Bresenham's i of this pixel (:657)
            const f2 z = splat2(az) + splat2(zs) * (f2){ (float)i0, (float)i1 };  // :667
            const bool w0 = (unsigned)i0 < (unsigned)cnt && z.x > bzf0, w1 = (unsigned)i1 < (unsigned)cnt && z.y > bzf1;
            bzf0 = w0 ? z.x : bzf0; bj0 = w0 ? j : bj0;
            bzf1 = w1 ? z.y : bzf1; bj1 = w1 ? j : bj1;
        }
        const uint32_t bz0 = __float_as_uint(bzf0), bz1 = __float_as_uint(bzf1);
        const int bt0 = bj0, bt1 = bj1;                                           // (the list slot of a span is its triangle)

        // ---- PixelShader (:549-589) of the two winners, packed ----
        v3p colour = splat3(V3(0.0f, 0.0f, 0.0f));  // Update() cleared pixelColours (:189)
        f2 fdist = splat2(0.0f);
#ifdef MIRT_SMALL_NOSHADE
        if (false) {
#else
        if (bj0 >= 0 || bj1 >= 0) {
#endif
            // a pixel nothing covers shades a copy of its neighbour (nobody reads the result): the lane's two halves are then both
            // real fragments, and whatever holds for the operands of the live one holds for both
            const int sj0 = bj0 >= 0 ? bj0 : bj1, sj1 = bj1 >= 0 ? bj1 : bj0;
            // The two spans' records word by word (SmallSpan as 20 words), each word of span a beside the same word of span b: read
            // one by one (volatile), a pair lands in the two registers the packed arithmetic reads it from.  As four 16-byte reads
            // per span the values arrive span-major and ~20 v_mov per pass re-pair them -- a vector instruction each, where the
            // extra LDS reads cost next to nothing.
            typedef const volatile __attribute__((address_space(3))) float lds_vfloat;
            lds_vfloat *wa = (lds_vfloat *)reinterpret_cast<const float *>(&s_list[rr][sj0]);
            lds_vfloat *wb = (lds_vfloat *)reinterpret_cast<const float *>(&s_list[rr][sj1]);
#define SPAN_PAIR(w) ((f2){ wa[w], wb[w] })
            const int axa = __float_as_int(wa[0]), axb = __float_as_int(wb[0]);
            const int safe2 = __float_as_int(wa[17]) & __float_as_int(wb[17]);
            const f2 zinv = { bj0 >= 0 ? bzf0 : bzf1, bj1 >= 0 ? bzf1 : bzf0 };
            const f2 fi = { (float)((bj0 >= 0 ? x0 : x1) - axa - 1), (float)((bj1 >= 0 ? x1 : x0) - axb - 1) };
            const v3p ap = V3P(SPAN_PAIR(4), SPAN_PAIR(5), SPAN_PAIR(6)), ps = V3P(SPAN_PAIR(7), SPAN_PAIR(8), SPAN_PAIR(9));
            const v3p p3 = add3p(ap, scale3p(ps, fi));                              // a.pos3d + pos3d*float(i) (:668)
            const v3p normal = V3P(SPAN_PAIR(10), SPAN_PAIR(11), SPAN_PAIR(12)), color = V3P(SPAN_PAIR(13), SPAN_PAIR(14), SPAN_PAIR(15));
#undef SPAN_PAIR
            v3p P;                                                                  // pPos3d /= p.zinv (:557); the operands' range is known per span
            div3p_sel(p3.x, p3.y, p3.z, zinv, __builtin_amdgcn_ballot_w64(safe2 == 0), P.x, P.y, P.z);
            const float *m = f.invrot;                                              // * glm::inverse(cameraRot) (:559): vec * mat
            P = V3P(m[0] * P.x + m[1] * P.y + m[2] * P.z, m[3] * P.x + m[4] * P.y + m[5] * P.z, m[6] * P.x + m[7] * P.y + m[8] * P.z);
            P = add3p(P, camp);                                                     // += cameraPos (:560)
            if (f.fd) fdist = distance3p(P, camp) - splat2(f.focal_plane);          // focalDistances (:563-565)
            v3p result = splat3(V3(0.0f, 0.0f, 0.0f));
            for (int k = 0; k < f.nlights; k++) {
                // r = distance(pPos3d, lightPos) (:574), A = 4 pi r^2 (:575), rDir = normalize(lightPos - pPos3d) (:577), B = lightColor / A (:579)
                const LightGeometry2 lg = light_geometry2(P, ld3(f.lpos[k]), ld3(f.lcol[k]), f.lights_in_range != 0, true, true);
                const f2 d = dot3p(lg.rDir, normal);                                // normal NOT re-normalised here (:578)
                const f2 mx = { (d.x < 0.0f) ? 0.0f : d.x, (d.y < 0.0f) ? 0.0f : d.y };   // std::max (:581)
                result = add3p(result, scale3p(lg.B, mx));
            }
            // currentReflectance(1,1,1) * (result + indirectLightPowerPerArea) * color (:587)
            const v3p lit = mul3p(mul3p(splat3(V3(1.0f, 1.0f, 1.0f)), add3p(result, splat3(ld3(f.indirect)))), color);
            colour = join3(bj0 >= 0 ? half0(lit) : V3(0.0f, 0.0f, 0.0f), bj1 >= 0 ? half1(lit) : V3(0.0f, 0.0f, 0.0f));
            if (bj0 < 0) fdist.x = 0.0f;
            if (bj1 < 0) fdist.y = 0.0f;
        }
        uint32_t word[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int x = h ? x1 : x0;
            const v3 c = h ? half1(colour) : half0(colour);
            // Update() paints every pixel black (:190); CalculateDOF then draws the interior only (:491-493)
            const bool interior = x >= 1 && x < f.W - 1 && y >= 1 && y < f.H - 1;
            word[h] = interior ? pack_xrgb(c) : 0u;
            if (!planes || !(h ? ok1 : ok0)) continue;                  // (one wave-uniform test for the optional planes together)
            const size_t px = (size_t)y * f.W + x;
            if (f.rgb) st3(f.rgb + 3 * px, c);
            if (f.zinv) f.zinv[px] = __uint_as_float(h ? bz1 : bz0);
            if (f.fd) f.fd[px] = h ? fdist.y : fdist.x;
            if (f.index) f.index[px] = h ? bt1 : bt0;
        }
        uint32_t *dst = f.xrgb + (size_t)(y - f.row_origin) * f.pitch_words + x0;
        if (ok1 && (reinterpret_cast<uintptr_t>(dst) & 7u) == 0) *reinterpret_cast<uint2 *>(dst) = make_uint2(word[0], word[1]);
        else { if (ok0) dst[0] = word[0]; if (ok1) dst[1] = word[1]; }
    }
    }                                                       // (items of this wave)
}

// ---- host side ----------------------------------------------------------------------------------------

void raster_scratch_free(RasterScratch &s)
{
    for (void *p : { (void *)s.setup, (void *)s.row_base, (void *)s.block_sums, (void *)s.slots, (void *)s.spans,
                     (void *)s.keys, (void *)s.counters })
        if (p) (void)hipFree(p);
    s = RasterScratch();
}

static int grow(void **p, size_t bytes)
{
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    return hipMalloc(p, bytes) == hipSuccess ? MIRT_OK : MIRT_ERR_OUT_OF_MEMORY;
}

static int ensure_rows(RasterScratch &s, size_t rows)
{
    if (rows <= s.cap_rows) return MIRT_OK;
    size_t cap = rows + rows / 8 + 1024;
    if (grow((void **)&s.slots, cap * 3 * SLOT_FIELDS * sizeof(float))) { s.cap_rows = 0; return MIRT_ERR_OUT_OF_MEMORY; }
    if (s.cap_spans < cap) {
        if (grow((void **)&s.spans, cap * sizeof(Span))) { s.cap_rows = 0; s.cap_spans = 0; return MIRT_ERR_OUT_OF_MEMORY; }
        s.cap_spans = cap;
    }
    s.cap_rows = cap;
    return MIRT_OK;
}

int raster_scratch_ensure(RasterScratch &s, int n, int W, int band_rows)
{
    if (!s.counters && grow((void **)&s.counters, 16)) return MIRT_ERR_OUT_OF_MEMORY;
    if (n > s.cap_tris) {
        if (grow((void **)&s.setup, (size_t)n * sizeof(TriSetup))) return MIRT_ERR_OUT_OF_MEMORY;
        if (grow((void **)&s.row_base, ((size_t)n + 1) * sizeof(uint32_t))) return MIRT_ERR_OUT_OF_MEMORY;
        if (grow((void **)&s.block_sums, ((size_t)n / SCAN_ITEMS + 2) * sizeof(uint32_t))) return MIRT_ERR_OUT_OF_MEMORY;
        s.cap_tris = n;
        s.sizing_valid = false;
    }
    const size_t px = (size_t)W * band_rows;
    if (px > s.cap_px) {
        s.keys_zero_px = 0;
        if (grow((void **)&s.keys, px * sizeof(unsigned long long))) { s.cap_px = 0; return MIRT_ERR_OUT_OF_MEMORY; }
        s.cap_px = px;
    }
    return ensure_rows(s, 4096);
}

static uint64_t frame_key(const RasterFrame &f, uint64_t scene_version)
{
    uint64_t h = 0xcbf29ce484222325ull ^ scene_version;
    auto mix = [&](const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; } };
    mix(f.cam, sizeof f.cam); mix(f.rot, sizeof f.rot); mix(&f.focal, 4); mix(&f.W, 4); mix(&f.H, 4);
    mix(&f.y0, 4); mix(&f.y1, 4); mix(&f.n, 4);
    return h | 1ull;
}

// Enqueues the whole rasteriser frame on `stream`.  ev (nullable) = the library's per-kernel event pairs,
// indexed 2*MIRT_K_*, ev_used (nullable) = which of them this frame recorded.  Returns 0 or a negative mirt_status.
int launch_raster(RasterFrame &f, RasterScratch &s, uint64_t scene_version, hipStream_t stream, hipEvent_t *ev, bool *ev_used)
{
    // (ev_used[k]: this frame recorded kernel slot k -- a path that skips a kernel, k_raster_small for one, must not report the
    // times an earlier frame of the stream left in that slot's events)
    auto begin = [&](int k) { if (ev) { (void)hipEventRecord(ev[2 * k], stream); if (ev_used) ev_used[k] = true; } };
    auto end = [&](int k) { if (ev) (void)hipEventRecord(ev[2 * k + 1], stream); };
    const int band_rows = f.y1 - f.y0;

    // The depth keys of the band must be zero (depthBuffer cleared by Update(), :188).  k_raster_resolve zeroes every key it
    // consumes, so after a completed frame the first `keys_zero_px` slots are zero again and only a larger band needs a memset.
    // (Scenes of at most 64 triangles never touch the keys: k_raster_small below.)
    static const bool small_off = [] { const char *e = getenv("MIRT_RASTER_SMALL"); return e && atoi(e) == 0; }();   // (A/B runs, tests of the atomic path)
    const bool small = f.n <= SMALL_MAX_TRIS && !small_off;
    const size_t band_px = (size_t)f.W * band_rows;
    if (!small && band_px > s.keys_zero_px) {
        begin(MIRT_K_CLEAR);
        s.keys_zero_px = 0;
        if (hipMemsetAsync(s.keys, 0, band_px * sizeof(unsigned long long), stream) != hipSuccess) return MIRT_ERR_HIP;
        s.keys_zero_px = band_px;
        end(MIRT_K_CLEAR);
    }

    auto set_scratch = [&]() { f.scratch = s; };
    begin(MIRT_K_RASTER_SETUP);
    set_scratch();
    if (f.n <= 4096) {
        hipLaunchKernelGGL(k_raster_vertex_scan, dim3(1), dim3(256), 0, stream, f);
    } else {
        if (hipMemsetAsync(s.counters + 2, 0, 4, stream) != hipSuccess) return MIRT_ERR_HIP;      // tallest triangle (atomicMax below)
        hipLaunchKernelGGL(k_raster_vertex, dim3((f.n + 255) / 256), dim3(256), 0, stream, f);
        enqueue_exclusive_scan(s.row_base, f.n, s.block_sums, s.counters, stream);
    }

    // The span table (and the slot table of k_raster_edges) is sized from the total row count, which only the device knows.
    //  * Small scenes -- n x band rows spans fit 64 MiB, e.g. the Cornell box at any size -- get the worst case once and
    //    never ask: no read-back, no sync, whatever the camera does.
    //  * Otherwise the count is read back (16 bytes + one sync of this stream) when the frame's geometry inputs changed since
    //    the last call; a frame with the same inputs reuses the count.
    static const int lds_rows_max = [] { const char *e = getenv("MIRT_RASTER_LDS_ROWS"); int v = e ? atoi(e) : 2560; return v < 0 ? 0 : (v > 2560 ? 2560 : v); }();
    const size_t worst_rows = (size_t)f.n * (size_t)band_rows;
    const bool worst_case = f.n <= 4096 && worst_rows * sizeof(Span) <= ((size_t)64 << 20) && lds_rows_max > 0;
    int lds_rows;
    bool tall;                                           // some triangle may be taller than the LDS kernel takes
    if (worst_case) {
        if (s.cap_spans < worst_rows) {
            if (grow((void **)&s.spans, worst_rows * sizeof(Span))) { s.cap_spans = 0; return MIRT_ERR_OUT_OF_MEMORY; }
            s.cap_spans = worst_rows;
        }
        lds_rows = std::min(band_rows, lds_rows_max);
        tall = band_rows > lds_rows;
        if (tall && ensure_rows(s, worst_rows)) return MIRT_ERR_OUT_OF_MEMORY;      // slot table of the chunked kernel
        set_scratch();
        f.scratch.cap_rows = s.cap_spans;
    } else {
        const uint64_t key = frame_key(f, scene_version);
        if (!s.sizing_valid || s.sizing_key != key) {
            uint32_t c[4] = { 0, 0, 0, 0 };               // [0] total rows, [1] overflow flag, [2] tallest triangle
            if (hipStreamSynchronize(stream) != hipSuccess) return MIRT_ERR_HIP;      // (the stream first, then a synchronous copy: mirt_capi.hip bin_pass)
            if (hipMemcpy(c, s.counters, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) return MIRT_ERR_HIP;
            if (ensure_rows(s, c[0])) return MIRT_ERR_OUT_OF_MEMORY;
            s.max_rows = c[2];
            s.sizing_key = key;
            s.sizing_valid = true;
        }
        set_scratch();
        lds_rows = (int)std::min<uint32_t>(s.max_rows, (uint32_t)lds_rows_max);
        tall = s.max_rows > (uint32_t)lds_rows;
    }
    // edge walk + span build, one workgroup per triangle: samples in LDS for triangles of up to lds_rows rows (60 bytes per
    // row, at most 150 KiB), the chunked global-table kernel for taller ones
    if (lds_rows > 0) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_raster_edges_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 15 * 4 * (2560 + 4));
            attr_set = true;
        }
        // 1024 threads where the span pass has rows to spread them over; small triangles keep 256 (more workgroups per CU)
        const int threads = lds_rows > 512 ? 1024 : 256;
        hipLaunchKernelGGL(k_raster_edges_lds, dim3(f.n), dim3(threads), (size_t)15 * 4 * (((lds_rows + 3) & ~3) + 4), stream, f, lds_rows);
    }
    if (tall) hipLaunchKernelGGL(k_raster_edges, dim3(f.n), dim3(256), 0, stream, f, lds_rows);
    end(MIRT_K_RASTER_SETUP);

    // Scenes of at most 64 triangles: one kernel does the depth test in registers and shades (no key buffer, no atomics).
    if (small) {
        begin(MIRT_K_RASTER_RESOLVE);
        // persistent waves: as many as stay resident (five per SIMD; MIRT_SMALL_WGS_PER_CU for A/B runs), each with a contiguous run of
        // (row, 512-pixel) items
        static const int cus = [] { int dev = 0; hipDeviceProp_t p; return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256; }();
        const long long items = (long long)((f.W + SMALL_PX - 1) / SMALL_PX) * band_rows;
        static const int per_cu = [] { const char *e = getenv("MIRT_SMALL_WGS_PER_CU"); int v = e ? atoi(e) : 0; return v > 0 ? v : 5; }();
        const unsigned wgs = (unsigned)std::max<long long>(1, std::min<long long>((items + SMALL_ROWS - 1) / SMALL_ROWS, (long long)cus * per_cu));
        hipLaunchKernelGGL(k_raster_small, dim3(wgs), dim3(256), 0, stream, f);
        end(MIRT_K_RASTER_RESOLVE);
        if (hipGetLastError() != hipSuccess) return MIRT_ERR_HIP;
        return MIRT_OK;
    }

    begin(MIRT_K_RASTER_FRAG);
    const int frag_blocks = (int)min((size_t)8192, (f.scratch.cap_rows + 3) / 4);
    hipLaunchKernelGGL(k_raster_frag, dim3(frag_blocks), dim3(256), 0, stream, f);
    end(MIRT_K_RASTER_FRAG);

    begin(MIRT_K_RASTER_RESOLVE);
    hipLaunchKernelGGL(k_raster_resolve, dim3((f.W + 255) / 256, band_rows), dim3(256), 0, stream, f);
    end(MIRT_K_RASTER_RESOLVE);
    if (hipGetLastError() != hipSuccess) { s.keys_zero_px = 0; return MIRT_ERR_HIP; }    // the keys may hold fragments nobody consumed
    return MIRT_OK;
}

// ---- the cull step of Update() on the device (rasteriser.cpp:404-447): one thread per triangle ----
// The per-frame constants come from the host (cull_setup: acosf / tanf from the host's libm, as in the reference); the
// per-triangle arithmetic is cull_one, the very function mirt_cull runs on the host.
__global__ __launch_bounds__(256) void k_cull(const float *__restrict__ tris15, int n, const CullParams cp, uint8_t *__restrict__ culled)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) culled[i] = cull_one(tris15 + (size_t)15 * i, cp);
}

}  // namespace mirt
