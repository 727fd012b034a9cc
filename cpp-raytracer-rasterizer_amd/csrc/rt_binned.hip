// rt_binned.hip -- binning kernels and the binned ray-trace kernel (see rt_binned.hpp for the argument why
// the candidate reduction cannot change any result).
#include "rt_binned.hpp"

#include <float.h>

namespace mirt {

// ---- hierarchical binning: three levels of the same conservative rectangle test, one kernel -----------
//
//   level 0  one THREAD per (triangle, frame, 64x64-bin cell): one rectangle test
//   level 1  the wave then takes its surviving lanes one at a time (ballot loop, item broadcast with v_readlane):
//            lane = one of the cell's 8x8 coarse cells (8x8 bins each)
//   level 2  for every surviving coarse cell: lane = one of its 64 bins -> count (pass 1) or fill (pass 2)
// All 64 lanes always work on the same item, so triangle size does not cause divergence, and no work queue
// (hence no contended queue counter) is needed.  COUNT and FILL run the identical tests, so the fill pass
// finds exactly the slots the count pass reserved.
//
// The rectangle test of rt_binned.hpp (rect_may_hit) is evaluated here in "folded" form.  The sign of e1e2b
// decides which of its two branches can hold (t >= 0 needs sign(e1e2d) == sign(e1e2b)), so the four functions
// are multiplied by that sign and their margin is added once per item:  F_k = sgn * g_k + m_k  must reach >= 0
// somewhere in the rectangle for all k.  On the regular bin grid the maximum of an affine function over the cell
// that starts at bin (I, J) and spans K bins is itself affine in (I, J):
//     max F = A + I*Bu + J*Bv + max(su*pad_lo, su*(K*du + pad_hi)) + max(sv*pad_lo, sv*(K*dv + pad_hi))
// i.e. two FMAs per function and cell.  The (u,v) box of add_bbox() becomes a range of bin indices.
constexpr int BIN_L0 = BIN_COARSE * BIN_COARSE;       // 64 bins per level-0 cell side

struct BinItem {
    float A1[4], A2[4], Bu[4], Bv[4];     // level-1 (K = 8) and level-2 (K = 1) constants, per-bin slopes
    float lou, hiu, lov, hiv;             // box as bin-index ranges: cell [I, I+K) overlaps iff I+K >= lou && I <= hiu
};

__device__ __forceinline__ float bcastf(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }

__device__ __forceinline__ float corner(float s, float lo, float hi) { return fmaxf(s * lo, s * hi); }

template <bool FILL>
__global__ __launch_bounds__(256) void k_bin(const float *__restrict__ tris15, const OriginRow *__restrict__ cam_tab,
                                             const OriginRow *__restrict__ light_tab, int n, BinSet bs, BinGridInfo gi)
{
    const int lane = threadIdx.x & 63;
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t tri = id / gi.cells_per_tri;
    // which (frame, level-0 cell) this thread owns: frame 0 has cam_cells cells, every cube face face_cells_x^2
    const uint32_t c = id - tri * gi.cells_per_tri;
    uint32_t frame, cx, cy;
    if (c < gi.cam_cells) { frame = 0; cx = c % gi.cam_cells_x; cy = gi.cam_cell_y0 + c / gi.cam_cells_x; }
    else {
        const uint32_t per_face = gi.face_cells_x * gi.face_cells_x, cc = c - gi.cam_cells;
        frame = 1 + cc / per_face;
        const uint32_t within = cc - (frame - 1) * per_face;
        cx = within % gi.face_cells_x; cy = within / gi.face_cells_x;
    }
    bool pass0 = false;
    BinItem it;
    memset(&it, 0, sizeof it);
    if (tri < (uint32_t)n) {
        const BinFrameDesc &fr = bs.frames[frame];
        const OriginRow &row = (fr.tab == 0) ? cam_tab[tri] : light_tab[(size_t)(fr.tab - 1) * n + tri];
        TriBinFns t = make_bin_fns(row, fr);
        add_bbox(t, tris15 + (size_t)15 * tri, fr);
        const float T = 2.384185791015625e-07f;       // |e1e2b| < 2^-22: t may underflow to +-0, either sign of e1e2d passes
        const bool both = fabsf(t.nb) < T || !(t.nb == t.nb);
        const float sgn = t.nb < 0.0f ? -1.0f : 1.0f;
        const EdgeFn *fn[4] = { &t.n, &t.p, &t.q, &t.s };
        float A0[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float su = sgn * fn[k]->cu, sv = sgn * fn[k]->cv;
            // value at the (unpadded) origin corner of bin (0,0), margin folded in
            const float base = sgn * fn[k]->c0 + fn[k]->m + su * fr.ulo + sv * fr.vlo;
            it.Bu[k] = su * fr.du;
            it.Bv[k] = sv * fr.dv;
            const float inf = __builtin_huge_valf();
            A0[k] = both ? inf : base + corner(su, fr.pad_lo, (float)BIN_L0 * fr.du + fr.pad_hi) + corner(sv, fr.pad_lo, (float)BIN_L0 * fr.dv + fr.pad_hi);
            it.A1[k] = both ? inf : base + corner(su, fr.pad_lo, (float)BIN_COARSE * fr.du + fr.pad_hi) + corner(sv, fr.pad_lo, (float)BIN_COARSE * fr.dv + fr.pad_hi);
            it.A2[k] = both ? inf : base + corner(su, fr.pad_lo, fr.du + fr.pad_hi) + corner(sv, fr.pad_lo, fr.dv + fr.pad_hi);
            if (both) { it.Bu[k] = 0.0f; it.Bv[k] = 0.0f; }
        }
        const float inf = __builtin_huge_valf();
        if (t.bstate == BOX_VALID) {
            // bin-index ranges, widened by 2^-18 (relative) against the rounding of this conversion
            it.lou = (t.bu0 - fr.ulo - fr.pad_hi) / fr.du; it.hiu = (t.bu1 - fr.ulo - fr.pad_lo) / fr.du;
            it.lov = (t.bv0 - fr.vlo - fr.pad_hi) / fr.dv; it.hiv = (t.bv1 - fr.vlo - fr.pad_lo) / fr.dv;
            it.lou -= 3.814697265625e-06f * (1.0f + fabsf(it.lou)); it.hiu += 3.814697265625e-06f * (1.0f + fabsf(it.hiu));
            it.lov -= 3.814697265625e-06f * (1.0f + fabsf(it.lov)); it.hiv += 3.814697265625e-06f * (1.0f + fabsf(it.hiv));
        } else if (t.bstate == BOX_EMPTY) {
            it.lou = it.lov = inf; it.hiu = it.hiv = -inf;
        } else {
            it.lou = it.lov = -inf; it.hiu = it.hiv = inf;
        }
        // level 0: the cell of 64x64 bins starting at bin (cx*64, cy*64)
        const float I = (float)(cx * BIN_L0), J = (float)(cy * BIN_L0);
        const int j0 = max((int)cy * BIN_L0, fr.j0), j1 = min((int)(cy + 1) * BIN_L0, fr.j1);
        bool ok = j1 > j0 && (int)(cx * BIN_L0) < fr.nbu;
#pragma unroll
        for (int k = 0; k < 4; k++) ok = ok && (__builtin_fmaf(J, it.Bv[k], __builtin_fmaf(I, it.Bu[k], A0[k])) >= 0.0f);
        ok = ok && (I + (float)BIN_L0 >= it.lou) && (I <= it.hiu) && (J + (float)BIN_L0 >= it.lov) && (J <= it.hiv);
        pass0 = ok;
    }
    unsigned long long m0 = __ballot(pass0);
#ifdef MIRT_BIN_STATS
    if (!FILL && lane == 0) atomicAdd(&bs.counters[2], (uint32_t)__popcll(m0));
#endif
    while (m0) {
        const int src = __builtin_ctzll(m0);
        m0 &= m0 - 1ull;
        // the surviving lane's item, made wave-uniform
        BinItem u;
#pragma unroll
        for (int k = 0; k < 4; k++) { u.A1[k] = bcastf(it.A1[k], src); u.A2[k] = bcastf(it.A2[k], src); u.Bu[k] = bcastf(it.Bu[k], src); u.Bv[k] = bcastf(it.Bv[k], src); }
        u.lou = bcastf(it.lou, src); u.hiu = bcastf(it.hiu, src); u.lov = bcastf(it.lov, src); u.hiv = bcastf(it.hiv, src);
        const uint32_t utri = (uint32_t)__builtin_amdgcn_readlane((int)tri, src);
        const uint32_t ufr = (uint32_t)__builtin_amdgcn_readlane((int)frame, src);
        const uint32_t ucx = (uint32_t)__builtin_amdgcn_readlane((int)cx, src), ucy = (uint32_t)__builtin_amdgcn_readlane((int)cy, src);
        const BinFrameDesc &fr = bs.frames[ufr];
        const int nbu = fr.nbu, fj0 = fr.j0, fj1 = fr.j1;
        const uint32_t fbase = fr.base;
        // level 1: lane = coarse cell (8x8 bins) inside the level-0 cell
        const int ci = (int)(ucx * BIN_L0) + (lane & 7) * BIN_COARSE, cj = (int)(ucy * BIN_L0) + (lane >> 3) * BIN_COARSE;
        const float CI = (float)ci, CJ = (float)cj;
        bool pass1 = ci < nbu && cj + BIN_COARSE > fj0 && cj < fj1;
#pragma unroll
        for (int k = 0; k < 4; k++) pass1 = pass1 && (__builtin_fmaf(CJ, u.Bv[k], __builtin_fmaf(CI, u.Bu[k], u.A1[k])) >= 0.0f);
        pass1 = pass1 && (CI + (float)BIN_COARSE >= u.lou) && (CI <= u.hiu) && (CJ + (float)BIN_COARSE >= u.lov) && (CJ <= u.hiv);
        unsigned long long m1 = __ballot(pass1);
#ifdef MIRT_BIN_STATS
        if (!FILL && lane == 0) atomicAdd(&bs.counters[3], (uint32_t)__popcll(m1));
#endif
        while (m1) {
            const int cl = __builtin_ctzll(m1);
            m1 &= m1 - 1ull;
            // level 2: lane = bin inside coarse cell `cl`
            const int i = (int)(ucx * BIN_L0) + (cl & 7) * BIN_COARSE + (lane & 7);
            const int j = (int)(ucy * BIN_L0) + (cl >> 3) * BIN_COARSE + (lane >> 3);
            const float FI = (float)i, FJ = (float)j;
            bool pass2 = i < nbu && j >= fj0 && j < fj1;
#pragma unroll
            for (int k = 0; k < 4; k++) pass2 = pass2 && (__builtin_fmaf(FJ, u.Bv[k], __builtin_fmaf(FI, u.Bu[k], u.A2[k])) >= 0.0f);
            pass2 = pass2 && (FI + 1.0f >= u.lou) && (FI <= u.hiu) && (FJ + 1.0f >= u.lov) && (FJ <= u.hiv);
            if (pass2) {
                const uint32_t bin = fbase + (uint32_t)j * nbu + i;
                if (!FILL) {
                    atomicAdd(&bs.bin_off[bin], 1u);                 // counts, scanned in place afterwards
                } else {
                    const uint32_t slot = bs.bin_off[bin] + atomicAdd(&bs.bin_fill[bin], 1u);
                    if (slot < bs.cap_entries) bs.entries[slot] = utri;
                    else atomicExch(&bs.counters[1], 1u);
                }
            }
        }
    }
}

template __global__ void k_bin<false>(const float *, const OriginRow *, const OriginRow *, int, BinSet, BinGridInfo);
template __global__ void k_bin<true>(const float *, const OriginRow *, const OriginRow *, int, BinSet, BinGridInfo);

// ---- k_rt_binned: fused primary + shadow + shade + resolve over the binned candidates ---------------
//
// Workgroup = 256 threads = 4 wave64; each wave owns one 8x8-pixel tile (= one camera bin), the block a
// 16x16 area.  Primary rays: the tile's candidate rows are gathered 64 at a time into the wave's LDS slice
// (one lane loads one candidate's 48-byte origin row), then every lane walks them with broadcast reads -- the
// same inner loop as brute force.  Shadow rays leave the light in incoherent directions, so every lane walks
// the list of its own light-cube bin straight from global memory (the tables live in L2 / Infinity Cache).
struct RtBinnedFrame {
    RtFrame f;
    BinSet bins;
    uint32_t cam_base;                       // bin base of the camera frame
    uint32_t light_base[MIRT_MAX_LIGHTS];    // bin base of face 0 of each light
    int tiles_x;                             // camera bins per row (nbu)
    int cube_bins;                           // light-cube bins per face side
};

// order-independent form of the reference's sequential ">=" update (raytracer.cpp:243-247):
// smaller distance wins, equal distance -> larger triangle index wins
__device__ __forceinline__ bool closer(float dist, int idx, float best_d, int best_i)
{
    return dist < best_d || (dist == best_d && idx > best_i);
}

template <bool AA>
__global__ __launch_bounds__(256) void k_rt_binned(const RtBinnedFrame bf)
{
    const RtFrame &f = bf.f;
    __shared__ __attribute__((aligned(16))) float4 s_rows[4][64 * 3];
    __shared__ uint32_t s_idx[4][64];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tx = (int)blockIdx.x * 2 + (wave & 1);
    const int ty = f.y0 / BIN_TILE + (int)blockIdx.y * 2 + (wave >> 1);
    const int x = tx * BIN_TILE + (lane & 7), y = ty * BIN_TILE + (lane >> 3);
    const bool tile_ok = tx < bf.tiles_x && ty * BIN_TILE < f.y1;
    const bool ok = tile_ok && x < f.W && y >= f.y0 && y < f.y1;
    if (!tile_ok) return;                                  // wave-uniform
    const v3 cam = ld3(f.cam);
    const int rs = AA ? f.aa : 1;                          // realSamples (:549-554); compile-time 1 without supersampling

    const uint32_t cbin = bf.cam_base + (uint32_t)ty * bf.tiles_x + tx;
    const uint32_t cbeg = bf.bins.bin_off[cbin], cend = bf.bins.bin_off[cbin + 1];

    float best_d = FLT_MAX;                                // Update() reset (:335-339), once per frame
    int best_i = -1;
    v3 pos = V3(0.0f, 0.0f, 0.0f), avg = V3(0.0f, 0.0f, 0.0f);
    unsigned ntests = 0;

    float y1 = aa_start(y, rs);                            // :566-569
    for (int z = 0; z < rs; z++) {
        float x1 = aa_start(x, rs);                        // :573-576
        for (int z2 = 0; z2 < rs; z2++) {
            // d = (x1 - W/2, y1 - H/2, focalLength); negD = -(cameraRot * d)   (raytracer.cpp:579-580, :229)
            const v3 d = V3(x1 - (float)f.W / 2.0f, y1 - (float)f.H / 2.0f, f.focal);
            const v3 nd = neg3(mat3_mul_vec(f.rot, d));
            // closest hit of THIS sub-ray among the tile's candidates (order-independent: min distance, max index) ...
            float sd = FLT_MAX;
            int si = -1;
            v3 sp = V3(0.0f, 0.0f, 0.0f);
            if (ok) ntests += cend - cbeg;
            for (uint32_t base = cbeg; base < cend; base += 64) {
                const int cnt = (int)min(64u, cend - base);
                if (lane < cnt) {
                    const uint32_t idx = bf.bins.entries[base + lane];
                    const float4 *src = reinterpret_cast<const float4 *>(f.cam_tab + idx);
                    s_idx[wave][lane] = idx;
                    s_rows[wave][3 * lane] = src[0];
                    s_rows[wave][3 * lane + 1] = src[1];
                    s_rows[wave][3 * lane + 2] = src[2];
                }
                // wave-private LDS slice: the wave's own writes are visible to it after the LDS counter drains
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 2
                for (int j = 0; j < cnt; j++) {
                    const float4 r0 = s_rows[wave][3 * j], r1 = s_rows[wave][3 * j + 1], r2 = s_rows[wave][3 * j + 2];
                    const TestDots td = test_dots(r0, r1, r2, nd);
                    if (maybe_hit(td)) {
                        const int idx = (int)s_idx[wave][j];
                        v3 hp;
                        float dist;
                        if (exact_hit(td, r0.w, f.tris15 + (size_t)15 * idx, cam, &hp, &dist))
                            if (closer(dist, idx, sd, si)) { sd = dist; si = idx; sp = hp; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            // ... merged into the pixel's running record exactly as the sequential `>=` sweep would (:243): the
            // sub-ray's best replaces the record when it is at least as close (a later sub-ray wins exact ties)
            const bool any = si >= 0;                      // ClosestIntersection's return value
            if (any && best_d >= sd) { best_d = sd; best_i = si; pos = sp; }

            const bool hit = ok && any;
            count_hits(f, (unsigned long long)__popcll(__ballot(hit)));
            if (hit) {
                const float *t = f.tris15 + (size_t)15 * best_i;
                const v3 nDir = normalize3(ld3(t + 9));            // (:300)
                const v3 tcol = ld3(t + 12);
                v3 result = V3(0.0f, 0.0f, 0.0f), result2 = V3(0.0f, 0.0f, 0.0f);
                for (int k = 0; k < f.nlights; k++) {
                    const v3 L = ld3(f.lpos[k]);
                    v3 rd;
                    float r;
                    v3 D = light_term(f, k, pos, nDir, &rd, &r);
                    const float thr = r * 0.99f;                   // (:313)
                    const uint32_t bin = cube_bin_of(rd, bf.light_base[k], bf.cube_bins);
                    const uint32_t beg = bf.bins.bin_off[bin], end = bf.bins.bin_off[bin + 1];
                    const OriginRow *tab = f.light_tab + (size_t)k * f.n;
                    uint32_t e = beg;
                    for (; e < end; e++) {
                        const uint32_t idx = bf.bins.entries[e];
                        const float4 *src = reinterpret_cast<const float4 *>(tab + idx);
                        const float4 r0 = src[0], r1 = src[1], r2 = src[2];
                        const TestDots td = test_dots(r0, r1, r2, rd);   // negD = rDir (:310, :229)
                        if (maybe_hit(td)) {
                            v3 hp;
                            float dist;
                            if (exact_hit(td, r0.w, f.tris15 + (size_t)15 * idx, L, &hp, &dist) && dist < thr) {
                                D = V3(0.0f, 0.0f, 0.0f);          // occluded (:313-314); any-hit is exact
                                e++;
                                break;
                            }
                        }
                    }
                    ntests += e - beg;
                    result = add3(result, D);                      // (:319)
                    if ((k + 1) % f.samples == 0) result2 = add3(result2, result);   // (:322) after each light's samples
                }
                const v3 Dl = mul3(result2, tcol);                 // (:325-326)
                const v3 T = add3(Dl, ld3(f.indirect));            // (:584-586)
                avg = add3(avg, mul3(tcol, T));                    // (:587-591)
                x1 += aa_step(rs);                                 // (:593) only after a hit
            }
        }
        y1 += aa_step(rs);                                         // (:596)
    }
    count_tests(f, ntests);
    if (!ok) return;
    avg = div3s(avg, (float)(rs * rs));                            // (:599)
    const size_t px = (size_t)y * f.W + x;
    if (f.rgb) st3(f.rgb + 3 * px, avg);
    if (f.index) f.index[px] = best_i;
    if (f.fd) f.fd[px] = best_i >= 0 ? best_d - f.focal_plane : 0.0f;          // focalDistances (:248-249)
    if (x >= 1 && x < f.W - 1 && y >= 1 && y < f.H - 1)            // (:618-620)
        f.xrgb[(size_t)(y - f.row_origin) * f.pitch_words + x] = pack_xrgb(avg);
}

template __global__ void k_rt_binned<false>(const RtBinnedFrame);
template __global__ void k_rt_binned<true>(const RtBinnedFrame);

}  // namespace mirt
