// rt_tile.hip -- the ray tracer for scenes of at most 64 triangles (the reference's own Cornell box has 30):
// everything lives in LDS and every wave prunes the triangle list for its tile before it walks it.
//
// k_rt_small (rt_kernels.hip) already keeps the whole scene in LDS, but each of its rays still runs the filter
// against all n triangles, twice (primary + shadow).  Here a wave owns a 16 x 8-pixel tile, two pixels per lane in packed
// FP32 (tile_body2; round 1's one-pixel-per-lane form, 7 % slower, is gone), and first builds two 64-bit candidate masks
// with ONE lane per triangle:
//   * primary rays: the conservative rectangle test of rt_binned.hpp (affine edge functions of the camera frame
//     over the tile's pixel rectangle);
//   * shadow rays of light k: the same sign conditions evaluated with interval arithmetic over the bounding box of
//     the wave's shadow-ray directions (six 64-lane min/max reductions in DPP, no LDS traffic).
// The wave then walks only the set bits, in ascending index order, with the very same filter + exact arithmetic as
// every other kernel -- so the `>=` tie rule and all results stay bit-identical (a pruned triangle could never have
// been accepted by any ray of the wave).  The per-frame tables (origin rows, geometry, edge functions, normalised
// normals, colours) are built once per frame by k_tile_tables and copied into LDS by every workgroup; one tile per wave.
#include "rt_binned.hpp"

#include <float.h>

namespace mirt {

struct RtTileFrame {
    RtFrame f;
    BinFrameDesc cam;        // camera ray family (P0, Pu, Pv, dmax); bins are not used, only the edge functions
    int tiles_x, tiles_y;    // tiles in the band
    unsigned long long *clear_hits;   // the OTHER hit-counter buffer of this stream: zeroed by k_tile_tables for the next frame
    float4 *tables;                   // tile_table_rows() float4, built by k_tile_tables, copied into LDS by every workgroup
};


// interval of g . x for x in the box [lo, hi]
__device__ __forceinline__ void dot_range(float gx, float gy, float gz, v3 lo, v3 hi, float *rlo, float *rhi)
{
    const float ax = gx * lo.x, bx = gx * hi.x, ay = gy * lo.y, by = gy * hi.y, az = gz * lo.z, bz = gz * hi.z;
    *rlo = fminf(ax, bx) + fminf(ay, by) + fminf(az, bz);
    *rhi = fmaxf(ax, bx) + fmaxf(ay, by) + fmaxf(az, bz);
}

// May some ray with negD inside the box [lo, hi] be accepted by the reference's test against this origin row?
// Same conditions and margins as rect_may_hit (rt_binned.hpp), with |negD| components bounded by 1 (+ rounding).
__device__ __forceinline__ bool box_may_hit(const float4 &r0, const float4 &r1, const float4 &r2, v3 lo, v3 hi)
{
    const float K = 7.62939453125e-06f * 1.0009765625f, C = 9.5367431640625e-07f;      // 2^-17 * dmax, 2^-20
    const float mn = K * (fabsf(r0.x) + fabsf(r0.y) + fabsf(r0.z)) + C;
    const float mp = K * (fabsf(r1.x) + fabsf(r1.y) + fabsf(r1.z)) + C;
    const float mq = K * (fabsf(r2.x) + fabsf(r2.y) + fabsf(r2.z)) + C;
    const float ms = 2.0f * (mn + mp + mq);
    float nlo, nhi, plo, phi, qlo, qhi, slo, shi;
    dot_range(r0.x, r0.y, r0.z, lo, hi, &nlo, &nhi);
    dot_range(r1.x, r1.y, r1.z, lo, hi, &plo, &phi);
    dot_range(r2.x, r2.y, r2.z, lo, hi, &qlo, &qhi);
    dot_range(r0.x - r1.x - r2.x, r0.y - r1.y - r2.y, r0.z - r1.z - r2.z, lo, hi, &slo, &shi);
    const float T = 2.384185791015625e-07f, nb = r0.w;
    const bool pos = (nhi > -mn) && (phi >= -mp) && (qhi >= -mq) && (shi >= -ms) && (nb > -T);
    const bool neg = (nlo < mn) && (plo <= mp) && (qlo <= mq) && (slo <= ms) && (nb < T);
    return pos || neg;
}

// The accept test and hit point (raytracer.cpp:237-242) for the two pixels of a lane at once (k_rt_tile2): the six divisions as three packed pairs, the triangle's geometry
// read once, hit points and distances in packed arithmetic -- operation for operation what exact_hit (rt_common.hpp) does per half.  A half
// the filter rejected computes values nobody reads (its `m` is false), like an inactive lane.
__device__ __forceinline__ void exact_hit_geo2(const TestDots2 &d, float e1e2b, const float4 *geo, v3 start, bool m0, bool m1,
                                               bool *hit0, bool *hit1, v3p *pos, f2 *dist)
{
    // (three quotients over one denominator, but div3p_sel's range test and branch cost this divergent loop more than the shared
    // reciprocal saves: 23.2 vs 22.3 us per frame)
    const f2 t = div2(splat2(e1e2b), d.den), u = div2(d.pu, d.den), v = div2(d.qv, d.den);     // raytracer.cpp:237
    const f2 uv = u + v;
    *hit0 = m0 && uv.x <= 1.0f && u.x >= 0.0f && v.x >= 0.0f && t.x >= 0.0f;                     // :239
    *hit1 = m1 && uv.y <= 1.0f && u.y >= 0.0f && v.y >= 0.0f && t.y >= 0.0f;
    if (*hit0 || *hit1) {
        const float4 g0 = geo[0], g1 = geo[1], g2 = geo[2];
        const v3p v0 = splat3(V3(g0.x, g0.y, g0.z)), e1 = splat3(V3(g0.w, g1.x, g1.y)), e2 = splat3(V3(g1.z, g1.w, g2.x));
        const v3p p = add3p(add3p(v0, scale3p(e1, u)), scale3p(e2, v));                         // :241
        *pos = p;
        *dist = distance3p(splat3(start), p);                                                    // :242
    }
}

// Per-frame tables of the tile kernels: origin rows of the camera and of every light position, geometry, the
// camera-frame edge functions and the shading constants of every triangle -- (12 + 3*nlights) float4 per triangle.
// k_tile_tables builds them once per frame (one lane per triangle) into tf.tables; every workgroup of the tile
// kernels copies them into LDS.  (Building them in each of the ~4000 workgroups cost about a tenth of the frame's
// VALU issue slots.)
struct TileTables { const float4 *cam, *geo, *fns, *shade, *light; };

__host__ __device__ __forceinline__ int tile_table_rows(int n, int nlights) { return n * (12 + 3 * nlights); }

// The rows of triangle t, written to tables laid out as described above (global memory or LDS).
__device__ __forceinline__ void tile_table_rows_of(const RtTileFrame &tf, int t, float4 *base)
{
    const RtFrame &f = tf.f;
    const int n = f.n;
    float4 *s_cam = base;                        // 3 rows per triangle
    float4 *s_geo = s_cam + 3 * n;               // 3
    float4 *s_fns = s_geo + 3 * n;               // 4: the camera-frame edge functions n, p, q, s
    float4 *s_shade = s_fns + 4 * n;             // 2: {normalize(normal), -}, {color, -}
    float4 *s_light = s_shade + 2 * n;           // nlights x 3 rows per triangle
    const float *t15 = f.tris15 + (size_t)15 * t;
    const OriginRow r = make_origin_row(t15, ld3(f.cam));
    s_cam[3 * t] = r.r0; s_cam[3 * t + 1] = r.r1; s_cam[3 * t + 2] = r.r2;
    const v3 v0 = ld3(t15), e1 = sub3(ld3(t15 + 3), v0), e2 = sub3(ld3(t15 + 6), v0);       // :216-217
    s_geo[3 * t] = make_float4(v0.x, v0.y, v0.z, e1.x);
    s_geo[3 * t + 1] = make_float4(e1.y, e1.z, e2.x, e2.y);
    s_geo[3 * t + 2] = make_float4(e2.z, 0.0f, 0.0f, 0.0f);
    const TriBinFns b = make_bin_fns(r, tf.cam);
    s_fns[4 * t] = make_float4(b.n.c0, b.n.cu, b.n.cv, b.n.m);
    s_fns[4 * t + 1] = make_float4(b.p.c0, b.p.cu, b.p.cv, b.p.m);
    s_fns[4 * t + 2] = make_float4(b.q.c0, b.q.cu, b.q.cv, b.q.m);
    s_fns[4 * t + 3] = make_float4(b.s.c0, b.s.cu, b.s.cv, b.s.m);
    const v3 nd = normalize3(ld3(t15 + 9));                                  // :300
    s_shade[2 * t] = make_float4(nd.x, nd.y, nd.z, 0.0f);
    s_shade[2 * t + 1] = make_float4(t15[12], t15[13], t15[14], 0.0f);
    for (int k = 0; k < f.nlights; k++) {
        const OriginRow rl = make_origin_row(t15, ld3(f.lpos[k]));
        float4 *dst = s_light + 3 * ((size_t)k * n + t);
        dst[0] = rl.r0; dst[1] = rl.r1; dst[2] = rl.r2;
    }
}

__global__ __launch_bounds__(64) void k_tile_tables(const RtTileFrame tf)
{
    for (int i = threadIdx.x; i < HIT_SHARDS * HIT_SHARD_STRIDE; i += 64) tf.clear_hits[i] = 0ull;   // counters of the next frame on this stream
    if ((int)threadIdx.x < tf.f.n) tile_table_rows_of(tf, threadIdx.x, tf.tables);
}

// tf.tables != nullptr: copy the tables k_tile_tables built; nullptr (small frames: a second launch would cost more
// than it saves): build them here, and let workgroup 0 clear the next frame's hit counters.
__device__ __forceinline__ TileTables tile_tables_load(const RtTileFrame &tf, float4 *s_all)
{
    const int n = tf.f.n, rows = tile_table_rows(n, tf.f.nlights);
    if (tf.tables) {
        for (int i = threadIdx.x; i < rows; i += blockDim.x) s_all[i] = tf.tables[i];
    } else {
        for (int t = threadIdx.x; t < n; t += blockDim.x) tile_table_rows_of(tf, t, s_all);
        if (blockIdx.x == 0)
            for (int i = threadIdx.x; i < HIT_SHARDS * HIT_SHARD_STRIDE; i += blockDim.x) tf.clear_hits[i] = 0ull;
    }
    __syncthreads();
    TileTables tb;
    tb.cam = s_all;
    tb.geo = tb.cam + 3 * n;
    tb.fns = tb.geo + 3 * n;
    tb.shade = tb.fns + 4 * n;
    tb.light = tb.shade + 2 * n;
    return tb;
}

// ---- two pixels per lane -------------------------------------------------------------------------------------------
// The same tile algorithm with a wave owning TW x (128/TW) pixels: lane l carries pixel (lx, ly) and the pixel
// 64/TW rows below it.  Every multiply and add of the two rays shares one packed instruction (mirt_math2.hpp), the
// LDS rows of a candidate triangle are read once for both rays, and the per-tile work (candidate masks, direction-box
// reductions) is spread over twice the pixels.  With supersampling each half carries its own sub-ray position.

template <int TW, bool AA>
__device__ __forceinline__ void tile_body2(const RtTileFrame &tf, int tx, int ty, const TileTables &tb)
{
    constexpr int TH = 64 / TW;                   // rows per half; the tile is 2*TH rows tall
    const RtFrame &f = tf.f;
    const int lane = threadIdx.x & 63, n = f.n;
    const int x0 = tx * TW, y0 = f.y0 + ty * 2 * TH;
    const int x = x0 + (lane % TW), ya = y0 + (lane / TW), yb = ya + TH;
    const bool okx = x < f.W, ok0 = okx && ya < f.y1, ok1 = okx && yb < f.y1;
    const v3 cam = ld3(f.cam);

    const int rs = AA ? f.aa : 1;                 // realSamples (:549-554); compile-time 1 without supersampling: the loops fold away
    const float reach = rs > 1 ? 0.5f : 0.0f;

    // ---- primary candidates: one lane per triangle tests the tile's pixel rectangle ----
    bool cand = false;
    if (lane < n) {
        TriBinFns t;
        const float4 a = tb.fns[4 * lane], b = tb.fns[4 * lane + 1], c = tb.fns[4 * lane + 2], d4 = tb.fns[4 * lane + 3];
        t.n.c0 = a.x; t.n.cu = a.y; t.n.cv = a.z; t.n.m = a.w;
        t.p.c0 = b.x; t.p.cu = b.y; t.p.cv = b.z; t.p.m = b.w;
        t.q.c0 = c.x; t.q.cu = c.y; t.q.cv = c.z; t.q.m = c.w;
        t.s.c0 = d4.x; t.s.cu = d4.y; t.s.cv = d4.z; t.s.m = d4.w;
        t.nb = tb.cam[3 * lane].w;
        t.bstate = BOX_NONE; t.bu0 = t.bu1 = t.bv0 = t.bv1 = 0.0f;
        // (with supersampling the sub-rays reach half a pixel beyond the pixel centres on every side)
        cand = rect_may_hit(t, (float)x0 - reach, (float)min(x0 + TW - 1, f.W - 1) + reach,
                            (float)y0 - reach, (float)min(y0 + 2 * TH - 1, f.y1 - 1) + reach);
    }
    const unsigned long long pmask = __ballot(cand);
    unsigned ntests = 0;                                                     // ray-triangle tests this lane runs

    float bd0 = FLT_MAX, bd1 = FLT_MAX;                                      // Update() reset (:335-339)
    int bi0 = -1, bi1 = -1;
    v3 pos0 = V3(0.0f, 0.0f, 0.0f), pos1 = pos0;
    // d = (x - W/2, y - H/2, focalLength); negD = -(cameraRot * d)   (raytracer.cpp:579-580, :229)
    const float hw = (float)f.W / 2.0f, hh = (float)f.H / 2.0f;
    v3p avg = splat3(V3(0.0f, 0.0f, 0.0f));
    // sub-ray stepping of Draw() (:566-596): both pixels of the lane share x, each half carries its own x1 / y1
    f2 y1 = { aa_start(ya, rs), aa_start(yb, rs) };
    for (int z = 0; z < rs; z++) {
    f2 x1 = splat2(aa_start(x, rs));
    for (int z2 = 0; z2 < rs; z2++) {
    const v3p d = V3P(x1 - splat2(hw), y1 - splat2(hh), splat2(f.focal));
    const v3p nd = neg3p(mat3_mul_vecp(f.rot, d));
    bool any0 = false, any1 = false;                                         // ClosestIntersection's return value
    {
        unsigned long long pm = pmask;
        ntests += (unsigned)__popcll(pm) * ((ok0 ? 1u : 0u) + (ok1 ? 1u : 0u));
        while (pm) {                                                         // ascending index: the `>=` rule holds
            const int j = __builtin_ctzll(pm);
            pm &= pm - 1ull;
            const float4 r0 = tb.cam[3 * j], r1 = tb.cam[3 * j + 1], r2 = tb.cam[3 * j + 2];
            const TestDots2 td = test_dots2(r0, r1, r2, nd);
            bool m0, m1;
            maybe_hit2(td, &m0, &m1);
            if (m0 || m1) {
                v3p hp;
                f2 dist;
                bool h0, h1;
                exact_hit_geo2(td, r0.w, tb.geo + 3 * j, cam, m0, m1, &h0, &h1, &hp, &dist);
                if (h0) {
                    any0 = true;
                    if (bd0 >= dist.x) { bd0 = dist.x; bi0 = j; pos0 = half0(hp); }      // :243-247
                }
                if (h1) {
                    any1 = true;
                    if (bd1 >= dist.y) { bd1 = dist.y; bi1 = j; pos1 = half1(hp); }
                }
            }
        }
    }
    const bool hit0 = ok0 && any0, hit1 = ok1 && any1;
    count_hits(f, (unsigned long long)(__popcll(__ballot(hit0)) + __popcll(__ballot(hit1))));
    if constexpr (!AA) {
        // Without supersampling the records are final here: the planes that carry them -- index, focal distance, the Intersection of the
        // `_ex` calls -- are written now instead of with the colour at the end, so that distance and hit point (eight registers for
        // outputs most callers do not ask for) are not held across the lights' loop.
        if (ok0) {
            const size_t px = (size_t)ya * f.W + x;
            if (f.index) f.index[px] = bi0;
            if (f.fd) f.fd[px] = bi0 >= 0 ? bd0 - f.focal_plane : 0.0f;            // focalDistances (:248-249)
            store_intersection(f, px, bi0, bd0, pos0);
        }
        if (ok1) {
            const size_t px = (size_t)yb * f.W + x;
            if (f.index) f.index[px] = bi1;
            if (f.fd) f.fd[px] = bi1 >= 0 ? bd1 - f.focal_plane : 0.0f;
            store_intersection(f, px, bi1, bd1, pos1);
        }
    }

    if (__any(hit0 || hit1)) {
        const v3p pos = join3(pos0, pos1);                                   // the record carried across sub-rays (:243-247)
        const int s0 = bi0 >= 0 ? bi0 : 0, s1 = bi1 >= 0 ? bi1 : 0;
        const float4 sa0 = tb.shade[2 * s0], sb0 = tb.shade[2 * s1];
        const v3p nDir = join3(V3(sa0.x, sa0.y, sa0.z), V3(sb0.x, sb0.y, sb0.z));   // glm::normalize(normal) (:300), per triangle
        v3p result = splat3(V3(0.0f, 0.0f, 0.0f)), result2 = result;
        for (int k = 0; k < f.nlights; k++) {
            // DirectLight's term before the shadow test (raytracer.cpp:294-304), both pixels at once
            // r = distance(pos, lightPos), A = 4 pi r^2, rDir = normalize(lightPos - pos), B = P / A with P = lightColor / samples (:296, divided on the host)
            const v3 L = ld3(f.lpos[k]);
            const LightGeometry2 lg = light_geometry2(pos, L, ld3(f.lcol[k]), f.lights_in_range != 0, hit0, hit1);
            const f2 r = lg.r;
            const v3p rd = lg.rDir, B = lg.B;
            const f2 dn = dot3p(rd, nDir);
            const f2 mx = { (dn.x < 0.0f) ? 0.0f : dn.x, (dn.y < 0.0f) ? 0.0f : dn.y };   // std::max(d, 0.0f)
            v3p D = scale3p(B, mx);
            const f2 thr = r * splat2(0.99f);                                // :313
            const float4 *tab = tb.light + (size_t)3 * n * k;
            // ---- shadow candidates: direction box of the wave's live rays, one lane per triangle ----
            const float inf = __builtin_huge_valf();
            const v3 lo = V3(wave_min_f(fminf(hit0 ? rd.x.x : inf, hit1 ? rd.x.y : inf)),
                             wave_min_f(fminf(hit0 ? rd.y.x : inf, hit1 ? rd.y.y : inf)),
                             wave_min_f(fminf(hit0 ? rd.z.x : inf, hit1 ? rd.z.y : inf)));
            const v3 hi = V3(wave_max_f(fmaxf(hit0 ? rd.x.x : -inf, hit1 ? rd.x.y : -inf)),
                             wave_max_f(fmaxf(hit0 ? rd.y.x : -inf, hit1 ? rd.y.y : -inf)),
                             wave_max_f(fmaxf(hit0 ? rd.z.x : -inf, hit1 ? rd.z.y : -inf)));
            bool sc = false;
            if (lane < n) sc = box_may_hit(tab[3 * lane], tab[3 * lane + 1], tab[3 * lane + 2], lo, hi);
            unsigned long long sm = __ballot(sc);
            ntests += (unsigned)__popcll(sm) * ((hit0 ? 1u : 0u) + (hit1 ? 1u : 0u));
            bool live0 = hit0, live1 = hit1;
            while (sm) {
                const int j = __builtin_ctzll(sm);
                sm &= sm - 1ull;
                const float4 r0 = tab[3 * j], r1 = tab[3 * j + 1], r2 = tab[3 * j + 2];
                const TestDots2 td = test_dots2(r0, r1, r2, rd);             // negD = rDir (:310, :229)
                bool m0, m1;
                maybe_hit2(td, &m0, &m1);
                m0 = m0 && live0; m1 = m1 && live1;
                if (m0 || m1) {
                    v3p hp;
                    f2 dist;
                    bool h0, h1;
                    exact_hit_geo2(td, r0.w, tb.geo + 3 * j, L, m0, m1, &h0, &h1, &hp, &dist);
                    if (h0 && dist.x < thr.x) {
                        live0 = false;                                        // occluded (:313-314); any-hit is exact
                        D.x.x = 0.0f; D.y.x = 0.0f; D.z.x = 0.0f;
                    }
                    if (h1 && dist.y < thr.y) {
                        live1 = false;
                        D.x.y = 0.0f; D.y.y = 0.0f; D.z.y = 0.0f;
                    }
                }
            }
            result = add3p(result, D);                                       // :319
            if ((k + 1) % f.samples == 0) result2 = add3p(result2, result);  // :322, after each light's samples
        }
        // (the triangles' colours only now, out of the LDS table: read beside the normals they were six registers held across the lights' loop)
        // (their rows' addresses derived afresh from the records: held, they would be two more registers across the loop)
        int c0 = bi0 >= 0 ? bi0 : 0, c1 = bi1 >= 0 ? bi1 : 0;
        asm volatile("" : "+v"(c0), "+v"(c1));
        const float4 sa1 = tb.shade[2 * c0 + 1], sb1 = tb.shade[2 * c1 + 1];
        const v3p tcol = join3(V3(sa1.x, sa1.y, sa1.z), V3(sb1.x, sb1.y, sb1.z));
        const v3p Dl = mul3p(result2, tcol);                                 // :325-326
        const v3p shaded = add3p(avg, mul3p(tcol, add3p(Dl, splat3(ld3(f.indirect)))));   // :584-591 (avgColor += R)
        avg = join3(hit0 ? half0(shaded) : half0(avg), hit1 ? half1(shaded) : half1(avg));
        if (AA) x1 = x1 + (f2){ hit0 ? aa_step(rs) : 0.0f, hit1 ? aa_step(rs) : 0.0f };   // :593, only after a hit
    }
    }   // z2
    if (AA) y1 = y1 + splat2(aa_step(rs));                                   // :596
    }   // z
    count_tests(f, ntests);
    if (AA) {                                                                // avgColor /= realSamples^2 (:599); /1 is the identity
        const f2 q = splat2((float)(rs * rs));
        avg = V3P(div2(avg.x, q), div2(avg.y, q), div2(avg.z, q));
    }
    // (the pixel's coordinates again, from the lane's number as the hardware counts it: two instructions instead of three registers
    // held from the top of the body to here -- with the planes above, what kept the kernel a register short of six waves per SIMD)
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int xe = x0 + (lane_e % TW), yae = y0 + (lane_e / TW), ybe = yae + TH;
    const bool ok0e = xe < f.W && yae < f.y1, ok1e = xe < f.W && ybe < f.y1;
    if (ok0e) {
        const v3 c = half0(avg);
        const size_t px = (size_t)yae * f.W + xe;
        if (f.rgb) st3(f.rgb + 3 * px, c);
        if constexpr (AA) {
            if (f.index) f.index[px] = bi0;
            if (f.fd) f.fd[px] = bi0 >= 0 ? bd0 - f.focal_plane : 0.0f;        // focalDistances (:248-249)
            store_intersection(f, px, bi0, bd0, pos0);
        }
        if (xe >= 1 && xe < f.W - 1 && yae >= 1 && yae < f.H - 1)            // :618-620
            f.xrgb[(size_t)(yae - f.row_origin) * f.pitch_words + xe] = pack_xrgb(c);
    }
    if (ok1e) {
        const v3 c = half1(avg);
        const size_t px = (size_t)ybe * f.W + xe;
        if (f.rgb) st3(f.rgb + 3 * px, c);
        if constexpr (AA) {
            if (f.index) f.index[px] = bi1;
            if (f.fd) f.fd[px] = bi1 >= 0 ? bd1 - f.focal_plane : 0.0f;
            store_intersection(f, px, bi1, bd1, pos1);
        }
        if (xe >= 1 && xe < f.W - 1 && ybe >= 1 && ybe < f.H - 1)
            f.xrgb[(size_t)(ybe - f.row_origin) * f.pitch_words + xe] = pack_xrgb(c);
    }
}

template <int TW, bool AA>
#ifndef MIRT_TILE_WAVES
#define MIRT_TILE_WAVES 6
#endif
// (six waves per SIMD -- 79 VGPRs -- without supersampling; the supersampling instantiation, which carries a second set of sub-ray state, four)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(AA ? 3 : MIRT_TILE_WAVES, AA ? 4 : MIRT_TILE_WAVES))) void k_rt_tile2(const RtTileFrame tf)
{
    extern __shared__ __attribute__((aligned(16))) float4 s_all[];
    const TileTables tab = tile_tables_load(tf, s_all);
    const long long ntiles = (long long)tf.tiles_x * tf.tiles_y;
    const int waves = blockDim.x >> 6;
    // (one tile per wave: the launch has a wave for every tile -- mirt_capi.hip --, and written as a loop over a wave's tiles
    // everything the body computes from the frame's parameters alone stays live across the whole body for a next round that never comes)
    const long long tile = (long long)blockIdx.x * waves + (threadIdx.x >> 6);
    if (tile < ntiles) tile_body2<TW, AA>(tf, (int)(tile % tf.tiles_x), (int)(tile / tf.tiles_x), tab);
}

template __global__ void k_rt_tile2<16, false>(const RtTileFrame);
template __global__ void k_rt_tile2<16, true>(const RtTileFrame);

}  // namespace mirt
