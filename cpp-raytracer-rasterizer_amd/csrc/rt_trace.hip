// rt_trace.hip -- the binned ray-trace kernel (k_rt_trace) and the tables it reads.
//
// Fused primary + shadow + shade + resolve over the binned candidates (rt_binned.hpp explains why the candidate
// reduction cannot change any result).  Workgroup = 4 wave64, each wave owns one 8x8-pixel tile = one camera bin.
//
// What made the first version (k_rt_binned, round 1) slow, measured on the 100 k soup at 1080p (tools/soup_stats.py):
// 139 VALU issue slots per ray-triangle test.  Nearly every candidate of a tile passes the conservative filter for SOME
// lane, so the wave ran the exact path -- three IEEE divides, a gather of the triangle, a square root, ~100 instructions
// -- once per candidate with one or two lanes active; and the shadow rays walked index lists with two dependent global
// loads per step (latency-bound: 87 us for 6.3 M tests).  This kernel separates the two stages:
//
//   filter stage   every lane runs the 3 dot products + the 7-instruction filter of rt_common.hpp against the tile's
//                  candidates (origin rows staged 64 at a time into the wave's LDS slice, broadcast reads); a (ray,
//                  candidate) pair the filter cannot reject is APPENDED to a wave-private LDS queue -- v_mbcnt ranks
//                  within the ballot, one ds_write_b128 {e1e2d, be2d, e1bd, e1e2b} + one ds_write_b64 {pixel, triangle};
//   exact stage    whenever 64 pairs are queued the wave drains them with ALL lanes busy: lane t takes pair t, does the
//                  reference's three divisions and accept test (raytracer.cpp:237-239), rebuilds the hit point from the
//                  geometry row (:241-242) and folds the result into the pixel's record with an LDS atomic --
//                  ds_min_u64 on the wavefront min-t key (distance bits << 32 | ~index: the `>=` tie rule of :243) for
//                  primary rays, a plain flag store for shadow rays (any-hit is exact, SURVEY A-5).
//
// Shadow rays read EXPANDED rows: the light-cube bins only depend on the scene and the light, so they are built once per
// (scene, lights) -- not per frame -- and stored bin-major as 48-byte origin rows with the triangle index in r2.w.  A lane walks
// its bin sequentially (lanes of one bin share every address), the next row is requested while the current one is
// tested, and nothing in the loop depends on an index load.
//
// Same filter and the same exact arithmetic as every other kernel => bit-identical results.
#include "rt_binned.hpp"

#include <float.h>

namespace mirt {

// ---- tables ---------------------------------------------------------------------------------------------------------

// geo[t] = {v0.xyz, e1.x | e1.yz, e2.xy | e2.z, 0, 0, 0}: what the accept path needs to rebuild the hit point
// pos = v0 + u*e1 + v*e2 (raytracer.cpp:216-217, :241).  Built once per scene upload.
__global__ __launch_bounds__(256) void k_geo_table(const float *__restrict__ tris15, int n, GeoRow *__restrict__ geo)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float *t15 = tris15 + (size_t)15 * t;
    const v3 v0 = ld3(t15), e1 = sub3(ld3(t15 + 3), v0), e2 = sub3(ld3(t15 + 6), v0);
    GeoRow g;
    g.g0 = make_float4(v0.x, v0.y, v0.z, e1.x);
    g.g1 = make_float4(e1.y, e1.z, e2.x, e2.y);
    g.g2 = make_float4(e2.z, 0.0f, 0.0f, 0.0f);
    geo[t] = g;
}

// rows[p] = origin row of triangle entries[p] for the light its bin belongs to, with the triangle index in r2.w: the sorted
// pair list of the light-cube frames, expanded so that a shadow ray reads its candidates sequentially.  bin_off points at the
// first light bin (the pairs of bins in front of it -- a camera frame binned in the same pass -- are not light pairs);
// bin_off[k * bins_per_light] is where the pairs of light k start.  pair_count / pair_cap (nullable): the pass that sized its
// list from an earlier frame's count leaves its tables untouched when the list overflowed -- nothing to expand then.
__global__ __launch_bounds__(256) void k_expand_light_rows(const uint32_t *__restrict__ bin_off, const uint32_t *__restrict__ entries,
                                                           int nlights, uint32_t bins_per_light,
                                                           const OriginRow *__restrict__ light_tab, int n,
                                                           LightRow *__restrict__ rows, const uint32_t *__restrict__ pair_count, uint32_t pair_cap)
{
    if (pair_count && *pair_count > pair_cap) return;
    const uint32_t first = bin_off[0], total = bin_off[(size_t)nlights * bins_per_light];
    for (uint32_t p = first + blockIdx.x * blockDim.x + threadIdx.x; p < total; p += gridDim.x * blockDim.x) {
        int k = 0;
        while (k + 1 < nlights && bin_off[(size_t)(k + 1) * bins_per_light] <= p) k++;
        const uint32_t tri = entries[p];
        LightRow r = light_tab[(size_t)k * n + tri];
        r.r2.w = __uint_as_float(tri);
        rows[p] = r;
    }
}

// ---- the wave's LDS slice -----------------------------------------------------------------------------------------------
#ifndef MIRT_TR_STAGE
#define MIRT_TR_STAGE 16
#endif
#ifndef MIRT_TR_DRAIN
#define MIRT_TR_DRAIN 64
#endif
// The kernel is latency-bound (dependent LDS / global round trips per candidate and per drain), so what a wave keeps in LDS
// decides how many waves hide each other's latency: 64 staged candidates + hit points cost 8.4 KB per wave = 4 workgroups
// per CU and 103 us on the 100 k soup; 16 staged candidates and (u, v) instead of the hit point: 5.4 KB, 7 workgroups, 84 us.
constexpr int TR_STAGE = MIRT_TR_STAGE;          // candidates staged per chunk (at most one per lane)
constexpr int TR_DRAIN = MIRT_TR_DRAIN;          // the exact stage runs when this many pairs are queued (<= 64)
constexpr int TR_QUEUE = TR_DRAIN + 64;          // queue slots: fewer than TR_DRAIN pairs are queued when a filter step appends at most 64
static_assert(TR_DRAIN >= 1 && TR_DRAIN <= 64, "a drain hands one pair to each lane");

struct TrWaveLds {
    float4 rows[TR_STAGE * 3];        // origin rows of the staged candidates
    float4 q[TR_QUEUE];               // {e1e2d, be2d, e1bd, e1e2b} of a queued (ray, candidate) pair
    unsigned long long best[64];      // wavefront min-t key of the pixel's closest accepted hit (this sub-ray)
    uint2 qa[TR_QUEUE];               // {pixel (lane) of the pair, triangle index | row index}
    float2 uv[64];                    // (u, v) of the hit that belongs to best[]: the hit point is rebuilt from them (:241)
    uint32_t idx[TR_STAGE];           // triangle ids of the staged candidates
    float thr[64];                    // shadow rays: r * 0.99f (:313)
    uint32_t flag[64];                // primary: some triangle was accepted (ClosestIntersection's return value); shadow: occluded
};
static_assert(sizeof(TrWaveLds) % 16 == 0, "per-wave LDS slice must keep 16-byte alignment");

__device__ __forceinline__ void wave_lds_fence()
{
    // wave-private LDS: the wave's own accesses execute in order; this only stops the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The exact stage for queue slots [0, count): raytracer.cpp:237-247.  geo + index * stride16 float4 = the pair's geometry row.
template <bool SHADOW>
__device__ __forceinline__ void tr_drain(TrWaveLds &s, int lane, int count, const float4 *__restrict__ geo, int stride16, v3 start)
{
    wave_lds_fence();
    if (lane < count) {
        const float4 e = s.q[lane];
        const uint2 a = s.qa[lane];
        const float t = e.w / e.x, u = e.y / e.x, v = e.z / e.x;                     // :237
        if (u + v <= 1.0f && u >= 0.0f && v >= 0.0f && t >= 0.0f) {                  // :239
            const float4 *g = geo + (size_t)a.y * stride16;
            const float4 g0 = g[0], g1 = g[1], g2 = g[2];
            const v3 v0 = V3(g0.x, g0.y, g0.z), e1 = V3(g0.w, g1.x, g1.y), e2 = V3(g1.z, g1.w, g2.x);
            const v3 p = add3(add3(v0, scale3(e1, u)), scale3(e2, v));              // :241
            const float dist = distance3(start, p);                                  // :242
            if (SHADOW) {
                if (dist < s.thr[a.x]) s.flag[a.x] = 1u;                             // :313-314
            } else {
                s.flag[a.x] = 1u;                                                    // `intersection = true` (:251)
                const unsigned long long key = min_t_key(dist, (int)a.y);
                atomicMin(&s.best[a.x], key);                                        // :243-247, order-free
                // the pair that holds the record now also owns the stored hit point (keys are unique per pixel: one
                // pair per triangle); a closer pair of a later drain overwrites both
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (s.best[a.x] == key) s.uv[a.x] = make_float2(u, v);
            }
        }
    }
    wave_lds_fence();
}

// Drains TR_DRAIN pairs and moves the rest of the queue to its front.
template <bool SHADOW>
__device__ __forceinline__ void tr_drain_full(TrWaveLds &s, int lane, int &qn, const float4 *__restrict__ geo, int stride16, v3 start)
{
    tr_drain<SHADOW>(s, lane, TR_DRAIN, geo, stride16, start);
    const int rest = qn - TR_DRAIN;
    float4 e = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint2 a = make_uint2(0u, 0u);
    if (lane < rest) { e = s.q[TR_DRAIN + lane]; a = s.qa[TR_DRAIN + lane]; }
    wave_lds_fence();
    if (lane < rest) { s.q[lane] = e; s.qa[lane] = a; }
    qn = rest;
    wave_lds_fence();
}

struct RtTraceFrame {
    RtFrame f;
    const uint32_t *cam_off;          // camera bins (8x8-pixel tiles): first entry of each, nbins + 1
    const uint32_t *cam_entries;      // triangle ids ordered by camera bin
    const GeoRow *geo;                // n geometry rows (k_geo_table)
    const uint32_t *light_off;        // light-cube bins of all light positions: first row of each, nlights*6*B*B + 1
    const LightRow *light_rows;       // expanded candidates ordered by light-cube bin (k_expand_light_rows)
    int tiles_x;                      // camera bins per row
    int cube_bins;                    // B: light-cube bins per face side
    int cam_shells;                   // depth shells per camera bin: bin b's list is cam_off[b * cam_shells] .. cam_off[(b + 1) * cam_shells]
    const uint32_t *pair_count;       // pairs this frame's binning produced / room in the pair list: beyond it the lists are
    uint32_t pair_cap;                // incomplete, the kernel does nothing and k_rt_brute_guard renders the frame
};

template <bool AA>
__global__ __launch_bounds__(256) void k_rt_trace(const RtTraceFrame tf)
{
    extern __shared__ __attribute__((aligned(16))) float4 s_all[];
    const RtFrame &f = tf.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    TrWaveLds &s = reinterpret_cast<TrWaveLds *>(s_all)[wave];

    const int tx = (int)blockIdx.x * 2 + (wave & 1);
    const int ty = f.y0 / BIN_TILE + (int)blockIdx.y * 2 + (wave >> 1);
    const int x = tx * BIN_TILE + (lane & 7), y = ty * BIN_TILE + (lane >> 3);
    const bool tile_ok = tx < tf.tiles_x && ty * BIN_TILE < f.y1;
    const bool ok = tile_ok && x < f.W && y >= f.y0 && y < f.y1;
    if (!tile_ok) return;                                  // wave-uniform
    if (__builtin_amdgcn_readfirstlane(*tf.pair_count) > tf.pair_cap) return;
    const v3 cam = ld3(f.cam);
    const int rs = AA ? f.aa : 1;                          // realSamples (:549-554); compile-time 1 without supersampling

    const uint32_t cbin = (uint32_t)ty * (uint32_t)tf.tiles_x + (uint32_t)tx;
    const uint32_t cbeg = tf.cam_off[(size_t)cbin * tf.cam_shells], cend = tf.cam_off[(size_t)(cbin + 1) * tf.cam_shells];
    const float4 *geo4 = reinterpret_cast<const float4 *>(tf.geo);
    const float4 *lrow4 = reinterpret_cast<const float4 *>(tf.light_rows);

    float best_d = FLT_MAX;                                // Update() reset (:335-339), once per frame
    int best_i = -1;
    v3 pos = V3(0.0f, 0.0f, 0.0f), avg = V3(0.0f, 0.0f, 0.0f);
    unsigned ntests = 0, ncand = 0;
    int qn = 0;                                            // queued pairs (wave-uniform)

    float y1 = aa_start(y, rs);                            // :566-569
    for (int z = 0; z < rs; z++) {
        float x1 = aa_start(x, rs);                        // :573-576
        for (int z2 = 0; z2 < rs; z2++) {
            // d = (x1 - W/2, y1 - H/2, focalLength); negD = -(cameraRot * d)   (raytracer.cpp:579-580, :229)
            const v3 d = V3(x1 - (float)f.W / 2.0f, y1 - (float)f.H / 2.0f, f.focal);
            const v3 nd = neg3(mat3_mul_vec(f.rot, d));

            // ---- primary ray: closest hit of THIS sub-ray among the tile's candidates ----
            // A candidate whose `near` bound (origin row r1.w: no hit point on it is closer to the camera) lies beyond the
            // sub-ray's current record cannot replace it, and the record already makes ClosestIntersection return true: the
            // lane does not queue it; once it lies beyond the record of EVERY pixel of the tile the wave does not even run
            // the filter.  The lists come roughly front to back (depth shells in the sort key), so after the first drains
            // most of a list is skipped.  The records are only updated by drains, i.e. the bounds lag -- never the result.
            s.best[lane] = MIN_T_NONE;
            s.flag[lane] = 0u;
            if (ok) ncand += cend - cbeg;
            float lane_best = FLT_MAX;                     // distance of the sub-ray's record so far
            float tile_best = FLT_MAX;                     // max of lane_best over the tile's pixels (wave-uniform)
            for (uint32_t base = cbeg; base < cend; base += TR_STAGE) {
                const int cnt = (int)min((uint32_t)TR_STAGE, cend - base);
                wave_lds_fence();                          // the previous chunk's row reads are done
                float my_near = 0.0f;
                if (lane < cnt) {
                    const uint32_t idx = tf.cam_entries[base + lane];
                    const float4 *src = reinterpret_cast<const float4 *>(f.cam_tab + idx);
                    const float4 a1 = src[1];
                    s.idx[lane] = idx;
                    s.rows[3 * lane] = src[0];
                    s.rows[3 * lane + 1] = a1;
                    s.rows[3 * lane + 2] = src[2];
                    my_near = a1.w;
                }
                wave_lds_fence();
                unsigned long long pm = __ballot(lane < cnt && !(my_near > tile_best));
                while (pm) {
                    const int j = __builtin_ctzll(pm);
                    pm &= pm - 1ull;
                    const float4 r0 = s.rows[3 * j], r1 = s.rows[3 * j + 1], r2 = s.rows[3 * j + 2];
                    const TestDots td = test_dots(r0, r1, r2, nd);
                    const bool live = ok && !(r1.w > lane_best);
                    if (live) ntests++;
                    const bool pass = live && maybe_hit(td);
                    const unsigned long long m = __ballot(pass);
                    if (m) {
                        const int at = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        if (pass) {
                            s.q[at] = make_float4(td.den, td.pu, td.qv, r0.w);
                            s.qa[at] = make_uint2((uint32_t)lane, s.idx[j]);
                        }
                        qn += __popcll(m);
                        if (qn >= TR_DRAIN) {
                            do tr_drain_full<false>(s, lane, qn, geo4, 3, cam); while (qn >= TR_DRAIN);    // (one pass when TR_DRAIN == 64)
                            lane_best = min_t_dist(s.best[lane]);
                            tile_best = wave_max_f(ok ? lane_best : -FLT_MAX);
                            pm &= __ballot(!(my_near > tile_best));
                        }
                    }
                }
            }
            if (qn) { tr_drain<false>(s, lane, qn, geo4, 3, cam); qn = 0; }
            // ... merged into the pixel's running record exactly as the sequential `>=` sweep would (:243): the
            // sub-ray's best replaces the record when it is at least as close (a later sub-ray wins exact ties)
            const unsigned long long skey = s.best[lane];
            const bool any = s.flag[lane] != 0u;           // ClosestIntersection's return value
            const float sd = min_t_dist(skey);
            if (any && best_d >= sd) {
                // the hit point, rebuilt as the exact stage built it: pos = v0 + u*e1 + v*e2 (:241) -- same operands, same bits
                const float2 suv = s.uv[lane];
                best_d = sd; best_i = min_t_index(skey);
                const float4 *gw = geo4 + (size_t)best_i * 3;
                const float4 g0 = gw[0], g1 = gw[1], g2 = gw[2];
                pos = add3(add3(V3(g0.x, g0.y, g0.z), scale3(V3(g0.w, g1.x, g1.y), suv.x)), scale3(V3(g1.z, g1.w, g2.x), suv.y));
            }

            const bool hit = ok && any;
            count_hits(f, (unsigned long long)__popcll(__ballot(hit)));
            if (__ballot(hit)) {
                const float *t = f.tris15 + (size_t)15 * (best_i >= 0 ? best_i : 0);
                const v3 nDir = normalize3(ld3(t + 9));            // (:300)
                const v3 tcol = ld3(t + 12);
                v3 result = V3(0.0f, 0.0f, 0.0f), result2 = V3(0.0f, 0.0f, 0.0f);
                for (int k = 0; k < f.nlights; k++) {
                    const v3 L = ld3(f.lpos[k]);
                    v3 rd;
                    float r;
                    v3 D = light_term(f, k, pos, nDir, &rd, &r);
                    wave_lds_fence();
                    const float thr = r * 0.99f;                   // (:313)
                    s.thr[lane] = thr;
                    s.flag[lane] = 0u;
                    uint32_t e = 0, end = 0;
                    if (hit) {
                        const uint32_t bin = cube_bin_of(rd, (uint32_t)k * 6u * (uint32_t)(tf.cube_bins * tf.cube_bins), tf.cube_bins);
                        e = tf.light_off[bin]; end = tf.light_off[bin + 1];
                    }
                    bool act = e < end;
                    ncand += end - e;
                    float4 c0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), c1 = c0, c2 = c0;
                    if (act) { const float4 *src = lrow4 + (size_t)e * 3; c0 = src[0]; c1 = src[1]; c2 = src[2]; }
                    while (__ballot(act)) {
                        // request the next row before this one is tested
                        const bool nact = act && (e + 1 < end);
                        float4 n0 = c0, n1 = c1, n2 = c2;
                        if (nact) { const float4 *src = lrow4 + (size_t)(e + 1) * 3; n0 = src[0]; n1 = src[1]; n2 = src[2]; }
                        const TestDots td = test_dots(c0, c1, c2, rd);           // negD = rDir (:310, :229)
                        // a candidate none of whose points is closer to the light than 0.99 r cannot occlude (:313)
                        const bool pass = act && !(c1.w > thr) && maybe_hit(td);
                        if (act) ntests++;
                        const unsigned long long m = __ballot(pass);
                        bool occ = false;
                        if (m) {
                            const int at = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                            if (pass) {
                                s.q[at] = make_float4(td.den, td.pu, td.qv, c0.w);
                                s.qa[at] = make_uint2((uint32_t)lane, __float_as_uint(c2.w));     // the candidate's triangle (r2.w)
                            }
                            qn += __popcll(m);
                            if (qn >= TR_DRAIN) {
                                do tr_drain_full<true>(s, lane, qn, geo4, 3, L); while (qn >= TR_DRAIN);
                                occ = s.flag[lane] != 0u;                        // a lane found occluded stops walking
                            }
                        }
                        c0 = n0; c1 = n1; c2 = n2;
                        e++;
                        act = nact && !occ;
                    }
                    if (qn) { tr_drain<true>(s, lane, qn, geo4, 3, L); qn = 0; }
                    if (s.flag[lane] != 0u) D = V3(0.0f, 0.0f, 0.0f);           // occluded (:313-314); any-hit is exact
                    result = add3(result, D);                      // (:319)
                    if ((k + 1) % f.samples == 0) result2 = add3(result2, result);   // (:322) after each light's samples
                }
                if (hit) {
                    const v3 Dl = mul3(result2, tcol);             // (:325-326)
                    const v3 T = add3(Dl, ld3(f.indirect));        // (:584-586)
                    avg = add3(avg, mul3(tcol, T));                // (:587-591)
                    x1 += aa_step(rs);                             // (:593) only after a hit
                }
            }
            wave_lds_fence();
        }
        y1 += aa_step(rs);                                         // (:596)
    }
    count_tests(f, ntests);
    count_candidates(f, ncand);
    if (!ok) return;
    avg = div3s(avg, (float)(rs * rs));                            // (:599)
    const size_t px = (size_t)y * f.W + x;
    if (f.rgb) st3(f.rgb + 3 * px, avg);
    if (f.index) f.index[px] = best_i;
    if (f.fd) f.fd[px] = best_i >= 0 ? best_d - f.focal_plane : 0.0f;          // focalDistances (:248-249)
    store_intersection(f, px, best_i, best_d, pos);
    if (x >= 1 && x < f.W - 1 && y >= 1 && y < f.H - 1)            // (:618-620)
        f.xrgb[(size_t)(y - f.row_origin) * f.pitch_words + x] = pack_xrgb(avg);
}

template __global__ void k_rt_trace<false>(const RtTraceFrame);
template __global__ void k_rt_trace<true>(const RtTraceFrame);

size_t rt_trace_lds_bytes() { return 4 * sizeof(TrWaveLds); }

}  // namespace mirt
