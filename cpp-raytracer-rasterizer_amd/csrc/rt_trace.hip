// rt_trace.hip -- the binned ray-trace kernel (k_rt_trace2) and the tables it reads.
//
// Fused primary + shadow + shade + resolve over the binned candidates (rt_binned.hpp explains why the candidate
// reduction cannot change any result).  Workgroup = 4 wave64 = a 32 x 16-pixel block; each wave owns TWO horizontally
// adjacent 8 x 8-pixel tiles (= two camera bins), lane l carrying pixel (l & 7, l >> 3) of tile A and of tile B.
//
// Round 2's kernel (one tile per wave, one ray per lane) spent 630 of its 1140 VALU instructions per wave on code that runs
// once per tile whatever the lists hold -- ray set-up, the two partial drains, the light term with its square roots and
// divisions, pack and store -- and walked each shadow ray through every triangle its light-cube bin holds, near or far.
// This kernel
//   * runs that per-tile code once per PAIR of tiles in packed FP32 (mirt_math2.hpp: v_pk_mul/add/fma round each half like
//     their scalar twins; div2 shares the multiply-adds of two IEEE divisions);
//   * filter stage, primary rays: the candidates of tile A and of tile B are staged INTERLEAVED in the wave's LDS slice
//     ({A.x, B.x, A.y, B.y} ...), so one ds_read_b128 delivers the register pairs the packed dot products want: 15 packed
//     instructions test candidate j of list A against pixel A and candidate j of list B against pixel B; a (ray, candidate)
//     pair the 7-instruction filter of rt_common.hpp cannot reject is APPENDED to a wave-private LDS queue;
//   * exact stage: whenever 64 pairs are queued the wave drains them with ALL lanes busy: lane t takes pair t, does the
//     reference's three divisions and accept test (raytracer.cpp:237-239), rebuilds the hit point from the geometry row
//     (:241-242) and folds the result into the pixel's record with an LDS atomic -- ds_min_u64 on the wavefront min-t key
//     (distance bits << 32 | ~index: the `>=` tie rule of :243) for primary rays, a flag store for shadow rays (any-hit is
//     exact, SURVEY A-5);
//   * shadow rays: the light-cube bins are ordered by DEPTH SHELL of the candidates' `near` bound (sort key = bin * shells +
//     shell), so a ray whose hit point lies at distance r from the light walks only the shells up to the one 0.99 r falls
//     into -- every later candidate has near > 0.99 r and cannot occlude (:313).  A bin's list grows with the square of the
//     distance from the light, so this drops most of it.  A lane walks the list of its pixel A and then that of its pixel B
//     back to back (the wave's steps are max(lenA + lenB) over its lanes, not max(lenA) + max(lenB));
//   * a frame whose pair list overflowed (sized from an earlier frame's count, mirt_capi.hip) is rendered by the SAME kernel
//     with "every triangle" as each tile's list -- brute force, same bits -- instead of by a guard launch behind it;
//   * workgroups are mapped to screen blocks so that the blocks an XCD (private L2) works on are neighbours.
//
// Same filter and the same exact arithmetic as every other kernel => bit-identical results.
#include "rt_binned.hpp"

#include <float.h>

namespace mirt {

// ---- tables ---------------------------------------------------------------------------------------------------------

// geo[t] = {v0.xyz, e1.x | e1.yz, e2.xy | e2.z, 0, 0, 0}: what the accept path needs to rebuild the hit point
// pos = v0 + u*e1 + v*e2 (raytracer.cpp:216-217, :241).  Built once per scene upload.
__global__ __launch_bounds__(256) void k_geo_table(const float *__restrict__ tris15, int n, GeoRow *__restrict__ geo, ShadeRow *__restrict__ shade)
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
    const v3 nd = normalize3(ld3(t15 + 9));                                  // (:300)
    ShadeRow sr;
    sr.n = make_float4(nd.x, nd.y, nd.z, 0.0f);
    sr.col = make_float4(t15[12], t15[13], t15[14], 0.0f);
    shade[t] = sr;
}

// rows[p] = origin row of triangle entries[p] for the light its key belongs to, with the `far` bound of that triangle from that
// light in r2.w (origin_far: no point of it is farther; k_select_faces computes it where the vertices are at hand -- this kernel
// used to fetch them again per row, uncoalesced), and row_tri[p] = the triangle (for the few pairs that reach the exact
// stage): the sorted pair list of the light-cube frames, expanded so that a shadow ray reads its candidates sequentially.  bin_off points at the
// first light key (the pairs of keys in front of it -- a camera frame binned in the same pass -- are not light pairs);
// bin_off[k * keys_per_light] is where the pairs of light k start (keys_per_light = 6 * B * B * depth shells).  pair_count /
// pair_cap (nullable): the pass that sized its list from an earlier frame's count leaves its tables untouched when the list
// overflowed -- nothing to expand then.
__global__ __launch_bounds__(256) void k_expand_light_rows(const uint32_t *__restrict__ bin_off, const uint32_t *__restrict__ entries,
                                                           int nlights, uint32_t keys_per_light,
                                                           const OriginRow *__restrict__ light_tab, int n,
                                                           LightRow *__restrict__ rows, const uint32_t *__restrict__ pair_count, uint32_t pair_cap,
                                                           uint32_t *__restrict__ row_tri /* nullable: entries itself serves */)
{
    if (pair_count && *pair_count > pair_cap) return;
    const uint32_t first = bin_off[0], total = bin_off[(size_t)nlights * keys_per_light];
    for (uint32_t p = first + blockIdx.x * blockDim.x + threadIdx.x; p < total; p += gridDim.x * blockDim.x) {
        int k = 0;
        while (k + 1 < nlights && bin_off[(size_t)(k + 1) * keys_per_light] <= p) k++;
        const uint32_t tri = entries[p];
        rows[p] = light_tab[(size_t)k * n + tri];          // (r2.w = the `far` bound: k_select_faces put it there, per (triangle, light))
        if (row_tri) row_tri[p] = tri;
    }
}

// ---- the wave's LDS slice -----------------------------------------------------------------------------------------------
#ifndef MIRT_TR_STAGE
#define MIRT_TR_STAGE 16
#endif
// What a wave keeps in LDS decides how many waves share a CU and what else fits beside them (round 2: 8.4 -> 5.4 KB per wave
// took the frame from 103 to 84 us; round 3, two tiles per wave: 9728 -> 8064 bytes lets the sort's workgroups run beside
// the trace waves of the frames in flight): 16 staged candidates per tile with their geometry rows, a queue of 96 pairs.
constexpr int TR_STAGE = MIRT_TR_STAGE;          // candidates staged per chunk and tile (lanes 0..15 stage tile A's, 16..31 tile B's)
constexpr int TR_DRAIN = 64;                     // the exact stage runs when this many pairs are queued: one pair per lane
constexpr int TR_QUEUE = 96;                     // queue slots: fewer than TR_DRAIN are queued when a step appends (<= 64 per tile; a step
                                                 // whose pairs do not fit drains first, tile by tile if need be)
constexpr int TR_PIX = 128;                      // pixels of a wave: tile A = 0..63, tile B = 64..127
static_assert(TR_STAGE == 16, "the staging lanes are split 16 / 16 between the two tiles");

// 8064 bytes per wave: 20 waves (5 per SIMD) fit the CU's 160 KB.
struct TrWaveLds {
    float4 rows[TR_STAGE * 6];        // origin rows of the staged candidates, tile A's and tile B's interleaved float by float
    float4 geo[TR_STAGE * 2 * 3];     // their geometry rows: slot = 16 * tile + position in the chunk
    float4 q[TR_QUEUE];               // {e1e2d, be2d, e1bd, e1e2b} of a queued (ray, candidate) pair
    unsigned long long best[TR_PIX];  // wavefront min-t key of the pixel's closest accepted hit (this sub-ray)
    uint2 qa[TR_QUEUE];               // {pixel of the pair | staging slot of its candidate << 8 (primary rays), triangle index}
    union {
        float px[TR_PIX];             // primary rays: the hit point that belongs to best[] (:241), read into registers at the merge ...
        float thr[TR_PIX];            // ... shadow rays, afterwards: r * 0.99f (:313)
    };
    float py[TR_PIX], pz[TR_PIX];
    uint8_t flag[TR_PIX];             // primary: some triangle was accepted (ClosestIntersection's return value); shadow: occluded
};
static_assert(sizeof(TrWaveLds) % 16 == 0, "per-wave LDS slice must keep 16-byte alignment");

__device__ __forceinline__ void wave_lds_fence()
{
    // wave-private LDS: the wave's own accesses execute in order; this only stops the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The exact stage for queue slots [0, count): raytracer.cpp:237-247.  The pair's geometry row: shadow rays fetch it (geo + 3 *
// triangle float4); primary rays find it in the wave's LDS slice, staged with the candidate's origin row -- no memory round
// trip inside the drain (their queue never outlives the staged chunk).
template <bool SHADOW>
__device__ __forceinline__ void tr_drain(TrWaveLds &s, int lane, int count, const float4 *__restrict__ geo, v3 start,
                                         const uint32_t *__restrict__ row_tri = nullptr, bool lazy_geo = false)
{
    wave_lds_fence();
    if (lane < count) {
        const float4 e = s.q[lane];
        uint2 a = s.qa[lane];
        float t, u, v;
        div3_sel(e.w, e.y, e.z, e.x, __builtin_amdgcn_ballot_w64(exact_quotients_outside(e.w, e.y, e.z, e.x)), t, u, v);   // :237: three quotients over e1e2d (queued pairs passed the filter)
        if (u + v <= 1.0f && u >= 0.0f && v >= 0.0f && t >= 0.0f) {                  // :239
            float4 g0, g1, g2;
            if (SHADOW) {
                const uint32_t tri = row_tri ? row_tri[a.y] : a.y;         // (a.y = the candidate's row; the origin tables' rows are the triangles)
                const float4 *g = geo + (size_t)tri * 3;
                g0 = g[0]; g1 = g[1]; g2 = g[2];
            } else if (lazy_geo) {
                const float4 *g = geo + (size_t)a.y * 3;                   // (a.y = the triangle)
                g0 = g[0]; g1 = g[1]; g2 = g[2];
                a.x &= 0xFFu;
            } else {
                const float4 *g = s.geo + (a.x >> 8) * 3;
                g0 = g[0]; g1 = g[1]; g2 = g[2];
                a.x &= 0xFFu;
            }
            const v3 v0 = V3(g0.x, g0.y, g0.z), e1 = V3(g0.w, g1.x, g1.y), e2 = V3(g1.z, g1.w, g2.x);
            const v3 p = add3(add3(v0, scale3(e1, u)), scale3(e2, v));              // :241
            const float dist = distance3(start, p);                                  // :242
            if (SHADOW) {
                if (dist < s.thr[a.x]) s.flag[a.x] = 1;                             // :313-314
            } else {
                s.flag[a.x] = 1;                                                     // `intersection = true` (:251)
                const unsigned long long key = min_t_key(dist, (int)a.y);
                atomicMin(&s.best[a.x], key);                                        // :243-247, order-free
                // the pair that holds the record now also owns the stored hit point (keys are unique per pixel: one
                // pair per triangle); a closer pair of a later drain overwrites both
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (s.best[a.x] == key) { s.px[a.x] = p.x; s.py[a.x] = p.y; s.pz[a.x] = p.z; }
            }
        }
    }
    wave_lds_fence();
}

// Drains TR_DRAIN pairs and moves the rest of the queue to its front.
template <bool SHADOW>
__device__ __forceinline__ void tr_drain_full(TrWaveLds &s, int lane, int &qn, const float4 *__restrict__ geo, v3 start,
                                              const uint32_t *__restrict__ row_tri = nullptr, bool lazy_geo = false)
{
    tr_drain<SHADOW>(s, lane, TR_DRAIN, geo, start, row_tri, lazy_geo);
    const int rest = qn - TR_DRAIN;
    float4 e = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint2 a = make_uint2(0u, 0u);
    if (lane < rest) { e = s.q[TR_DRAIN + lane]; a = s.qa[TR_DRAIN + lane]; }
    wave_lds_fence();
    if (lane < rest) { s.q[lane] = e; s.qa[lane] = a; }
    qn = rest;
    wave_lds_fence();
}

// the wave's mask of a predicate, straight from the compare (__ballot goes through a 0/1 VGPR and a second compare)
__device__ __forceinline__ unsigned long long wballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

__device__ __forceinline__ int wave_rank(unsigned long long m)      // lanes below this one that are set in m
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// cube_bin_of (rt_binned.hpp) for the two shadow rays of a lane: the same selections per half.
__device__ __forceinline__ void cube_bin_of2(const v3p &rd, uint32_t face_base0, int cube_bins, uint32_t *bin0, uint32_t *bin1)
{
    f2 a, b, m;
    int face[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const float x = h ? rd.x.y : rd.x.x, y = h ? rd.y.y : rd.y.x, z = h ? rd.z.y : rd.z.x;
        const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
        int k;
        float mm, aa, bb, sgn;
        if (ax >= ay && ax >= az) { k = 0; mm = ax; sgn = x; aa = y; bb = z; }
        else if (ay >= az) { k = 1; mm = ay; sgn = y; aa = z; bb = x; }
        else { k = 2; mm = az; sgn = z; aa = x; bb = y; }
        face[h] = 2 * k + (sgn < 0.0f ? 1 : 0);
        if (h) { a.y = aa; b.y = bb; m.y = mm; } else { a.x = aa; b.x = bb; m.x = mm; }
    }
    // u, v in [-1,1]; NaN (degenerate ray, never accepted by any triangle) falls into bin 0.  The quotients only CHOOSE the bin: one
    // reciprocal per ray (1 ulp) and two products put u, v within 2^-22 of the exact ones, and a bin's list holds every triangle a
    // ray within 2^-18 of its rectangle can hit (the cubes' pad, fill_light_frames) -- so the bin a ray lands in, this one or its
    // neighbour across an edge it grazes, has the ray's candidates either way.  (Four IEEE divisions here were 4 % of the kernel's
    // vector instructions.)
    const f2 im = { __builtin_amdgcn_rcpf(m.x), __builtin_amdgcn_rcpf(m.y) };
    const f2 u = a * im, v = b * im;
    const float half = 0.5f * (float)cube_bins;
    const uint32_t per_face = (uint32_t)(cube_bins * cube_bins);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const float uu = h ? u.y : u.x, vv = h ? v.y : v.x;
        int i = (int)floorf((uu + 1.0f) * half);
        int j = (int)floorf((vv + 1.0f) * half);
        i = min(max(i, 0), cube_bins - 1);
        j = min(max(j, 0), cube_bins - 1);
        const uint32_t bin = face_base0 + (uint32_t)face[h] * per_face + (uint32_t)j * cube_bins + (uint32_t)i;
        if (h) *bin1 = bin; else *bin0 = bin;
    }
}

// A tile pair as the trace kernel's waves pick it up: the first tile's camera bin, where its list starts (the second tile's
// follows it), and the two lengths.
struct TilePairRec { uint32_t tile, beg, nA, nB; };
constexpr int ORDER_CLASSES = 8;                 // classes of max(nA, nB) / ORDER_CLASS_STEP, the last one open-ended
constexpr int ORDER_GROUPS = 8;                  // one list per XCD group (workgroup id % 8)
constexpr uint32_t ORDER_CLASS_STEP = 16;        // = TR_STAGE: a class is a number of staging chunks

struct RtTraceFrame {
    RtFrame f;
    const uint32_t *cam_off;          // camera bins (8x8-pixel tiles): first entry of each key, nbins * cam_shells + 1
    const uint32_t *cam_entries;      // triangle ids ordered by camera key
    const GeoRow *geo;                // n geometry rows (k_geo_table)
    const ShadeRow *shade;            // n shading rows (k_geo_table)
    const uint32_t *light_off;        // light-cube keys of all light positions: first row of each, nlights*6*B*B*light_shells + 1
    const LightRow *light_rows;       // expanded candidates ordered by light-cube key (k_expand_light_rows)
    const uint32_t *light_tri;        // the triangle of each of them
    const BinFrameDesc *light_frames; // 6 per light position: the depth-shell parameters (the faces of a light share them)
    int tiles_x;                      // camera bins per row
    int cube_bins;                    // B: light-cube bins per face side
    int cam_shells;                   // depth shells per camera bin: bin b's list is cam_off[b * cam_shells] .. cam_off[(b + 1) * cam_shells]
    int light_shells;                 // depth shells per light-cube bin
    const uint32_t *pair_count;       // pairs this frame's binning produced / room in the pair list: beyond it the lists are
    uint32_t pair_cap;                // incomplete and every tile takes the whole triangle list instead (brute force)
    const TilePairRec *order;         // the frame's tile pairs by XCD group and list-length class (k_tile_order): ORDER_GROUPS x ORDER_CLASSES
                                      // segments of order_seg records
    const uint32_t *order_count;      // records in each segment
    uint32_t order_seg;               // room per segment = the most pairs a group can have = waves per group
    const uint32_t *sel;              // the triangles k_prep_select found the frame may see, and how many: what a frame that fell
    const uint32_t *sel_count;        // back to brute force walks (their origin rows are the ones the frame has built)
    int lazy_geo;                     // 1: geometry rows are fetched by the exact stage, for the pairs it accepts, instead of being staged
                                      // with every candidate (scenes whose tables no longer fit the caches: see the staging code)
    const uint32_t *light_pair_count; // the same for the light-cube lists when a pass of their own built them (nullable: the shared,
    uint32_t light_pair_cap;          // cached cube never overflows): beyond it the shadow rays walk the lights' origin tables
};

// One wave renders one PAIR of horizontally adjacent tiles.  Two things decide which wave takes which pair:
//  * Tiles differ a lot in what they cost -- on the 100 k soup 9 % of them hold half of all candidates, ~90 each against an
//    average of 14 -- and a wave that starts such a pair late is what the whole launch then waits for (measured: the longest wave
//    lived 60 us of a launch that would take 58 us with every wave slot always full, and took 85).  So pairs are filed by the
//    length of their longer list into ORDER_CLASSES classes and taken longest class first: the classic greedy schedule.
//  * Consecutive workgroup ids go to consecutive XCDs, each with an L2 of its own (4 MiB).  With the pairs of the whole frame
//    in one longest-first list every XCD touched every origin row and geometry row of the frame: 87 MB of fabric reads per
//    launch for 9.6 MB of tables (L2 hit rate 0.39).  So the frame's pairs of tile rows are dealt to the eight XCD groups in
//    turn (group = (tile row / 4) % 8, stripes of 32 pixel rows: every group samples the whole height of the frame, so they stay balanced) and each
//    group has its OWN longest-first list: a triangle of the soup is then wanted by ~2 groups instead of 8, and a group's
//    share of the tables fits its L2.
// k_tile_order builds the lists: one wave per 64 pairs of one tile row (so the group is wave-uniform), one global atomic
// instruction per wave (lane c adds the wave's count of class c to the group's counter).  A record carries the pair's list
// bounds, so the trace wave's first load is its last indirection.  counters: [0] = pairs the binning pass produced (> pair_cap:
// the lists are incomplete, the frame is brute force and every record goes to class 0), [16 + 8 * group + c] = records of
// class c of the group (zeroed by k_prep_origin).
__global__ __launch_bounds__(64) void k_tile_order(const uint32_t *__restrict__ cam_off, int cam_shells, int tiles_x, int j0, int j1,
                                                   uint32_t *__restrict__ counters, uint32_t pair_cap, TilePairRec *__restrict__ order, uint32_t seg)
{
    const int pairs_x = (tiles_x + 1) / 2, lane = threadIdx.x;
    const int ty = j0 + (int)blockIdx.y, px = (int)blockIdx.x * 64 + lane;
    const uint32_t group = ((uint32_t)ty >> ORDER_STRIPE_SHIFT) & (ORDER_GROUPS - 1);
    TilePairRec r = { 0u, 0u, 0u, 0u };
    int cls = -1;
    if (px < pairs_x) {
        r.tile = (uint32_t)(ty * tiles_x + 2 * px);
        cls = 0;
        if (counters[0] <= pair_cap) {
            const uint32_t b0 = cam_off[(size_t)r.tile * cam_shells], b1 = cam_off[(size_t)(r.tile + 1) * cam_shells];
            r.beg = b0; r.nA = b1 - b0;
            r.nB = 2 * px + 1 < tiles_x ? cam_off[(size_t)(r.tile + 2) * cam_shells] - b1 : 0u;
            cls = (int)min(max(r.nA, r.nB) / ORDER_CLASS_STEP, (uint32_t)(ORDER_CLASSES - 1));
        }
    }
    unsigned long long m[ORDER_CLASSES];
    uint32_t mine = 0;
#pragma unroll
    for (int c = 0; c < ORDER_CLASSES; c++) {
        m[c] = __builtin_amdgcn_ballot_w64(cls == c);
        if (lane == c) mine = (uint32_t)__popcll(m[c]);
    }
    uint32_t base = 0;
    if (lane < ORDER_CLASSES && mine) base = atomicAdd(&counters[16 + group * ORDER_CLASSES + lane], mine);
#pragma unroll
    for (int c = 0; c < ORDER_CLASSES; c++) {
        const uint32_t bc = (uint32_t)__builtin_amdgcn_readlane((int)base, c);
        if (cls == c) order[((size_t)group * ORDER_CLASSES + c) * seg + bc + (uint32_t)__popcll(m[c] & ((1ull << lane) - 1ull))] = r;
    }
}

// waves per SIMD the register allocator aims for: 4 (<= 128 VGPRs) -- the supersampling variant, which carries a second set of
// sub-ray state, may take 3 rather than spill
#ifndef MIRT_TR_WAVES
#define MIRT_TR_WAVES 4
#endif
#ifndef MIRT_TR_WAVES_MIN
#define MIRT_TR_WAVES_MIN 3
#endif
// STATS: the kernel's own counts (tests executed, candidates offered, wave steps of the two filter loops, drains: what
// mirt_get_stats reports and the roofline's `achieved` is made of).  Keeping them costs five scalar instructions in every step of
// both filter loops -- a fifth of the scalar stream --, so the frames of a render loop run without them; a frame rendered with
// profiling on (mirt_set_profiling: the frames whose kernel times bench.py reads) counts.  The hit count -- the frame's shadow rays
// -- is one atomic per wave and always kept.
// ... and FIVE (<= 96 VGPRs) for the instantiation the large scenes' frames run: the 1 M-triangle frame at 8K gains 4-5 % from the fifth
// wave (1.09 against 1.13-1.17 ms per frame), the 100 k-triangle frame with four in flight loses 2 % to it (65.3 against 63.9 us: five
// trace waves per SIMD leave the binning kernels of the frames beside it less room), so mirt_capi.hip launches it from 400 k
// triangles on and for the frame that runs alone (115.6 against 117.6 us).  Asking for 10 KiB of LDS per wave to hold the 96-register
// code at four waves was tried instead of a second instantiation: 66.7 us per frame -- the LDS the trace waves then sit on is what
// the binning kernels of the frames beside them need.  (Round 3 and the first half of round 4 "measured" a fifth wave with builds of 98 VGPRs -- which the hardware runs at
// four, registers being handed out in eights -- or with spills; the kernel fits 96 since the shaded triangles' colours are fetched
// after the shadow walk instead of being held across it.)
template <bool AA, bool STATS, int WAVES = MIRT_TR_WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES == 5 ? 5 : MIRT_TR_WAVES_MIN, WAVES))) void k_rt_trace2(const RtTraceFrame tf)
{
    extern __shared__ __attribute__((aligned(16))) float4 s_all[];
    const RtFrame &f = tf.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    TrWaveLds &s = reinterpret_cast<TrWaveLds *>(s_all)[wave];

    // Wave -> tile pair: record w of its XCD group's list (k_tile_order), longest lists first.  (Waves never synchronise with each other, so a
    // workgroup is just a scheduling unit of 1, 2 or 4 of them.)
    const uint32_t group = blockIdx.x & (ORDER_GROUPS - 1);                       // = this workgroup's XCD group
    uint32_t w = (blockIdx.x >> 3) * (blockDim.x >> 6) + (uint32_t)wave;          // this wave among the group's
    TilePairRec rec = { 0u, 0u, 0u, 0u };
    {
        bool found = false;
#pragma unroll
        for (int c = ORDER_CLASSES - 1; c >= 0; c--) {
            const uint32_t cnt = tf.order_count[group * ORDER_CLASSES + c];
            if (!found && w < cnt) { rec = tf.order[((size_t)group * ORDER_CLASSES + c) * tf.order_seg + w]; found = true; }
            if (!found) w -= cnt;
        }
        if (!found) return;                                // beyond the group's pairs
    }
    rec.tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec.tile); rec.beg = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec.beg);
    rec.nA = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec.nA); rec.nB = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec.nB);
    const int txA = (int)(rec.tile % (uint32_t)tf.tiles_x), ty = (int)(rec.tile / (uint32_t)tf.tiles_x);
    const bool enB = txA + 1 < tf.tiles_x;
    const int xA = txA * BIN_TILE + (lane & 7), xB = xA + BIN_TILE, y = ty * BIN_TILE + (lane >> 3);
    const bool okY = y >= f.y0 && y < f.y1;
    const bool okA = okY && xA < f.W, okB = enB && okY && xB < f.W;
    const unsigned long long okmA = wballot(okA), okmB = wballot(okB);
    const unsigned nokA = (unsigned)__popcll(okmA), nokB = (unsigned)__popcll(okmB);
#ifdef MIRT_TR_TIMING
    // (experiments) where a wave's lifetime goes: TM_SEG(i) adds the time since the previous stamp to segment i
    const long long tm0 = __builtin_amdgcn_s_memtime();
    long long tm_last = tm0, tm_seg[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#define TM_SEG(i) { const long long now_ = __builtin_amdgcn_s_memtime(); tm_seg[i] += now_ - tm_last; tm_last = now_; }
#else
#define TM_SEG(i)
#endif
    // the pair list overflowed (it was sized from an earlier frame's count): no lists -- every tile takes every triangle
    const bool brute = (uint32_t)__builtin_amdgcn_readfirstlane((int)*tf.pair_count) > tf.pair_cap;
    // ... and the light lists: their own pass's count where there was one
    const bool brute_l = tf.light_pair_count ? (uint32_t)__builtin_amdgcn_readfirstlane((int)*tf.light_pair_count) > tf.light_pair_cap : false;
    const v3 cam = ld3(f.cam);
    const int rs = AA ? f.aa : 1;                          // realSamples (:549-554); compile-time 1 without supersampling

    const uint32_t begA = rec.beg, begB = rec.beg + rec.nA;
    const uint32_t nall = brute ? min((uint32_t)__builtin_amdgcn_readfirstlane((int)*tf.sel_count), (uint32_t)f.n) : 0u;
    const uint32_t nA = brute ? nall : rec.nA, nB = brute ? (enB ? nall : 0u) : rec.nB;
    const uint32_t nmax = max(nA, nB);
    const float4 *geo4 = reinterpret_cast<const float4 *>(tf.geo);
    const bool lazy = tf.lazy_geo != 0;
    TM_SEG(0)

    float bdA = FLT_MAX, bdB = FLT_MAX;                    // Update() reset (:335-339), once per frame
    int biA = -1, biB = -1;
    v3 posA = V3(0.0f, 0.0f, 0.0f), posB = posA;
    v3p avg = splat3(V3(0.0f, 0.0f, 0.0f));
    unsigned ntests = 0, ncand = 0, nsteps_p = 0, nsteps_s = 0, ndrains = 0;     // wave-uniform counters
    unsigned ncand_l = 0;                                  // per lane: shadow candidates offered
    int qn = 0;                                            // queued pairs (wave-uniform)
    const float hw = (float)f.W / 2.0f, hh = (float)f.H / 2.0f;

    f2 y1 = splat2(aa_start(y, rs));                       // :566-569
    for (int z = 0; z < rs; z++) {
        f2 x1 = { aa_start(xA, rs), aa_start(xB, rs) };    // :573-576
        for (int z2 = 0; z2 < rs; z2++) {
            // d = (x1 - W/2, y1 - H/2, focalLength); negD = -(cameraRot * d)   (raytracer.cpp:579-580, :229)
            const v3p d = V3P(x1 - splat2(hw), y1 - splat2(hh), splat2(f.focal));
            const v3p nd = neg3p(mat3_mul_vecp(f.rot, d));

            // ---- primary rays: closest hit of THIS sub-ray among the tiles' candidates ----
            // A candidate whose `near` bound (origin row r1.w: no hit point on it is closer to the camera) lies beyond the
            // sub-ray's current record cannot replace it, and the record already makes ClosestIntersection return true: the
            // lane does not queue it; once it lies beyond the record of EVERY pixel of its tile the wave does not even run
            // the filter on it.  The lists come roughly front to back (depth shells in the sort key), so after the first
            // drains most of a list is skipped.  The records are only updated by drains, i.e. the bounds lag -- never the result.
            s.best[lane] = MIN_T_NONE; s.best[lane + 64] = MIN_T_NONE;
            s.flag[lane] = 0; s.flag[lane + 64] = 0;
            if (STATS) ncand += nokA * nA + nokB * nB;
            // distance of the sub-ray's record so far; -inf for a lane without a pixel (outside the frame or the rows of this call):
            // every candidate's `near` lies beyond it, so such a lane never passes the near test below and needs no mask of its own
            float lbA = okA ? FLT_MAX : -__builtin_huge_valf(), lbB = okB ? FLT_MAX : -__builtin_huge_valf();
            float tbA = FLT_MAX, tbB = FLT_MAX;            // their maxima over the tile's pixels (wave-uniform)
            for (uint32_t base = 0; base < nmax; base += TR_STAGE) {
                const int cntA = (int)min((uint32_t)TR_STAGE, nA > base ? nA - base : 0u);
                const int cntB = (int)min((uint32_t)TR_STAGE, nB > base ? nB - base : 0u);
                wave_lds_fence();                          // the previous chunk's row reads are done
                float my_near = 0.0f;
                const int sh = lane >> 4, sj = lane & 15;  // lanes 0..15 stage tile A's candidates, 16..31 tile B's
                const bool stage = lane < 32 && sj < (sh ? cntB : cntA);
                if (stage) {
                    const uint32_t idx = brute ? tf.sel[base + (uint32_t)sj] : tf.cam_entries[(sh ? begB : begA) + base + (uint32_t)sj];
                    const float4 *src = reinterpret_cast<const float4 *>(f.cam_tab + idx);
                    const float4 a0 = src[0], a1 = src[1], a2 = src[2];
                    // the candidate's geometry row beside it -- what the exact stage needs for the hit point of an ACCEPTED pair, under
                    // one in a hundred of the candidates -- so that the drain makes no round trip to memory: right while the tables
                    // sit in the caches (100 k triangles: 9.6 MB), half of all the kernel fetches once they do not (1 M triangles at
                    // 8K: 96 MB of rows, 1.7 GB fetched per launch, L2 hit rate 0.44): there the drain fetches the few rows it wants
                    if (!lazy) {
                        const float4 *gsrc = geo4 + (size_t)idx * 3;
                        float4 *gdst = s.geo + lane * 3;
                        gdst[0] = gsrc[0]; gdst[1] = gsrc[1]; gdst[2] = gsrc[2];
                    }
                    float *dst = reinterpret_cast<float *>(s.rows) + sj * 24 + sh;
                    dst[0] = a0.x; dst[2] = a0.y; dst[4] = a0.z; dst[6] = a0.w;
                    dst[8] = a1.x; dst[10] = a1.y; dst[12] = a1.z; dst[14] = a1.w;
                    dst[16] = a2.x; dst[18] = a2.y; dst[20] = a2.z; dst[22] = __uint_as_float(idx);
                    my_near = a1.w;
                } else if (lane < 32) {
                    // a slot its list does not fill: `near` = +inf, so no pixel's record lets it through (the slot's twin in the
                    // other list may be real, and the step that tests it evaluates this one's stale row beside it)
                    reinterpret_cast<float *>(s.rows)[sj * 24 + sh + 14] = __builtin_huge_valf();
                }
                wave_lds_fence();
                TM_SEG(1)
                uint32_t pmA = (uint32_t)wballot(stage && sh == 0 && !(my_near > tbA)) & 0xFFFFu;
                uint32_t pmB = (uint32_t)(wballot(stage && sh == 1 && !(my_near > tbB)) >> 16) & 0xFFFFu;
                uint32_t pm = pmA | pmB;
                while (pm) {
                    const int j = __builtin_ctz(pm);
                    pm &= pm - 1u;
                    if (STATS) nsteps_p++;
                    const float4 *R = s.rows + 6 * j;
                    const float4 R0 = R[0], R1 = R[1], R2 = R[2], R3 = R[3], R4 = R[4], R5 = R[5];
                    // e1e2d, be2d, e1bd (raytracer.cpp:232-234) of (pixel A, candidate j of list A) and (pixel B, candidate j of list B)
                    TestDots2 td;
                    td.den = (f2){ R0.x, R0.y } * nd.x + (f2){ R0.z, R0.w } * nd.y + (f2){ R1.x, R1.y } * nd.z;
                    td.pu = (f2){ R2.x, R2.y } * nd.x + (f2){ R2.z, R2.w } * nd.y + (f2){ R3.x, R3.y } * nd.z;
                    td.qv = (f2){ R4.x, R4.y } * nd.x + (f2){ R4.z, R4.w } * nd.y + (f2){ R5.x, R5.y } * nd.z;
                    // the filter's verdicts and the `near` tests as wave masks straight from the compares, combined on the scalar unit
                    const f2 fm = maybe_hit2_margin(td);
                    const unsigned long long nearA = wballot(!(R3.z > lbA)), nearB = wballot(!(R3.w > lbB));
                    const unsigned long long filtA = wballot(fm.x >= MAYBE_HIT_THRESHOLD), filtB = wballot(fm.y >= MAYBE_HIT_THRESHOLD);
                    // (a candidate the wave has pruned -- near beyond the record of EVERY pixel of its tile -- fails the per-pixel
                    // test in every lane, and an unfilled slot carries near = +inf: no need to consult pmA / pmB here)
                    // (... nor the masks of the lanes that have a pixel: the others carry a record of -inf)
                    const unsigned long long mA = nearA & filtA, mB = nearB & filtB;
                    const bool passA0 = (mA >> lane) & 1ull, passB = (mB >> lane) & 1ull;
                    if (STATS) ntests += (unsigned)__popcll(nearA) + (unsigned)__popcll(nearB);
                    if (mA | mB) {
                        int cA = __popcll(mA);
                        const int cB = __popcll(mB);
                        bool passA = passA0;
                        if (qn + cA + cB > TR_QUEUE) {                             // (rare: no room for this step)
                            TM_SEG(2)
                            if (qn) { tr_drain<false>(s, lane, qn, geo4, cam, nullptr, lazy); qn = 0; if (STATS) ndrains++; }
                            if (cA + cB > TR_QUEUE) {                              // the step alone does not fit: tile A's pairs go first
                                if (passA) {
                                    const int at = wave_rank(mA);
                                    s.q[at] = make_float4(td.den.x, td.pu.x, td.qv.x, R1.z);
                                    s.qa[at] = make_uint2((uint32_t)lane | ((uint32_t)j << 8), __float_as_uint(R5.z));
                                }
                                tr_drain<false>(s, lane, cA, geo4, cam, nullptr, lazy); if (STATS) ndrains++;
                                passA = false; cA = 0;
                            }
                            TM_SEG(3)
                        }
                        if (passA) {
                            const int at = qn + wave_rank(mA);
                            s.q[at] = make_float4(td.den.x, td.pu.x, td.qv.x, R1.z);
                            s.qa[at] = make_uint2((uint32_t)lane | ((uint32_t)j << 8), __float_as_uint(R5.z));
                        }
                        if (passB) {
                            const int at = qn + cA + wave_rank(mB);
                            s.q[at] = make_float4(td.den.y, td.pu.y, td.qv.y, R1.w);
                            s.qa[at] = make_uint2(((uint32_t)lane + 64u) | ((uint32_t)(16 + j) << 8), __float_as_uint(R5.w));
                        }
                        qn += cA + cB;
                        if (qn >= TR_DRAIN) {
                            TM_SEG(2)
                            do { tr_drain_full<false>(s, lane, qn, geo4, cam, nullptr, lazy); if (STATS) ndrains++; } while (qn >= TR_DRAIN);
                            TM_SEG(3)
                            lbA = okA ? min_t_dist(s.best[lane]) : lbA; lbB = okB ? min_t_dist(s.best[lane + 64]) : lbB;
                            tbA = wave_max_f(lbA); tbB = wave_max_f(lbB);
                            pmA &= (uint32_t)wballot(stage && sh == 0 && !(my_near > tbA));
                            pmB &= (uint32_t)(wballot(stage && sh == 1 && !(my_near > tbB)) >> 16);
                            pm &= pmA | pmB;
                        }
                    }
                }
                // the chunk's queued pairs are settled before the next chunk replaces the staged rows (their geometry lives there);
                // the records they leave prune the next chunk
                TM_SEG(2)
                if (qn) {
                    tr_drain<false>(s, lane, qn, geo4, cam, nullptr, lazy); qn = 0; if (STATS) ndrains++;
                    if (base + TR_STAGE < nmax) {
                        lbA = okA ? min_t_dist(s.best[lane]) : lbA; lbB = okB ? min_t_dist(s.best[lane + 64]) : lbB;
                        tbA = wave_max_f(lbA); tbB = wave_max_f(lbB);
                    }
                }
                TM_SEG(3)
            }
            // ... merged into the pixel's running record exactly as the sequential `>=` sweep would (:243): the
            // sub-ray's best replaces the record when it is at least as close (a later sub-ray wins exact ties)
            const unsigned long long keyA = s.best[lane], keyB = s.best[lane + 64];
            const bool anyA = s.flag[lane] != 0, anyB = s.flag[lane + 64] != 0;     // ClosestIntersection's return value
            if (anyA && bdA >= min_t_dist(keyA)) {
                bdA = min_t_dist(keyA); biA = min_t_index(keyA);
                posA = V3(s.px[lane], s.py[lane], s.pz[lane]);       // the hit point as the exact stage computed it (:241)
            }
            if (anyB && bdB >= min_t_dist(keyB)) {
                bdB = min_t_dist(keyB); biB = min_t_index(keyB);
                posB = V3(s.px[lane + 64], s.py[lane + 64], s.pz[lane + 64]);
            }

            TM_SEG(4)
            const bool hitA = okA && anyA, hitB = okB && anyB;
            const unsigned long long hmA = wballot(hitA), hmB = wballot(hitB);
            const unsigned nhits = (unsigned)(__popcll(hmA) + __popcll(hmB));
            if (hmA | hmB) {
                const ShadeRow *srA = tf.shade + (biA >= 0 ? biA : 0), *srB = tf.shade + (biB >= 0 ? biB : 0);
                const float4 nA4 = srA->n, nB4 = srB->n;
                // (the four-wave instantiations fetch the triangles' colours here, beside the normals, and hold them across the shadow
                // walk: the six registers that put the kernel at 98 and so at four waves per SIMD, which is what the small scenes' frames
                // in flight want -- see above the kernel)
                float4 cA4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), cB4 = cA4;
                if constexpr (WAVES != 5) { cA4 = srA->col; cB4 = srB->col; }
                const v3p pos = join3(posA, posB);
                const v3p nDir = join3(V3(nA4.x, nA4.y, nA4.z), V3(nB4.x, nB4.y, nB4.z));   // glm::normalize(normal) (:300), per triangle
                v3p result = splat3(V3(0.0f, 0.0f, 0.0f)), result2 = result;
                for (int k = 0; k < f.nlights; k++) {
                    // DirectLight's term before the shadow test (raytracer.cpp:294-304), both pixels at once
                    // r = distance(pos, lightPos), A = 4 pi r^2, rDir = normalize(lightPos - pos), B = P / A with P = lightColor / samples (:296, divided on the host)
                    const v3 L = ld3(f.lpos[k]);
                    const LightGeometry2 lg = light_geometry2(pos, L, ld3(f.lcol[k]), f.lights_in_range != 0, hitA, hitB);
                    const f2 r = lg.r;
                    const v3p rd = lg.rDir, B = lg.B;
                    const f2 dn = dot3p(rd, nDir);
                    const f2 mx = { (dn.x < 0.0f) ? 0.0f : dn.x, (dn.y < 0.0f) ? 0.0f : dn.y };   // std::max(d, 0.0f)
                    v3p D = scale3p(B, mx);
                    const f2 thr = r * splat2(0.99f);              // (:313)
                    TM_SEG(5)
                    wave_lds_fence();
                    s.thr[lane] = thr.x; s.thr[lane + 64] = thr.y;
                    s.flag[lane] = 0; s.flag[lane + 64] = 0;
                    // the candidates of each shadow ray: the rows of its light-cube bin, shells 0 .. shell(0.99 r) -- a row of a
                    // later shell has near > 0.99 r (bin_shell_of is monotone in its argument) and cannot occlude (:313)
                    uint32_t eA = 0, endA = 0, eB = 0, endB = 0;
                    const float4 *rows4;
                    if (brute_l) {
                        rows4 = reinterpret_cast<const float4 *>(f.light_tab + (size_t)k * f.n);
                        if (hitA) endA = (uint32_t)f.n;
                        if (hitB) endB = (uint32_t)f.n;
                    } else {
                        rows4 = reinterpret_cast<const float4 *>(tf.light_rows);
                        uint32_t binA, binB;
                        cube_bin_of2(rd, (uint32_t)k * 6u * (uint32_t)(tf.cube_bins * tf.cube_bins), tf.cube_bins, &binA, &binB);
                        // (shell of 0.99 r: bin_shell_of's formula on wave-uniform parameters, loaded once per light)
                        const BinFrameDesc *lf = tf.light_frames + 6 * k;
                        const int ns = tf.light_shells;
                        const float sd0 = lf->shell_d0, siw = lf->shell_iw;
                        const uint32_t shA = ns > 1 ? (uint32_t)min(max((int)((thr.x - sd0) * siw), 0), ns - 1) : 0u;
                        const uint32_t shB = ns > 1 ? (uint32_t)min(max((int)((thr.y - sd0) * siw), 0), ns - 1) : 0u;
                        if (hitA) {
                            const uint32_t key = binA * (uint32_t)ns;
                            eA = tf.light_off[key]; endA = tf.light_off[key + shA + 1u];
                        }
                        if (hitB) {
                            const uint32_t key = binB * (uint32_t)ns;
                            eB = tf.light_off[key]; endB = tf.light_off[key + shB + 1u];
                        }
                    }
                    if (STATS) ncand_l += (endA - eA) + (endB - eB);
                    // The lane walks list A, then list B.  Per step: the filter; a candidate it lets through is either a CERTAIN
                    // occluder (sure_hit, and the whole triangle closer to the light than 0.99 r: nothing left to compute, the ray
                    // is done) or goes to the exact stage's queue.
                    // The walk's state lives in plain integer and float registers and every step is straight-line code: what a lane
                    // is doing follows from comparisons made afresh each step (active: e < end; list B still to come: on list A and
                    // eB < endB), the filter and the certain-occluder test are evaluated for every lane, and the next row is loaded by
                    // every lane (row 0 where a lane has none to fetch).  Written with booleans carried around the loop and the
                    // tests nested in ifs, the compiler kept each boolean as a scalar lane mask and merged it with exec at every
                    // divergent join: ~60 scalar instructions per step beside ~30 vector ones -- and at the four waves per SIMD this
                    // kernel runs at, a scalar instruction costs what a vector one costs (profiles/r03_issue_model.txt).
                    const uint32_t *row_tri = brute_l ? nullptr : tf.light_tri;
                    const bool firstA = eA < endA;
                    uint32_t e = firstA ? eA : eB, end = firstA ? endA : endB, pix = firstA ? (uint32_t)lane : (uint32_t)lane + 64u;
                    v3 crd = firstA ? half0(rd) : half1(rd);
                    float cthr = firstA ? thr.x : thr.y;
                    uint32_t occ = 0u;                               // bit 0 / 1: the lane itself found a certain occluder of pixel A / B
                    float4 c0, c1, c2;
                    { const float4 *src = rows4 + (size_t)(e < end ? e : 0u) * 3; c0 = src[0]; c1 = src[1]; c2 = src[2]; }
                    TM_SEG(6)
                    const int not_brute = brute_l ? 0 : 1;
                    for (;;) {
                        const bool act = e < end;
                        const unsigned long long am = wballot(act);
                        if (!am) break;
                        if (STATS) { nsteps_s++; ntests += (unsigned)__popcll(am); }
                        const TestDots td = test_dots(c0, c1, c2, crd);          // negD = rDir (:310, :229)
                        // a candidate none of whose points is closer to the light than 0.99 r cannot occlude (:313)
                        const int pass = (int)act & (int)!(c1.w > cthr) & (int)maybe_hit(td);
                        const int sure = pass & not_brute & (int)(c2.w < cthr) & (int)sure_hit(td, c0.w);
                        const int queue = pass & (sure ^ 1);
                        const unsigned long long m = wballot(queue != 0);
                        int done = sure;                                          // this list is settled for the lane
                        if (m) {
                            if (qn + __popcll(m) > TR_QUEUE) { tr_drain<true>(s, lane, qn, geo4, L, row_tri); qn = 0; if (STATS) ndrains++; }   // (rare: no room for this step)
                            if (queue) {
                                const int at = qn + wave_rank(m);
                                s.q[at] = make_float4(td.den, td.pu, td.qv, c0.w);
                                s.qa[at] = make_uint2(pix, e);                   // the candidate's row
                            }
                            qn += __popcll(m);
                            if (qn >= TR_DRAIN) {
                                TM_SEG(7)
                                do { tr_drain_full<true>(s, lane, qn, geo4, L, row_tri); if (STATS) ndrains++; } while (qn >= TR_DRAIN);
                                TM_SEG(8)
                                done |= (int)act & (int)(s.flag[pix] != 0);     // found occluded: the rest of this list is moot
                            }
                        }
                        occ |= sure ? (pix < 64u ? 1u : 2u) : 0u;
                        // on: the next row of this list, or -- list done or ray occluded -- the first of list B
                        const int cont = (int)act & (done ^ 1) & (int)(e + 1u < end);
                        const int sw = (int)act & (cont ^ 1) & (int)(pix < 64u) & (int)(eB < endB);
                        crd = sw ? half1(rd) : crd;
                        cthr = sw ? thr.y : cthr;
                        pix = sw ? (uint32_t)lane + 64u : pix;
                        e = cont ? e + 1u : (sw ? eB : end);                     // (neither: e == end, the lane is done)
                        end = sw ? endB : end;
                        const float4 *src = rows4 + (size_t)((cont | sw) ? e : 0u) * 3;
                        c0 = src[0]; c1 = src[1]; c2 = src[2];
                    }
                    TM_SEG(7)
                    if (qn) { tr_drain<true>(s, lane, qn, geo4, L, row_tri); qn = 0; if (STATS) ndrains++; }
                    TM_SEG(8)
                    const bool occA = (occ & 1u) != 0u, occB = (occ & 2u) != 0u;
                    // occluded (:313-314); any-hit is exact
                    if (occA || s.flag[lane] != 0) { D.x.x = 0.0f; D.y.x = 0.0f; D.z.x = 0.0f; }
                    if (occB || s.flag[lane + 64] != 0) { D.x.y = 0.0f; D.y.y = 0.0f; D.z.y = 0.0f; }
                    result = add3p(result, D);                     // (:319)
                    if ((k + 1) % f.samples == 0) result2 = add3p(result2, result);   // (:322) after each light's samples
                }
                // (the triangles' colours only now: fetched beside the normals they were six registers held across the whole shadow walk,
                // the two that kept the kernel from a fifth wave per SIMD)
                if constexpr (WAVES == 5) { cA4 = srA->col; cB4 = srB->col; }
                const v3p tcol = join3(V3(cA4.x, cA4.y, cA4.z), V3(cB4.x, cB4.y, cB4.z));
                const v3p Dl = mul3p(result2, tcol);               // (:325-326)
                const v3p shaded = add3p(avg, mul3p(tcol, add3p(Dl, splat3(ld3(f.indirect)))));   // (:584-591)
                avg = join3(hitA ? half0(shaded) : half0(avg), hitB ? half1(shaded) : half1(avg));
                if (AA) x1 = x1 + (f2){ hitA ? aa_step(rs) : 0.0f, hitB ? aa_step(rs) : 0.0f };   // (:593) only after a hit
            }
            // the frame's counters: hits (= shadow rays per light position), then the kernel's own statistics at the end
            if (lane == 0 && nhits) {
                const unsigned shard = (blockIdx.x + (threadIdx.x >> 6) * 61u) % HIT_SHARDS;
                atomicAdd(f.hit_count + (size_t)shard * HIT_SHARD_STRIDE, (unsigned long long)nhits);
            }
            wave_lds_fence();
        }
        if (AA) y1 = y1 + splat2(aa_step(rs));                     // (:596)
    }
    {
        // tests executed, candidates offered (list entries x rays), wave steps of the two filter loops, drains: one atomic
        // instruction, lane i adding to word 1 + i of the wave's shard
        if (STATS) ncand += wave_sum(ncand_l);
        const unsigned shard = (blockIdx.x + (threadIdx.x >> 6) * 61u) % HIT_SHARDS;
#ifdef MIRT_TR_TIMING
        // (experiments) words 3..12 carry the segments' sums instead of the step counts, word 13 the longest lifetime
        TM_SEG(9)
        if (lane == 0) {
            for (int i = 0; i < 10; i++) atomicAdd(f.hit_count + (size_t)shard * HIT_SHARD_STRIDE + 3 + i, (unsigned long long)tm_seg[i]);
            atomicMax(f.hit_count + (size_t)shard * HIT_SHARD_STRIDE + 13, (unsigned long long)(tm_last - tm0));
        }
        const unsigned v = lane == 0 ? ntests : ncand;
        if (lane < 2 && v) atomicAdd(f.hit_count + (size_t)shard * HIT_SHARD_STRIDE + 1 + lane, (unsigned long long)v);
#else
        const unsigned v = lane == 0 ? ntests : lane == 1 ? ncand : lane == 2 ? nsteps_p : lane == 3 ? nsteps_s : ndrains;
        if (STATS && lane < 5 && v) atomicAdd(f.hit_count + (size_t)shard * HIT_SHARD_STRIDE + 1 + lane, (unsigned long long)v);
#endif
    }
    if (AA) {                                                      // avgColor /= realSamples^2 (:599); /1 is the identity
        const f2 q = splat2((float)(rs * rs));
        avg = V3P(div2(avg.x, q), div2(avg.y, q), div2(avg.z, q));
    }
    if (okA) {
        const v3 c = half0(avg);
        const size_t px = (size_t)y * f.W + xA;
        if (f.rgb) st3(f.rgb + 3 * px, c);
        if (f.index) f.index[px] = biA;
        if (f.fd) f.fd[px] = biA >= 0 ? bdA - f.focal_plane : 0.0f;            // focalDistances (:248-249)
        store_intersection(f, px, biA, bdA, posA);
        if (xA >= 1 && xA < f.W - 1 && y >= 1 && y < f.H - 1)      // (:618-620)
            f.xrgb[(size_t)(y - f.row_origin) * f.pitch_words + xA] = pack_xrgb(c);
    }
    if (okB) {
        const v3 c = half1(avg);
        const size_t px = (size_t)y * f.W + xB;
        if (f.rgb) st3(f.rgb + 3 * px, c);
        if (f.index) f.index[px] = biB;
        if (f.fd) f.fd[px] = biB >= 0 ? bdB - f.focal_plane : 0.0f;
        store_intersection(f, px, biB, bdB, posB);
        if (xB >= 1 && xB < f.W - 1 && y >= 1 && y < f.H - 1)
            f.xrgb[(size_t)(y - f.row_origin) * f.pitch_words + xB] = pack_xrgb(c);
    }
}

template __global__ void k_rt_trace2<false, false>(const RtTraceFrame);         // the frames of the loop: four waves per SIMD ...
template __global__ void k_rt_trace2<false, false, 5>(const RtTraceFrame);      // ... or five (mirt_capi.hip)
template __global__ void k_rt_trace2<false, true>(const RtTraceFrame);
template __global__ void k_rt_trace2<true, false>(const RtTraceFrame);
template __global__ void k_rt_trace2<true, true>(const RtTraceFrame);

size_t rt_trace_lds_bytes(int waves) { return (size_t)waves * sizeof(TrWaveLds); }

}  // namespace mirt
