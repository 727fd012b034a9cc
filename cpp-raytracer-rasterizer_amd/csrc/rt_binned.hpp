// rt_binned.hpp -- result-preserving candidate reduction for the ray tracer.
//
// Brute force tests every ray against every triangle (what ClosestIntersection does); at 100k triangles that
// is ~10^11 tests per frame.  The accept test of the reference,
//     u + v <= 1 && u >= 0 && v >= 0 && t >= 0,  u = be2d/e1e2d, v = e1bd/e1e2d, t = e1e2b/e1e2d,
// only looks at the signs of three functions that are LINEAR in the ray direction negD:
//     e1e2d = e1e2 . negD,   be2d = be2 . negD,   e1bd = e1b . negD         (raytracer.cpp:232-234)
// so for a family of rays negD ~ P0 + u*Pu + v*Pv (all rays of one origin through a 2-D parameter grid) each
// of them is an affine "edge function" of (u, v), and a rectangle of the grid can contain an accepted ray only
// if the corner bounds of those functions allow it.  That is a conservative tile test with no projection and
// no clipping, valid for triangles beside or behind the origin too.  Two families are used:
//   * the camera:  negD = -(R0*(x - W/2) + R1*(y - H/2) + R2*f), (u, v) = pixel (x, y); bins = 8x8-pixel tiles;
//   * each light:  six cube faces, negD = rDir ~ s*e_k + u*e_k1 + v*e_k2 with (u, v) in [-1,1]^2; bins = a
//     regular B x B grid per face.
// Every ray then tests only the triangles of its bin, with the SAME filter + exact arithmetic as brute force
// (rt_common.hpp), so accepted hits and their distances are bit-identical; the closest-hit tie rule (">=" in
// index order, raytracer.cpp:243) is restated order-independently as (min distance, then max index).
//
// Margins: a rectangle keeps a triangle unless an edge function is provably below -m (resp. above +m) on all
// of it, with m = 2^-17 * (|g.x|+|g.y|+|g.z|) * dmax + 2^-20.  dmax bounds |negD| components over the family,
// so 2^-17*M is >100x the worst float rounding of the reference's own dot product (<= 4 ulp of M) and of the
// bound evaluation; the 2^-20 covers the filter's absolute threshold (2^-22, scaled by <= sqrt(3) on cube faces).
#pragma once

#include "rt_common.hpp"

namespace mirt {

#ifndef MIRT_ORDER_STRIPE_SHIFT
#define MIRT_ORDER_STRIPE_SHIFT 2
#endif
constexpr int ORDER_STRIPE_SHIFT = MIRT_ORDER_STRIPE_SHIFT;   // log2 of the tile rows per stripe dealt to one XCD group (rt_trace.hip: k_tile_order)
constexpr int BIN_TILE = 8;            // camera bins are 8x8 pixels = one wave64
constexpr int BIN_COARSE = 8;          // coarse cell = 8x8 fine bins (the three-level walk of huge items, rt_binned.hip)
// (k_bin_pairs<WG>: workgroups of 512 threads -- a work item's 256 triangles are set up by the first four waves, the flattened bin tests
// and the flushes run on all eight -- or of 256, every wave doing both: rt_binned.hip says which when)
constexpr int CUBE_BINS_MIN = 64;      // per-face light-cube grid is B x B; B = 64 by default (128 / 256 selectable)
constexpr int MAX_BIN_FRAMES = 1 + 6 * MIRT_MAX_LIGHTS;

struct BinFrameDesc {
    float P0[3], Pu[3], Pv[3];   // negD ~ P0 + u*Pu + v*Pv
    float S[3];                  // the family's ray origin (camera or light position)
    float ru[3], rv[3], rw[3];   // inverse map: for g = S - P,  (u, v) = (ru.g, rv.g) / (rw.g), valid while rw.g > 0
    float ulo, vlo, du, dv;      // bin (i,j) covers u in [ulo + i*du + pad_lo, ulo + (i+1)*du + pad_hi]
    float pad_lo, pad_hi;
    float dmax;                  // bound on |negD| components over the family
    int nbu, nbv;                // fine bins along u and v
    int j0, j1;                  // rows of bins to build: [j0, j1)
    uint32_t base;               // index of this frame's bin (0,0) in the global bin arrays
    int tab;                     // origin table: 0 = camera, 1 + k = light k
    // Depth shells: a bin's list is ordered front to back in `nshell` coarse steps of the candidates' `near` bound (origin
    // row r1.w) -- sort key = bin * nshell + shell.  Purely an ordering: consumers skip candidates by their own `near`, so no
    // result depends on it (nshell = 1: unordered).
    int nshell;
    float shell_d0, shell_iw;    // shell = clamp((int)((near - shell_d0) * shell_iw), 0, nshell - 1)
};

__device__ __forceinline__ uint32_t bin_shell_of(const BinFrameDesc &fr, float near)
{
    if (fr.nshell <= 1) return 0u;
    const float s = (near - fr.shell_d0) * fr.shell_iw;
    return (uint32_t)min(max((int)s, 0), fr.nshell - 1);         // NaN converts to 0
}

// Geometry row of a triangle: {v0.xyz, e1.x | e1.yz, e2.xy | e2.z, 0, 0, 0} -- what the accept path needs to rebuild the hit
// point (raytracer.cpp:216-217, :241).  One per triangle (k_geo_table, rt_trace.hip).
struct GeoRow { float4 g0, g1, g2; };
// What shading needs of the closest triangle, as two 16-byte loads: glm::normalize(triangles[i].normal) (raytracer.cpp:300 -- per
// triangle the same operands and operations as per pixel, so the same bits) and its colour.  One per triangle (k_geo_table).
struct ShadeRow { float4 n, col; };
// A shadow-ray candidate expanded in light-cube bin order: the origin row of the triangle for that light, with the triangle's
// index in the one slot an origin row leaves free (r2.w, as bits) -- what the exact stage needs to find the geometry row.
// (Round 2 first stored {origin row, geometry row}, 96 bytes: 250 MB of reads per 1080p frame of the 100 k soup, half of them
// geometry that only the few accepted pairs ever look at.)
typedef OriginRow LightRow;

struct BinSet {
    const BinFrameDesc *frames;   // device array of nframes descriptors, or NULL: the single frame `frame0` below
    BinFrameDesc frame0;          // the camera frame travels as a kernel argument (no copy per frame)
    int nframes;
    uint32_t nbins;               // total bins over all frames
    uint32_t *bin_off;            // nbins + 1: first entry of every bin (bin_off[nbins] = total entries), k_bin_offsets
    uint32_t *entries;            // candidate triangle indices ordered by bin (the sorted pair values)
    uint32_t cap_entries;
    uint32_t *counters;           // [0] pairs produced by k_bin_pairs (may exceed the capacity: then the frame is redone)
    uint32_t *bucket_cnt;         // nullable: pairs per bucket = bin >> bucket_shift, for the bucket sort (bin_bucket_sort.hip)
    uint32_t nbuckets;
    int bucket_shift;
    int chunk_tris;               // triangles per work item of k_bin_pairs: 256 or 64
    // the camera frame's triangles, when k_prep_select has chosen them (nullable: every triangle): the work items of the frame
    // whose origin table is the camera's (tab == 0) walk this list instead of the scene
    const uint32_t *sel;
    const uint32_t *sel_count;
    // ... and the light-cube faces' (k_select_faces): the i-th frame that is not the listed camera frame walks face list i --
    // face_counts[i] triangle indices at face_lists + i * face_stride (nullable: every triangle for every face)
    const uint32_t *face_lists;
    const uint32_t *face_counts;
    uint32_t face_stride;
};

// What k_prep_select leaves behind for the frame's other kernels, per stream.
constexpr int SEL_HIST_MAX = 256;      // coarse tile rows of the cost histogram (a frame's tile rows >> hist_shift)
struct SelectOut {
    OriginRow *cam_tab;           // rows of the SELECTED triangles only (the others keep whatever an earlier frame left)
    uint32_t *sel;                // their indices (n slots)
    uint32_t *sel_count;          // how many; zero on entry
    uint32_t *sel_count_next;     // the counter the NEXT pass will use: zeroed here
    uint32_t *hist;               // nullable: SEL_HIST_MAX words, += estimated (tile, triangle) pairs per coarse tile row of the WHOLE frame
    int hist_shift;
    uint32_t *zero_faces;            // nullable: the face lists' counters of this frame's light cubes (k_select_faces), zero_faces_n words
    int zero_faces_n;
    unsigned long long *zero_hits;   // nullable: the frame's hit counters (k_prep_origin's duty as first kernel of a frame)
    uint32_t *zero_counter;          // nullable: the binning pass's counters: [0] and [16..79] are zeroed
};

// The unsorted (bin, triangle) pair list k_bin_pairs writes and bin_sort.hip orders by bin.
struct BinPairs {
    uint32_t *keys;               // bin ids
    uint32_t *vals;               // triangle indices
    uint32_t cap;
};

// affine edge function over (u,v) with its safety margin
struct EdgeFn { float c0, cu, cv, m; };

__device__ __forceinline__ EdgeFn make_edge_fn(float gx, float gy, float gz, const BinFrameDesc &fr)
{
    EdgeFn e;
    e.c0 = gx * fr.P0[0] + gy * fr.P0[1] + gz * fr.P0[2];
    e.cu = gx * fr.Pu[0] + gy * fr.Pu[1] + gz * fr.Pu[2];
    e.cv = gx * fr.Pv[0] + gy * fr.Pv[1] + gz * fr.Pv[2];
    e.m = 7.62939453125e-06f * ((fabsf(gx) + fabsf(gy) + fabsf(gz)) * fr.dmax) + 9.5367431640625e-07f;
    return e;
}

struct TriBinFns {
    EdgeFn n, p, q, s;
    float nb;
    float bu0, bu1, bv0, bv1;    // conservative (u,v) bounding box of the accept region (state BOX_VALID)
    int bstate;                  // BOX_NONE: rely on the edge functions; BOX_VALID; BOX_EMPTY: no ray of the family can hit
};
enum { BOX_NONE = 0, BOX_VALID = 1, BOX_EMPTY = 2 };

__device__ __forceinline__ TriBinFns make_bin_fns(const OriginRow &r, const BinFrameDesc &fr)
{
    TriBinFns t;
    t.n = make_edge_fn(r.r0.x, r.r0.y, r.r0.z, fr);
    t.p = make_edge_fn(r.r1.x, r.r1.y, r.r1.z, fr);
    t.q = make_edge_fn(r.r2.x, r.r2.y, r.r2.z, fr);
    // slack function e1e2d - be2d - e1bd, bounded term by term (keeps the margin honest under cancellation)
    t.s.c0 = t.n.c0 - t.p.c0 - t.q.c0;
    t.s.cu = t.n.cu - t.p.cu - t.q.cu;
    t.s.cv = t.n.cv - t.p.cv - t.q.cv;
    // s = n - p - q term by term: the three margins add up; the filter's own slack on top is D*2^-20 <= 2^-3 * m_n (m_n is
    // 2^-17 * |e1e2|_1 * dmax >= 2^-17 * D), so a quarter more covers it
    t.s.m = 1.25f * (t.n.m + t.p.m + t.q.m);
    t.nb = r.r0.w;
    t.bu0 = t.bu1 = t.bv0 = t.bv1 = 0.0f;
    t.bstate = BOX_NONE;
    return t;
}

// Bounding box of the triangle's accept region in the frame's (u,v) parameters.
//
// The edge-function test alone lets through rectangles that lie outside the triangle but are crossed by the
// extensions of its edges (the classic false positives of half-plane-only rasterisation), and it cannot see that a
// triangle lies BEHIND the family's projection plane (every edge line still crosses a big rectangle).  Both are
// settled from the three vertices, g_j = S - v_j, projected with the frame's inverse map (u,v) = (ru.g, rv.g)/(rw.g):
//
//  * every accepted ray has a >= -m_p, b >= -m_q, s >= -m_s (the margins the edge functions already use).  With signed
//    distances to the three projected edge lines the identity  sum(edge_len * dist) = -2*area  holds everywhere, so:
//  * all three vertices clearly in front (rw.g > 0): the accept region lies inside the projected triangle with each
//    edge line pushed out by d = largest margin distance -> box of that pushed-out triangle, padded by the rounding
//    of the projection (BOX_VALID);
//  * all three clearly behind (rw.g < 0): the projected lines bound the ANTIPODAL triangle, inside which the three
//    functions have the rejecting sign; a point within margin of all three accepting sides would need d >= r_in.  If
//    d < r_in/2 (r_in bounded from below) no ray of this family can be accepted at all (BOX_EMPTY);
//  * anything else (a vertex near the plane, degenerate projections, NaN) keeps BOX_NONE and relies on the edge
//    functions, which are always valid.
__device__ __forceinline__ void add_bbox(TriBinFns &t, const float *t15, const BinFrameDesc &fr)
{
    float us[3], vs[3], pad = 0.0f;
    int front = 0, behind = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const float gx = fr.S[0] - t15[3 * j], gy = fr.S[1] - t15[3 * j + 1], gz = fr.S[2] - t15[3 * j + 2];
        const float w = fr.rw[0] * gx + fr.rw[1] * gy + fr.rw[2] * gz;
        const float wm = fabsf(fr.rw[0] * gx) + fabsf(fr.rw[1] * gy) + fabsf(fr.rw[2] * gz);
        front += (w > 0.00390625f * wm);                           // clearly in front / behind: |w| > 2^-8 of its terms
        behind += (w < -0.00390625f * wm);
        const float un = fr.ru[0] * gx + fr.ru[1] * gy + fr.ru[2] * gz, vn = fr.rv[0] * gx + fr.rv[1] * gy + fr.rv[2] * gz;
        const float um = fabsf(fr.ru[0] * gx) + fabsf(fr.ru[1] * gy) + fabsf(fr.ru[2] * gz);
        const float vm = fabsf(fr.rv[0] * gx) + fabsf(fr.rv[1] * gy) + fabsf(fr.rv[2] * gz);
        us[j] = un / w; vs[j] = vn / w;
        // rounding of the projection: un, vn, w each carry <= 2^-22 of their term sums (3 products, 2 sums), the
        // quotient 2^-24 more:  |du| <= (2^-22*um + |u|*2^-22*wm)/|w| + 2^-24*|u|; doubled for comfort
        const float iw = 1.0f / fabsf(w);
        pad = fmaxf(pad, 4.76837158203125e-07f * ((um + fabsf(us[j]) * wm) * iw + (vm + fabsf(vs[j]) * wm) * iw) +
                             2.384185791015625e-07f * (fabsf(us[j]) + fabsf(vs[j])));
    }
    if (front != 3 && behind != 3) return;
    const float u0 = fminf(fminf(us[0], us[1]), us[2]), u1 = fmaxf(fmaxf(us[0], us[1]), us[2]);
    const float v0 = fminf(fminf(vs[0], vs[1]), vs[2]), v1 = fmaxf(fmaxf(vs[0], vs[1]), vs[2]);
    const float ext = fmaxf(u1 - u0, v1 - v0);
    // inradius = 2*area / perimeter of the projected triangle
    const float ax = us[1] - us[0], ay = vs[1] - vs[0], bx = us[2] - us[0], by = vs[2] - vs[0], cx = us[2] - us[1], cy = vs[2] - vs[1];
    const float area2 = fabsf(ax * by - ay * bx);
    const float per = sqrtf(ax * ax + ay * ay) + sqrtf(bx * bx + by * by) + sqrtf(cx * cx + cy * cy);
    // margin distances of the three edge functions, in (u,v) units
    const float dp = t.p.m / sqrtf(t.p.cu * t.p.cu + t.p.cv * t.p.cv);
    const float dq = t.q.m / sqrtf(t.q.cu * t.q.cu + t.q.cv * t.q.cv);
    const float ds = t.s.m / sqrtf(t.s.cu * t.s.cu + t.s.cv * t.s.cv);
    const float d = fmaxf(fmaxf(dp, dq), ds) + pad;
    if (behind == 3) {
        // needs d < r_in = 2*area/perimeter.  r_in is taken from below: every vertex may be off by `pad` (area changes by
        // at most pad*perimeter, doubled) and the cross product cancels (2^-21 of its two products); half of that bound
        // is the threshold.  NaN or a vanishing area keep BOX_NONE.
        const float area_lo = area2 - 2.0f * pad * per - 4.76837158203125e-07f * (fabsf(ax * by) + fabsf(ay * bx));
        if (d < 0.5f * (area_lo / per)) t.bstate = BOX_EMPTY;
        return;
    }
    // front: the region {dist_i >= -d} is the triangle with every edge line pushed out by d, i.e. the triangle whose
    // vertex i sits at  V_i - d*(ua + ub)/|ua x ub|  (ua, ub = unit vectors along the two edges leaving V_i; the
    // offset is d/sin(angle/2) along the outward bisector).  Its box is tight even for needles, whose tip runs far
    // out along the needle only.  Degenerate corners give a non-finite offset and keep BOX_NONE.
    const float l01 = sqrtf(ax * ax + ay * ay), l02 = sqrtf(bx * bx + by * by), l12 = sqrtf(cx * cx + cy * cy);
    const float e01x = ax / l01, e01y = ay / l01, e02x = bx / l02, e02y = by / l02, e12x = cx / l12, e12y = cy / l12;
    const float dd = 1.25f * d;
    const float k0 = dd / fabsf(e01x * e02y - e01y * e02x), k1 = dd / fabsf(e01x * e12y - e01y * e12x), k2 = dd / fabsf(e02x * e12y - e02y * e12x);
    const float px0 = us[0] - k0 * (e01x + e02x), py0 = vs[0] - k0 * (e01y + e02y);
    const float px1 = us[1] - k1 * (e12x - e01x), py1 = vs[1] - k1 * (e12y - e01y);
    const float px2 = us[2] + k2 * (e02x + e12x), py2 = vs[2] + k2 * (e02y + e12y);
    const float slack = 2.0f * pad + 1.0e-6f * ext;
    const float bu0 = fminf(fminf(px0, px1), px2) - slack, bu1 = fmaxf(fmaxf(px0, px1), px2) + slack;
    const float bv0 = fminf(fminf(py0, py1), py2) - slack, bv1 = fmaxf(fmaxf(py0, py1), py2) + slack;
    if (!(bu0 > -1.0e30f && bu1 < 1.0e30f && bv0 > -1.0e30f && bv1 < 1.0e30f)) return;     // also NaN
    // the pushed-out triangle contains the original one; keep that explicit against rounding of the offsets
    t.bu0 = fminf(bu0, u0 - slack); t.bu1 = fmaxf(bu1, u1 + slack); t.bv0 = fminf(bv0, v0 - slack); t.bv1 = fmaxf(bv1, v1 + slack);
    t.bstate = BOX_VALID;
}

// ---- the cheap pre-test in front of all that (k_prep_select) -------------------------------------------------------------------
// A frame that renders a BAND of rows (one rank of a sharded frame: mirt_raytrace_sharded) binned every triangle of the scene
// for it: origin row, edge functions and box of add_bbox() -- about a thousand instructions -- for the seven eighths of the
// triangles that cannot reach the band.  This test costs a sixth of that and settles most of them.  It is add_bbox()'s own
// argument with a coarser box: for a triangle whose three vertices lie clearly in front of the family's projection plane, every
// accepted ray lies inside the projected triangle with its edge lines pushed out by d = the largest margin distance of the three
// edge functions; vertex i of that pushed-out triangle sits d / sin(angle_i / 2) from the projected vertex, and
//     sin(angle_i / 2) >= sin(angle_i) / 2 = area2 / (2 * l_a * l_b) >= area2 / (2 * (extu^2 + extv^2))
// bounds every such offset by  disp = 2 * dd * (extu^2 + extv^2) / area2  (dd = 1.25 * d as in add_bbox) without a square root
// or a division per corner.  The box [u0 - disp, u1 + disp] x [v0 - disp, v1 + disp] therefore CONTAINS add_bbox()'s; a frame
// whose rows and columns it misses (by a whole bin more than add_bbox()'s own conversion would ask for) holds no pair of the
// triangle.  Reciprocals and reciprocal square roots are the hardware's one-ulp approximations; the projection's pad carries
// 2^-21 per quotient instead of add_bbox()'s 2^-22 for them, margin distances and the displacement are raised by 2^-19.
// A triangle whose three vertices lie clearly BEHIND the plane -- what five of the six faces of a light's cube see of most triangles
// -- cannot be hit at all while d stays below half its projected inradius (add_bbox's BOX_EMPTY, with the inradius bounded from
// below by area_lo over the box's perimeter).
// Everything else -- a vertex near the plane, a vanishing area, an origin in the triangle's plane (|e1e2b| below the
// threshold k_bin_pairs uses for "either sign"), any NaN -- answers "may be seen" and leaves the decision to the full set-up.
// Returns false only when NO ray of rows [j0, j1) x columns [0, nbu) of the frame can be accepted on the triangle; *boxed says
// whether the box outputs (bin-index ranges over the whole grid, unclamped floats) are valid -- the cost histogram uses them.
struct PreBox { float lou, hiu, lov, hiv; };
__device__ __forceinline__ bool frame_may_see(const OriginRow &row, v3 va, v3 vb, v3 vc, const BinFrameDesc &fr, PreBox *box, bool *boxed)
{
    *boxed = false;
    const float nbv = row.r0.w;
    if (!(fabsf(nbv) >= 1.6940658945086007e-21f)) return true;            // either sign of e1e2d may pass t >= 0 (k_bin_pairs: `both`); also NaN
    const v3 vert[3] = { va, vb, vc };
    float us[3], vs[3], pad = 0.0f;
    int front = 0, behind = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const float gx = fr.S[0] - vert[j].x, gy = fr.S[1] - vert[j].y, gz = fr.S[2] - vert[j].z;
        const float w = fr.rw[0] * gx + fr.rw[1] * gy + fr.rw[2] * gz;
        const float wm = fabsf(fr.rw[0] * gx) + fabsf(fr.rw[1] * gy) + fabsf(fr.rw[2] * gz);
        front += (w > 0.00390625f * wm);
        behind += (w < -0.00390625f * wm);
        const float un = fr.ru[0] * gx + fr.ru[1] * gy + fr.ru[2] * gz, vn = fr.rv[0] * gx + fr.rv[1] * gy + fr.rv[2] * gz;
        const float um = fabsf(fr.ru[0] * gx) + fabsf(fr.ru[1] * gy) + fabsf(fr.ru[2] * gz);
        const float vm = fabsf(fr.rv[0] * gx) + fabsf(fr.rv[1] * gy) + fabsf(fr.rv[2] * gz);
        const float iw = __builtin_amdgcn_rcpf(w);
        us[j] = un * iw; vs[j] = vn * iw;
        const float aiw = fabsf(iw);
        pad = fmaxf(pad, 4.76837158203125e-07f * ((um + fabsf(us[j]) * wm) * aiw + (vm + fabsf(vs[j]) * wm) * aiw) +
                             4.76837158203125e-07f * (fabsf(us[j]) + fabsf(vs[j])));
    }
    if (front != 3 && behind != 3) return true;
    // margin distances of the three edge functions p, q, s = n - p - q (make_edge_fn / make_bin_fns: same gradients, same margins)
    float cu[3], cv[3], mg[3];
    const float4 rr[3] = { row.r0, row.r1, row.r2 };
#pragma unroll
    for (int k = 0; k < 3; k++) {
        cu[k] = rr[k].x * fr.Pu[0] + rr[k].y * fr.Pu[1] + rr[k].z * fr.Pu[2];
        cv[k] = rr[k].x * fr.Pv[0] + rr[k].y * fr.Pv[1] + rr[k].z * fr.Pv[2];
        mg[k] = 7.62939453125e-06f * ((fabsf(rr[k].x) + fabsf(rr[k].y) + fabsf(rr[k].z)) * fr.dmax) + 9.5367431640625e-07f;
    }
    const float scu = cu[0] - cu[1] - cu[2], scv = cv[0] - cv[1] - cv[2], sm = 1.25f * (mg[0] + mg[1] + mg[2]);
    const float dp = mg[1] * __builtin_amdgcn_rsqf(cu[1] * cu[1] + cv[1] * cv[1]);
    const float dq = mg[2] * __builtin_amdgcn_rsqf(cu[2] * cu[2] + cv[2] * cv[2]);
    const float ds = sm * __builtin_amdgcn_rsqf(scu * scu + scv * scv);
    const float d = fmaxf(fmaxf(dp, dq), ds) * 1.0000019073486328125f + pad;
    const float u0 = fminf(fminf(us[0], us[1]), us[2]), u1 = fmaxf(fmaxf(us[0], us[1]), us[2]);
    const float v0 = fminf(fminf(vs[0], vs[1]), vs[2]), v1 = fmaxf(fmaxf(vs[0], vs[1]), vs[2]);
    const float extu = u1 - u0, extv = v1 - v0;
    const float ax = us[1] - us[0], ay = vs[1] - vs[0], bx = us[2] - us[0], by = vs[2] - vs[0];
    const float t1 = ax * by, t2 = ay * bx;
    // the area from below: every vertex may be off by `pad` (the area changes by at most pad * perimeter, doubled; the perimeter of
    // a triangle is at most that of its box) and the cross product cancels (2^-21 of its two products)
    const float area_lo = fabsf(t1 - t2) - 4.0f * pad * (extu + extv) - 4.76837158203125e-07f * (fabsf(t1) + fabsf(t2));
    if (!(area_lo > 0.0f)) return true;                                    // (also NaN)
    if (behind == 3) {
        // all three vertices clearly BEHIND the projection plane (five of the six faces of a light's cube see most triangles so):
        // the projected lines bound the antipodal triangle, inside which the three functions have the rejecting sign; a point within
        // margin of all three accepting sides needs d >= r_in (add_bbox).  r_in from below: the area from below over the perimeter
        // from above (that of the box); half of that is the threshold, as there.  Otherwise: the full set-up decides.
        return !(d < 0.5f * (area_lo * __builtin_amdgcn_rcpf(2.0f * (extu + extv)) * 0.99999809265136719f));
    }
    const float disp = (2.5f * d) * (extu * extu + extv * extv) * __builtin_amdgcn_rcpf(area_lo) * 1.0000019073486328125f;
    const float slack = 2.0f * pad + 1.0e-6f * fmaxf(extu, extv);
    const float bu0 = u0 - disp - slack, bu1 = u1 + disp + slack, bv0 = v0 - disp - slack, bv1 = v1 + disp + slack;
    if (!(bu0 > -1.0e30f && bu1 < 1.0e30f && bv0 > -1.0e30f && bv1 < 1.0e30f)) return true;   // also NaN
    // bin-index ranges as k_bin_pairs converts add_bbox()'s box (bin i passes `i + 1 >= lou && i <= hiu`), widened likewise
    const float idu = __builtin_amdgcn_rcpf(fr.du), idv = __builtin_amdgcn_rcpf(fr.dv);
    PreBox b;
    b.lou = (bu0 - fr.ulo - fr.pad_hi) * idu; b.hiu = (bu1 - fr.ulo - fr.pad_lo) * idu;
    b.lov = (bv0 - fr.vlo - fr.pad_hi) * idv; b.hiv = (bv1 - fr.vlo - fr.pad_lo) * idv;
    b.lou -= 7.62939453125e-06f * (1.0f + fabsf(b.lou)); b.hiu += 7.62939453125e-06f * (1.0f + fabsf(b.hiu));
    b.lov -= 7.62939453125e-06f * (1.0f + fabsf(b.lov)); b.hiv += 7.62939453125e-06f * (1.0f + fabsf(b.hiv));
    *box = b;
    *boxed = true;
    // a whole bin of room on every side beyond what the conversion asks for
    const bool miss = (b.hiu < -1.0f) || (b.lou > (float)fr.nbu + 1.0f) || (b.hiv < (float)fr.j0 - 1.0f) || (b.lov > (float)fr.j1 + 1.0f);
    return !miss;
}

__device__ __forceinline__ void fn_range(const EdgeFn &e, float u0, float u1, float v0, float v1, float *lo, float *hi)
{
    const float a0 = e.cu * u0, a1 = e.cu * u1, b0 = e.cv * v0, b1 = e.cv * v1;
    *hi = e.c0 + fmaxf(a0, a1) + fmaxf(b0, b1);
    *lo = e.c0 + fminf(a0, a1) + fminf(b0, b1);
}

// May some ray with (u,v) in the rectangle be accepted by the reference's test?  (conservative)
__device__ __forceinline__ bool rect_may_hit(const TriBinFns &t, float u0, float u1, float v0, float v1)
{
    float nlo, nhi, plo, phi, qlo, qhi, slo, shi;
    fn_range(t.n, u0, u1, v0, v1, &nlo, &nhi);
    fn_range(t.p, u0, u1, v0, v1, &plo, &phi);
    fn_range(t.q, u0, u1, v0, v1, &qlo, &qhi);
    fn_range(t.s, u0, u1, v0, v1, &slo, &shi);
    const float T = 2.384185791015625e-07f;      // |e1e2b| below 2^-22 may underflow t to +-0, which passes t >= 0
    const bool pos = (nhi > -t.n.m) && (phi >= -t.p.m) && (qhi >= -t.q.m) && (shi >= -t.s.m) && (t.nb > -T);
    const bool neg = (nlo < t.n.m) && (plo <= t.p.m) && (qlo <= t.q.m) && (slo <= t.s.m) && (t.nb < T);
    const bool box = (t.bstate == BOX_NONE) ||
                     (t.bstate == BOX_VALID && u1 >= t.bu0 && u0 <= t.bu1 && v1 >= t.bv0 && v0 <= t.bv1);
    return (pos || neg) && box;
}

// Which light-cube face and bin a shadow ray with negD = rd belongs to.  Returns the global bin index.
__device__ __forceinline__ uint32_t cube_bin_of(v3 rd, uint32_t face_base0 /* base of face 0 of this light */, int cube_bins)
{
    const float ax = fabsf(rd.x), ay = fabsf(rd.y), az = fabsf(rd.z);
    int k;
    float m, a, b, sgn;
    if (ax >= ay && ax >= az) { k = 0; m = ax; sgn = rd.x; a = rd.y; b = rd.z; }
    else if (ay >= az) { k = 1; m = ay; sgn = rd.y; a = rd.z; b = rd.x; }
    else { k = 2; m = az; sgn = rd.z; a = rd.x; b = rd.y; }
    const int face = 2 * k + (sgn < 0.0f ? 1 : 0);
    // u, v in [-1,1]; NaN (degenerate ray, never accepted by any triangle) falls into bin 0
    const float u = a / m, v = b / m;
    const float half = 0.5f * (float)cube_bins;
    int i = (int)floorf((u + 1.0f) * half);
    int j = (int)floorf((v + 1.0f) * half);
    i = min(max(i, 0), cube_bins - 1);
    j = min(max(j, 0), cube_bins - 1);
    return face_base0 + (uint32_t)face * (uint32_t)(cube_bins * cube_bins) + (uint32_t)j * cube_bins + (uint32_t)i;
}

}  // namespace mirt
