// rt_binned.hip -- binning kernels and the binned ray-trace kernel (see rt_binned.hpp for the argument why
// the candidate reduction cannot change any result).
#include "rt_binned.hpp"

#include <float.h>

namespace mirt {

// ---- binning: (bin, triangle) pairs in one pass, ordered by a radix sort ----------------------------------------
//
// k_bin_pairs: persistent workgroups; a work item is (256 consecutive triangles, one frame), one thread per triangle.
// The thread sets up the folded edge functions and the (u,v) box of its triangle in that frame once.  Then
//   * boxes of at most 32 x 32 bins (everything in a large mesh): the (triangle, bin-of-its-box) tests of the whole
//     workgroup are FLATTENED -- an exclusive prefix of the box sizes in LDS, and thread t of each round takes test
//     number t (binary search for its triangle, constants from LDS) -- so a triangle that covers 300 bins and one that
//     covers 2 cost what they should, and every lane works.  (First version: one wave per such triangle walking 64 bins
//     at a time with 5 of 64 lanes passing: 250 us.)
//   * larger boxes or none (walls, triangles crossing a cube face's plane): the WAVE takes such lanes one at a time
//     (ballot loop, item broadcast with v_readlane) through three levels of the same conservative rectangle test --
//     lane = 64x64-bin cell, lane = 8x8-bin coarse cell, lane = bin -- after the LANE has tried level 0 itself, which
//     already discards most of them.
// Passing (bin, triangle) pairs are appended to an LDS buffer (one LDS atomic per wave and round) and leave through
// coalesced copies with ONE atomic on the global list's length per flush.
// bin_sort.hip then orders the pairs by bin (rocPRIM radix sort over the bits a bin id needs) and k_bin_offsets finds
// every bin's range by binary search: `entries` = the sorted triangle ids, `bin_off` as before.
//
// Why not count / scan / fill with one atomic per pair (the first version): a CU retires one global atomic per ~24
// cycles (tools/sortbench.hip: 25 G atomics/s chip-wide, returning or not), the fill pass's atomic -> dependent store
// chain made it latency-bound (174 + 365 us measured for the 1.8 M pairs of the 100 k soup), and atomics on ONE
// address (a list length bumped per workgroup) retire one per ~7 ns.  Sorting the same pairs takes ~80 us.
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
    float A0[4], A1[4], A2[4], Bu[4], Bv[4];   // level-0 (K = 64), level-1 (K = 8) and level-2 (K = 1) constants, per-bin slopes
    float lou, hiu, lov, hiv;                  // box as bin-index ranges: cell [I, I+K) overlaps iff I+K >= lou && I <= hiu
};

__device__ __forceinline__ float bcastf(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }

__device__ __forceinline__ float corner(float s, float lo, float hi) { return fmaxf(s * lo, s * hi); }

// the cell of K x K bins that starts at bin (I, J): may it hold an accepted ray?  A = the item's constants for K
__device__ __forceinline__ bool cell_may_hit(const BinItem &u, const float (&A)[4], float I, float J, float K)
{
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 4; k++) ok = ok && (__builtin_fmaf(J, u.Bv[k], __builtin_fmaf(I, u.Bu[k], A[k])) >= 0.0f);
    return ok && (I + K >= u.lou) && (I <= u.hiu) && (J + K >= u.lov) && (J <= u.hiv);
}

// ---- joint emptiness of a huge item's accept region inside the frame ---------------------------------------------
// Every accepted ray has all four folded functions G_k(u,v) = sgn*g_k(u,v) + m_k >= 0 at its (u,v) (that is what the cell
// tests rely on), and lies inside the frame's rectangle: eight half-planes.  The cell tests look at one function at a time
// (is it >= 0 SOMEWHERE in the cell?), which cannot see that the half-planes have no COMMON point -- a triangle behind a
// cube face's plane, or crossing it far outside the face, passes them in a dozen bins around the crossings of its edge
// lines.  By Helly's theorem the eight convex sets have a common point iff every three of them have, and three half-planes
// a_i + b_i.x >= 0 have none exactly when some weights w_i >= 0 (not all zero) give sum(w_i b_i) = 0 and sum(w_i a_i) < 0
// (Farkas); in the plane those weights are the pairwise cross products of the gradients.  Triples with two or three
// rectangle sides are the corner tests level 0 already does, so 4 + 6*4 triples remain.  Rounding: the computed weights
// are off by at most 2^-22 of their two products, which leaves a residual gradient; the certificate is accepted only below
// minus 2^-20 * sum(W_i * (|a_i| + |b_i|.(U,V))) with W_i the sum of the two product magnitudes.
struct BinLin { float a, bu, bv; };

__device__ __forceinline__ bool bin_triple_infeasible(const BinLin &f, const BinLin &g, const BinLin &h, float U, float V)
{
    const float f1 = g.bu * h.bv, f2 = g.bv * h.bu, g1 = h.bu * f.bv, g2 = h.bv * f.bu, h1 = f.bu * g.bv, h2 = f.bv * g.bu;
    float wf = f1 - f2, wg = g1 - g2, wh = h1 - h2;
    const bool pos = wf >= 0.0f && wg >= 0.0f && wh >= 0.0f, neg = wf <= 0.0f && wg <= 0.0f && wh <= 0.0f;
    if (!(pos || neg)) return false;                 // the gradients do not positively span the plane (also NaN)
    if (neg) { wf = -wf; wg = -wg; wh = -wh; }
    if (!(wf + wg + wh > 0.0f)) return false;
    const float C = wf * f.a + wg * g.a + wh * h.a;
    const float slack = (fabsf(f1) + fabsf(f2)) * (fabsf(f.a) + fabsf(f.bu) * U + fabsf(f.bv) * V) +
                        (fabsf(g1) + fabsf(g2)) * (fabsf(g.a) + fabsf(g.bu) * U + fabsf(g.bv) * V) +
                        (fabsf(h1) + fabsf(h2)) * (fabsf(h.a) + fabsf(h.bu) * U + fabsf(h.bv) * V);
    return C < -9.5367431640625e-07f * slack;        // 2^-20
}

// true: no (u,v) of the frame's rectangle [u0,u1] x [v0,v1] satisfies all four G_k >= 0
__device__ __forceinline__ bool bin_jointly_empty(const BinLin (&G)[4], float u0, float u1, float v0, float v1)
{
    const float U = fmaxf(fabsf(u0), fabsf(u1)), V = fmaxf(fabsf(v0), fabsf(v1));
    const BinLin side[4] = { { -u0, 1.0f, 0.0f }, { u1, -1.0f, 0.0f }, { -v0, 0.0f, 1.0f }, { v1, 0.0f, -1.0f } };
    bool empty = false;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = i + 1; j < 4; j++) {
#pragma unroll
            for (int k = j + 1; k < 4; k++) empty = empty || bin_triple_infeasible(G[i], G[j], G[k], U, V);
#pragma unroll
            for (int sd = 0; sd < 4; sd++) empty = empty || bin_triple_infeasible(G[i], G[j], side[sd], U, V);
        }
    return empty;
}

struct BinFrameGrid { int nbu, fj0, fj1, cells_x, cy0, ncell; uint32_t fbase, nshell; };

constexpr int BIN_PAIR_BUF = 4096;                    // pairs a workgroup stages in LDS between flushes (32 KiB)
constexpr int BIN_TESTS_PER_THREAD = 2;               // flattened bin tests a thread runs per round (independent chains: their LDS latencies overlap)
constexpr int BIN_DIRECT_SIDE = 32;                   // boxes up to 32 x 32 bins are tested bin by bin, flattened over the workgroup

// Where the pairs of the huge items go: the workgroup's LDS buffer, allocated with an LDS atomic per level-2 step, and
// once that is full straight to the global list (one global atomic per step -- only scenes of huge triangles get there).
struct BinLargeSink {
    uint32_t *s_keys, *s_vals;    // LDS, BIN_PAIR_BUF each
    uint32_t *s_fill;             // LDS: slots handed out so far (may run past BIN_PAIR_BUF)
    uint32_t *s_valid;            // LDS: slots [0, *s_valid) hold pairs (the step that did not fit any more lowers it)
    uint32_t *g_count;            // the global list's length
    BinPairs out;
    uint32_t *bucket_cnt;         // nullable: the sort's pairs-per-bucket counters (pairs that bypass the LDS buffer count here)
    int bucket_shift;
};

// Walks one large item (wave-uniform `u`) through the three levels and emits its pairs.
// `cells0` = the owner lane's level-0 verdicts for the first 64 cells (wave-uniform).
__device__ __forceinline__ void bin_walk_large(const BinItem &u, uint32_t utri, uint32_t ushell, const BinFrameGrid &gr, int lane,
                                               const BinLargeSink &sink, unsigned long long cells0)
{
#ifdef MIRT_BIN_STATS
    uint32_t dbg_steps = 0, dbg_pairs = 0, dbg_l1 = 0;
#endif
    for (int c0 = 0; c0 < gr.ncell; c0 += 64) {
        // level 0: lane = cell of 64x64 bins
        const int c = c0 + lane, cx = c % gr.cells_x, cy = gr.cy0 + c / gr.cells_x;
        unsigned long long m0;
        if (c0 == 0 && gr.ncell <= 64) m0 = cells0;
        else m0 = __ballot(c < gr.ncell && cell_may_hit(u, u.A0, (float)(cx * BIN_L0), (float)(cy * BIN_L0), (float)BIN_L0));
        while (m0) {
            const int l0 = __builtin_ctzll(m0);
            m0 &= m0 - 1ull;
            const int ucx = __builtin_amdgcn_readlane(cx, l0), ucy = __builtin_amdgcn_readlane(cy, l0);
            // level 1: lane = coarse cell (8x8 bins) inside the level-0 cell
            const int ci = ucx * BIN_L0 + (lane & 7) * BIN_COARSE, cj = ucy * BIN_L0 + (lane >> 3) * BIN_COARSE;
            const bool pass1 = ci < gr.nbu && cj + BIN_COARSE > gr.fj0 && cj < gr.fj1 && cell_may_hit(u, u.A1, (float)ci, (float)cj, (float)BIN_COARSE);
            unsigned long long m1 = __ballot(pass1);
#ifdef MIRT_BIN_STATS
            dbg_l1++;
#endif
            while (m1) {
                const int cl = __builtin_ctzll(m1);
                m1 &= m1 - 1ull;
                // level 2: lane = bin inside coarse cell `cl`
                const int i = ucx * BIN_L0 + (cl & 7) * BIN_COARSE + (lane & 7);
                const int j = ucy * BIN_L0 + (cl >> 3) * BIN_COARSE + (lane >> 3);
                const bool pass2 = i < gr.nbu && j >= gr.fj0 && j < gr.fj1 && cell_may_hit(u, u.A2, (float)i, (float)j, 1.0f);
                const unsigned long long m2 = __ballot(pass2);
#ifdef MIRT_BIN_STATS
                dbg_steps++; dbg_pairs += (uint32_t)__popcll(m2);
#endif
                if (!m2) continue;
                const uint32_t np = (uint32_t)__popcll(m2), rank = (uint32_t)__popcll(m2 & ((1ull << lane) - 1ull));
                const uint32_t key = (gr.fbase + (uint32_t)j * gr.nbu + i) * gr.nshell + ushell;
                uint32_t at0 = 0;
                if (lane == 0) at0 = atomicAdd(sink.s_fill, np);
                at0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)at0);
                if (at0 + np <= (uint32_t)BIN_PAIR_BUF) {
                    if (pass2) { sink.s_keys[at0 + rank] = key; sink.s_vals[at0 + rank] = utri; }
                } else {
                    if (lane == 0) {
                        if (at0 < (uint32_t)BIN_PAIR_BUF) *sink.s_valid = at0;     // exactly one step straddles the end of the buffer
                        at0 = atomicAdd(sink.g_count, np);
                    }
                    at0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)at0);
                    if (pass2 && at0 + rank < sink.out.cap) { sink.out.keys[at0 + rank] = key; sink.out.vals[at0 + rank] = utri; }
                    if (pass2 && sink.bucket_cnt) atomicAdd(&sink.bucket_cnt[key >> sink.bucket_shift], 1u);
                }
            }
        }
    }
#ifdef MIRT_BIN_STATS
    if (lane == 0) {
        uint32_t *c = sink.g_count;      // counters[0]; the statistics live behind it
        atomicAdd(&c[2], 1u); atomicAdd(&c[3], dbg_l1); atomicAdd(&c[4], dbg_steps); atomicAdd(&c[5], dbg_pairs);
        atomicMax(&c[6], dbg_steps);
        if (dbg_steps > 100) atomicAdd(&c[7], 1u);
    }
#endif
}

// ---- k_prep_select: the camera's origin rows for the triangles the frame can see, and their list ---------------------------------
// First kernel of a binned frame (it takes over k_prep_origin's duties there).  One thread per triangle: the origin row -- the
// very code of k_prep_origin, so the same bits --, then frame_may_see (rt_binned.hpp).  A triangle that may be seen gets its row
// written and its index appended to `sel` (one global atomic per wave); k_bin_pairs walks that list instead of the scene, and a
// frame that falls back to brute force (k_rt_trace2) walks it too.  For a band of a sharded frame that is an eighth of the scene,
// which is what makes the band's binning cost follow the band.
// The histogram (optional): every boxed triangle adds the bins of its box, row by row, to the coarse tile rows of the WHOLE
// frame -- whatever rows this call renders -- an estimate of where the frame's (tile, triangle) pairs lie that every rank of a
// sharded frame computes identically (integer sums), so that all of them derive the same cost-weighted bands without exchanging
// anything (mirt_capi.hip: weighted partition).  Kept in LDS per workgroup, flushed once.
constexpr int SEL_WG = 1024;                          // threads of a k_prep_select workgroup: one workgroup per CU, 4 waves per SIMD
constexpr int SEL_STAGE = 8192;                       // indices a workgroup stages in LDS between hand-overs to the global list (32 KiB)

// (An atomic per wave on the list's length -- the first version -- serialised: one address retires an atomic every ~7 ns, and the
// 15 600 waves of a 1 M-triangle scene made that 109 us, more than everything the selection saves.  Now a workgroup stages the
// indices it keeps in LDS and reserves its slice of the list ONCE, when it has walked all its chunks or the stage is full: 256
// atomics per launch, and the indices leave as coalesced runs.)
__global__ __launch_bounds__(SEL_WG) void k_prep_select(const float *__restrict__ tris15, int n, const BinFrameDesc fr, const SelectOut out)
{
    __shared__ uint32_t s_hist[SEL_HIST_MAX];
    __shared__ uint32_t s_sel[SEL_STAGE];
    __shared__ uint32_t s_fill, s_base;
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0) {
        if (out.zero_hits)
            for (int g = threadIdx.x; g < HIT_SHARDS * HIT_SHARD_STRIDE; g += SEL_WG) out.zero_hits[g] = 0ull;
        if (out.zero_counter && (threadIdx.x == 0 || (threadIdx.x >= 16 && threadIdx.x < 80))) out.zero_counter[threadIdx.x] = 0u;
        if (threadIdx.x == 0) *out.sel_count_next = 0u;
        if (out.zero_faces && (int)threadIdx.x < out.zero_faces_n) out.zero_faces[threadIdx.x] = 0u;
    }
    if (out.hist && threadIdx.x < SEL_HIST_MAX) s_hist[threadIdx.x] = 0u;
    if (threadIdx.x == 0) s_fill = 0u;
    __syncthreads();
    const v3 S = V3(fr.S[0], fr.S[1], fr.S[2]);
    const int nchunks = (n + SEL_WG - 1) / SEL_WG;
    const int hist_rows = ((fr.nbv - 1) >> out.hist_shift) + 1;
    // hands the staged indices over to the global list (called by all threads)
    auto flush = [&]() {
        __syncthreads();
        const uint32_t staged = s_fill;
        if (threadIdx.x == 0 && staged) s_base = atomicAdd(out.sel_count, staged);
        __syncthreads();
        const uint32_t base = s_base;
        for (uint32_t i = threadIdx.x; i < staged; i += SEL_WG) out.sel[base + i] = s_sel[i];
        __syncthreads();
        if (threadIdx.x == 0) s_fill = 0u;
        __syncthreads();
    };
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        if (s_fill + (uint32_t)SEL_WG > (uint32_t)SEL_STAGE) flush();       // (uniform: s_fill only changes between the barriers below)
        const int i = c * SEL_WG + (int)threadIdx.x;
        bool keep = false;
        OriginRow r;
        if (i < n) {
            const float *t15 = tris15 + (size_t)15 * i;
            r = make_origin_row(t15, S);
            PreBox box;
            bool boxed;
            keep = frame_may_see(r, ld3(t15), ld3(t15 + 3), ld3(t15 + 6), fr, &box, &boxed);
            if (out.hist && boxed) {
                // the bins of the box, clamped to the grid (floats far outside the int range clamp first)
                const float flo_u = fmaxf(ceilf(box.lou) - 1.0f, 0.0f), fhi_u = fminf(floorf(box.hiu), (float)(fr.nbu - 1));
                const float flo_v = fmaxf(ceilf(box.lov) - 1.0f, 0.0f), fhi_v = fminf(floorf(box.hiv), (float)(fr.nbv - 1));
                if (flo_u <= fhi_u && flo_v <= fhi_v) {
                    const uint32_t wi = (uint32_t)((int)fhi_u - (int)flo_u + 1);
                    const int ja = (int)flo_v, jb = (int)fhi_v;
                    for (int cr = ja >> out.hist_shift; cr <= (jb >> out.hist_shift) && cr < hist_rows; cr++) {
                        const int lo = max(ja, cr << out.hist_shift), hi = min(jb, ((cr + 1) << out.hist_shift) - 1);
                        atomicAdd(&s_hist[cr], wi * (uint32_t)(hi - lo + 1));
                    }
                }
            }
            if (keep) out.cam_tab[i] = r;
        }
        // the wave's kept indices into the stage: one LDS atomic per wave
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        if (m) {
            uint32_t at0 = 0;
            if (lane == 0) at0 = atomicAdd(&s_fill, (uint32_t)__popcll(m));
            at0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)at0);
            if (keep) s_sel[at0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)i;
        }
        __syncthreads();                                  // s_fill as every thread will read it at the top of the next round
    }
    flush();
    if (out.hist && threadIdx.x < SEL_HIST_MAX) {
        const uint32_t v = s_hist[threadIdx.x];
        if (v && (int)threadIdx.x < hist_rows) atomicAdd(&out.hist[threadIdx.x], v);
    }
}

// ---- k_select_faces: the lights' origin rows, and per cube face the triangles it can see ---------------------------------------
// A light's cube has six faces and a triangle lies in front of three of them and projects into one, give or take: binning every
// triangle for every face ran the thousand-instruction set-up six times per (triangle, light) to find five of them empty.  This
// kernel -- it takes over k_prep_origin's work for the lights of a frame that bins its own cubes, and of the shared cube's build --
// writes the origin row of (triangle, light) and puts the triangle on the list of every face frame_may_see does not rule out
// (rt_binned.hpp: clearly behind the face's plane and far from degenerate, or clearly in front and outside the face's square);
// k_bin_pairs walks the lists.  grid.y = light, a workgroup of 1024 threads strides over the triangles; kept indices are staged
// per face in LDS and leave with one atomic per (workgroup, face, hand-over).  face_counts: zero on entry (k_prep_select zeroes
// them for a frame; the cube build fills them itself).
constexpr int FACE_STAGE = 2048;                      // indices per face a workgroup stages between hand-overs (6 x 8 KiB)
__global__ __launch_bounds__(SEL_WG) void k_select_faces(const float *__restrict__ tris15, int n, const float *__restrict__ origins /* (1 + nl) x 3 */,
                                                        const BinFrameDesc *__restrict__ faces /* 6 per light */, OriginRow *__restrict__ light_tab,
                                                        uint32_t *__restrict__ lists, uint32_t stride, uint32_t *__restrict__ face_counts)
{
    __shared__ uint32_t s_sel[6][FACE_STAGE];
    __shared__ uint32_t s_fill[6], s_base[6];
    const int lane = threadIdx.x & 63, k = (int)blockIdx.y;
    if (threadIdx.x < 6) s_fill[threadIdx.x] = 0u;
    __syncthreads();
    const v3 S = ld3(origins + 3 * (1 + k));
    const int nchunks = (n + SEL_WG - 1) / SEL_WG;
    auto flush = [&]() {
        __syncthreads();
        if (threadIdx.x < 6 && s_fill[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&face_counts[k * 6 + (int)threadIdx.x], s_fill[threadIdx.x]);
        __syncthreads();
#pragma unroll
        for (int f = 0; f < 6; f++) {
            const uint32_t staged = s_fill[f], base = s_base[f];
            uint32_t *dst = lists + (size_t)(k * 6 + f) * stride + base;
            for (uint32_t i = threadIdx.x; i < staged; i += SEL_WG) dst[i] = s_sel[f][i];
        }
        __syncthreads();
        if (threadIdx.x < 6) s_fill[threadIdx.x] = 0u;
        __syncthreads();
    };
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        uint32_t most = 0;
#pragma unroll
        for (int f = 0; f < 6; f++) most = max(most, s_fill[f]);
        if (most + (uint32_t)SEL_WG > (uint32_t)FACE_STAGE) flush();        // (uniform: the fills only change between the barriers below)
        const int i = c * SEL_WG + (int)threadIdx.x;
        uint32_t keep = 0;
        if (i < n) {
            const float *t15 = tris15 + (size_t)15 * i;
            OriginRow r = make_origin_row(t15, S);
            const v3 va = ld3(t15), vb = ld3(t15 + 3), vc = ld3(t15 + 6);
            r.r2.w = origin_far(va, vb, vc, S);           // the spare word: no point of the triangle is farther from this light (k_expand_light_rows copies it)
            light_tab[(size_t)k * n + i] = r;
            for (int f = 0; f < 6; f++) {
                PreBox box;
                bool boxed;
                if (frame_may_see(r, va, vb, vc, faces[k * 6 + f], &box, &boxed)) keep |= 1u << f;
            }
        }
#pragma unroll
        for (int f = 0; f < 6; f++) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64((keep >> f) & 1u);
            if (m) {
                uint32_t at0 = 0;
                if (lane == 0) at0 = atomicAdd(&s_fill[f], (uint32_t)__popcll(m));
                at0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)at0);
                if ((keep >> f) & 1u) s_sel[f][at0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)i;
            }
        }
        __syncthreads();
    }
    flush();
}

// What the flattened bin-by-bin tests need of a direct item, in LDS: A2, Bu, Bv and box = {lou, hiu, lov, hiv}, one array per
// field (neighbouring lanes read neighbouring items: 16-byte stride, where a 64-byte record put them on the same banks).

template <int WG>
__global__ __launch_bounds__(WG) void k_bin_pairs(const float *__restrict__ tris15, const OriginRow *__restrict__ cam_tab,
                                                   const OriginRow *__restrict__ light_tab, int n, BinSet bs, BinPairs out)
{
    // pairs per bucket (bin >> bucket_shift) for the sort that follows (bin_bucket_sort.hip): bs.bucket_cnt != NULL
    extern __shared__ uint32_t s_bucket[];
    __shared__ uint32_t s_keys[BIN_PAIR_BUF], s_vals[BIN_PAIR_BUF];
    __shared__ float4 s_A2[256], s_Bu[256], s_Bv[256], s_box[256];
    __shared__ uint32_t s_org[256];                   // i_lo | j_lo << 16 of a direct item's box
    __shared__ uint32_t s_ni[256];                    // its width in bins
    __shared__ uint32_t s_tri[256];                   // its triangle
    __shared__ uint32_t s_pre[257];                   // exclusive prefix of the box sizes
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_base, s_fill, s_valid;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // triangles per work item: 256 (one per thread), or 64 when that leaves the chip under-filled -- a single camera frame of
    // 100 k triangles is only 391 items of 256 for 256 CUs, each a serial ~10 us; with 64 the setup of an item occupies one
    // wave and its flattened tests still use all four
    const int chunk_tris = bs.chunk_tris;
    // Work items: (chunk of the scene, frame) -- except for the frame of the camera's origin table when k_prep_select has listed
    // the triangles it can see (bs.sel): that frame's items are the chunks of the list, and they come first.
    const int nchunks = (n + chunk_tris - 1) / chunk_tris;
    const bool listed = bs.sel != nullptr;            // (then frame 0 is the camera's: rt_enqueue_binned)
    const uint32_t nsel = listed ? min(*bs.sel_count, (uint32_t)n) : 0u;
    // a SHORT list (the 17 000 triangles the top band of the 1 M-triangle frame at 8K sees: 67 items of 256 for 256 CUs, each a
    // serial ~10 us of set-up) goes in items of 64: the set-up of an item then occupies one wave, its flattened tests still all eight
    const int cam_chunk = (listed && nsel < 32768u) ? 64 : chunk_tris;
    const int ncam = listed ? (int)((nsel + (uint32_t)cam_chunk - 1u) / (uint32_t)cam_chunk) : 0;
    // The other frames -- the light-cube faces -- each walk the scene, or, when k_select_faces has listed what each face can see
    // (bs.face_lists), their lists: s_first[f] = the first work item of face frame f (items of chunk_tris triangles), an exclusive
    // prefix over at most MAX_BIN_FRAMES frames, computed by every workgroup for itself.
    const int nrest = listed ? bs.nframes - 1 : bs.nframes;
    const bool faces_listed = bs.face_lists != nullptr;
    __shared__ uint32_t s_first[MAX_BIN_FRAMES + 1];
    if (faces_listed) {
        if (threadIdx.x == 0) {
            uint32_t run = 0;
            for (int ff = 0; ff < nrest; ff++) { s_first[ff] = run; run += (min(bs.face_counts[ff], (uint32_t)n) + (uint32_t)chunk_tris - 1u) / (uint32_t)chunk_tris; }
            s_first[nrest] = run;
        }
        __syncthreads();
    }
    const int nwork = ncam + (faces_listed ? (int)s_first[nrest] : nchunks * nrest);
    if (threadIdx.x == 0) { s_fill = 0u; s_valid = (uint32_t)BIN_PAIR_BUF; }
    if (bs.bucket_cnt)
        for (uint32_t b = threadIdx.x; b < bs.nbuckets; b += WG) s_bucket[b] = 0u;

    // hands the staged pairs over to the global list (called by all threads)
    auto flush = [&]() {
        __syncthreads();
        const uint32_t staged = min(s_fill, s_valid);
        if (threadIdx.x == 0 && staged) s_base = atomicAdd(&bs.counters[0], staged);
        __syncthreads();
        const uint32_t base = s_base;
        for (uint32_t i = threadIdx.x; i < staged; i += WG) {
            const uint32_t at = base + i;
            if (at < out.cap) { out.keys[at] = s_keys[i]; out.vals[at] = s_vals[i]; }
            if (bs.bucket_cnt) atomicAdd(&s_bucket[s_keys[i] >> bs.bucket_shift], 1u);
        }
        __syncthreads();
        if (bs.bucket_cnt && staged)
            for (uint32_t b = threadIdx.x; b < bs.nbuckets; b += WG) {
                const uint32_t c = s_bucket[b];
                if (c) { atomicAdd(&bs.bucket_cnt[b], c); s_bucket[b] = 0u; }
            }
        if (threadIdx.x == 0) { s_fill = 0u; s_valid = (uint32_t)BIN_PAIR_BUF; }
        __syncthreads();
    };

#ifdef MIRT_BIN_STAMPS
    long long st_setup = 0, st_prefix = 0, st_rounds = 0, st_huge = 0, st_flush = 0, st_t;
    const long long st_begin = __builtin_amdgcn_s_memtime();
    int st_nrounds = 0;
#define STAMP(acc) { const long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - st_t; st_t = now_; }
#else
#define STAMP(acc)
#endif
    for (int w = blockIdx.x; w < nwork; w += gridDim.x) {
#ifdef MIRT_BIN_STAMPS
        st_t = __builtin_amdgcn_s_memtime();
#endif
        int chunk, frame;
        const uint32_t *list = nullptr;                   // the item's triangles: a chunk of this list, or of the scene
        uint32_t nhere = (uint32_t)n;
        const bool from_cam_list = w < ncam;
        if (from_cam_list) { chunk = w; frame = 0; list = bs.sel; nhere = nsel; }
        else if (faces_listed) {
            const uint32_t w2 = (uint32_t)(w - ncam);
            int lo = 0, hi = nrest;                       // the face frame whose items [s_first[f], s_first[f + 1]) hold w2
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_first[mid] <= w2) lo = mid; else hi = mid; }
            chunk = (int)(w2 - s_first[lo]);
            frame = lo + (listed ? 1 : 0);
            list = bs.face_lists + (size_t)lo * bs.face_stride;
            nhere = min(bs.face_counts[lo], (uint32_t)n);
        } else { const int w2 = w - ncam; chunk = w2 / nrest; frame = w2 - chunk * nrest + (listed ? 1 : 0); }
        const int item_tris = from_cam_list ? cam_chunk : chunk_tris;
        const uint32_t slot = (uint32_t)chunk * (uint32_t)item_tris + threadIdx.x;       // place in the list or in the scene
        const uint32_t tri = ((int)threadIdx.x < item_tris && slot < nhere) ? (list ? list[slot] : slot) : 0xFFFFFFFFu;
        const BinFrameDesc fr = bs.frames ? bs.frames[frame] : bs.frame0;
        BinFrameGrid gr;
        gr.nbu = fr.nbu; gr.fj0 = fr.j0; gr.fj1 = fr.j1; gr.fbase = fr.base; gr.nshell = (uint32_t)max(fr.nshell, 1);
        gr.cells_x = (gr.nbu + BIN_L0 - 1) / BIN_L0; gr.cy0 = gr.fj0 / BIN_L0;
        gr.ncell = gr.cells_x * ((gr.fj1 + BIN_L0 - 1) / BIN_L0 - gr.cy0);
        const int nbu = gr.nbu, fj0 = gr.fj0, fj1 = gr.fj1;
        BinItem it;
        memset(&it, 0, sizeof it);
        int kind = 0;                                     // 0: no bin can hold a hit, 1: direct (box of at most 32 x 32 bins), 2: huge (wave walks)
#ifdef MIRT_BIN_STATS
        int dbg_state = 0;
#endif
        int i_lo = 0, i_hi = -1, j_lo = 0, j_hi = -1;
        unsigned long long cells = 0;                     // huge items: level-0 cells (first 64) that may hold a hit
        uint32_t shell = 0;                               // depth shell of the triangle in this frame (orders the bins' lists)
        if (tri < (uint32_t)n && fj1 > fj0) {
            const OriginRow &row = (fr.tab == 0) ? cam_tab[tri] : light_tab[(size_t)(fr.tab - 1) * n + tri];
            shell = bin_shell_of(fr, row.r1.w);
            TriBinFns t = make_bin_fns(row, fr);
            add_bbox(t, tris15 + (size_t)15 * tri, fr);
#ifdef MIRT_BIN_STATS
            dbg_state = t.bstate == BOX_VALID ? 0 : (t.bstate == BOX_NONE ? 1 : 2);
#endif
            // t = e1e2b / e1e2d passes `t >= 0` with the "wrong" sign of e1e2d only if it comes out as -0, i.e. e1e2b is
            // zero or the quotient underflows past the smallest subnormal 2^-149.  Binning runs only for operands the host
            // has bounded (|coordinate| < 1e8, |negD| < 1e6: mirt_capi.hip `safe`), so |e1e2d| < 2^79 and the quotient
            // cannot underflow unless |e1e2b| < 2^-70: below 2^-69 either sign of e1e2d is allowed.  (rect_may_hit keeps
            // the scene-independent 2^-22; with that here, the handful of triangles of a 100 k soup whose plane passes
            // within 1e-4 of the camera were put into every bin of the screen.)
            const float T = 1.6940658945086007e-21f;
            const bool both = fabsf(t.nb) < T || !(t.nb == t.nb);
            const float sgn = t.nb < 0.0f ? -1.0f : 1.0f;
            const EdgeFn *fn[4] = { &t.n, &t.p, &t.q, &t.s };
            const float inf = __builtin_huge_valf();
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float su = sgn * fn[k]->cu, sv = sgn * fn[k]->cv;
                // value at the (unpadded) origin corner of bin (0,0), margin folded in
                const float base = sgn * fn[k]->c0 + fn[k]->m + su * fr.ulo + sv * fr.vlo;
                it.Bu[k] = both ? 0.0f : su * fr.du;
                it.Bv[k] = both ? 0.0f : sv * fr.dv;
                it.A0[k] = both ? inf : base + corner(su, fr.pad_lo, (float)BIN_L0 * fr.du + fr.pad_hi) + corner(sv, fr.pad_lo, (float)BIN_L0 * fr.dv + fr.pad_hi);
                it.A1[k] = both ? inf : base + corner(su, fr.pad_lo, (float)BIN_COARSE * fr.du + fr.pad_hi) + corner(sv, fr.pad_lo, (float)BIN_COARSE * fr.dv + fr.pad_hi);
                it.A2[k] = both ? inf : base + corner(su, fr.pad_lo, fr.du + fr.pad_hi) + corner(sv, fr.pad_lo, fr.dv + fr.pad_hi);
            }
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
            // The bins the box admits: bin i passes `i + 1 >= lou && i <= hiu` exactly when ceil(lou) - 1 <= i <= floor(hiu);
            // clamped to the frame's grid.  (NaN bounds fail every comparison, here and in cell_may_hit alike.)
            const float flo_u = fmaxf(ceilf(it.lou) - 1.0f, 0.0f), fhi_u = fminf(floorf(it.hiu), (float)(nbu - 1));
            const float flo_v = fmaxf(ceilf(it.lov) - 1.0f, (float)fj0), fhi_v = fminf(floorf(it.hiv), (float)(fj1 - 1));
            if (flo_u <= fhi_u && flo_v <= fhi_v) {
                i_lo = (int)flo_u; i_hi = (int)fhi_u; j_lo = (int)flo_v; j_hi = (int)fhi_v;
                kind = (i_hi - i_lo < BIN_DIRECT_SIDE && j_hi - j_lo < BIN_DIRECT_SIDE) ? 1 : 2;
            }
            // Level 0 of a huge item by the lane itself: most of them (triangles a cube face sees edge-on or behind its
            // plane have no usable box, so they land here) fail every 64x64-bin cell and never occupy the wave.
            if (kind == 2) {
                for (int c = 0; c < gr.ncell && c < 64; c++) {
                    const int cx = c % gr.cells_x, cy = gr.cy0 + c / gr.cells_x;
                    if (cell_may_hit(it, it.A0, (float)(cx * BIN_L0), (float)(cy * BIN_L0), (float)BIN_L0)) cells |= 1ull << c;
                }
                if (gr.ncell > 64) cells = ~0ull;          // more cells than mask bits (frames beyond 4096 x 4096 bins): the walk tests them
                if (!cells) kind = 0;
                // ... and the joint test: is there ANY point of the frame where all four functions hold at once?
                if (kind == 2 && !both) {
                    BinLin G[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) { G[k].a = sgn * fn[k]->c0 + fn[k]->m; G[k].bu = sgn * fn[k]->cu; G[k].bv = sgn * fn[k]->cv; }
                    const float u0 = fr.ulo + fr.pad_lo, u1 = fr.ulo + (float)nbu * fr.du + fr.pad_hi;
                    const float v0 = fr.vlo + (float)fj0 * fr.dv + fr.pad_lo, v1 = fr.vlo + (float)fj1 * fr.dv + fr.pad_hi;
                    if (bin_jointly_empty(G, u0, u1, v0, v1)) kind = 0;
                }
            }
        }



        STAMP(st_setup)
        // ---- direct items: constants to LDS, exclusive prefix of their box sizes over the workgroup ----
        const uint32_t ni = (uint32_t)(i_hi - i_lo + 1);
        const uint32_t nb = kind == 1 ? ni * (uint32_t)(j_hi - j_lo + 1) : 0u;
        uint32_t incl = nb;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        __syncthreads();                              // the previous work item's readers of the LDS tables are done
        if (lane == 63 && wave < 4) s_wave[wave] = incl;     // (the triangles of an item sit in the first four waves)
        if (kind == 1) {
            s_A2[threadIdx.x] = make_float4(it.A2[0], it.A2[1], it.A2[2], it.A2[3]);
            s_Bu[threadIdx.x] = make_float4(it.Bu[0], it.Bu[1], it.Bu[2], it.Bu[3]);
            s_Bv[threadIdx.x] = make_float4(it.Bv[0], it.Bv[1], it.Bv[2], it.Bv[3]);
            s_box[threadIdx.x] = make_float4(it.lou, it.hiu, it.lov, it.hiv);
            s_org[threadIdx.x] = (uint32_t)i_lo | ((uint32_t)j_lo << 16);
            s_ni[threadIdx.x] = ni | (shell << 8);
            s_tri[threadIdx.x] = tri;
        }
        __syncthreads();
        const uint32_t t0w = s_wave[0], t1w = s_wave[1], t2w = s_wave[2], t3w = s_wave[3];
        if (threadIdx.x < 256) s_pre[threadIdx.x] = (wave > 0 ? t0w : 0u) + (wave > 1 ? t1w : 0u) + (wave > 2 ? t2w : 0u) + incl - nb;
        const uint32_t T = t0w + t1w + t2w + t3w;
        if (threadIdx.x == 0) s_pre[256] = T;

#ifdef MIRT_BIN_STATS
        if (kind == 2) { atomicAdd(&bs.counters[11 + dbg_state], 1u); if (frame == 0) atomicAdd(&bs.counters[14], 1u); }
        if (threadIdx.x == 0) { atomicAdd(&bs.counters[8], T); atomicMax(&bs.counters[9], T); }
        { const unsigned long long md = __ballot(kind == 1); if (lane == 0) atomicAdd(&bs.counters[10], (uint32_t)__popcll(md)); }
#endif
        STAMP(st_prefix)
        // ---- flattened bin-by-bin tests: thread t of a round takes tests t, t + WG, ... ----
        constexpr int TPT = BIN_TESTS_PER_THREAD;
        for (uint32_t r0 = 0; r0 < T; r0 += WG * TPT) {
#ifdef MIRT_BIN_STAMPS
            st_nrounds++;
#endif
            __syncthreads();                          // s_pre written / the previous round's appends counted
            if (s_fill + (uint32_t)(WG * TPT) > (uint32_t)BIN_PAIR_BUF) flush();
            bool pass[TPT];
            uint32_t key[TPT], val[TPT];
#pragma unroll
            for (int q = 0; q < TPT; q++) {
                const uint32_t t = r0 + (uint32_t)(q * WG) + threadIdx.x;
                pass[q] = false; key[q] = 0; val[q] = 0;
                if (t < T) {
                    uint32_t lo = 0, hi = 256;            // the item whose range [s_pre[i], s_pre[i+1]) holds t
#pragma unroll
                    for (int step = 0; step < 8; step++) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (s_pre[mid] <= t) lo = mid; else hi = mid;
                    }
                    const uint32_t nis = s_ni[lo], wi = nis & 0xFFu, sh = nis >> 8;
                    const uint32_t b = t - s_pre[lo], org = s_org[lo];
                    const uint32_t recip = 65536u / wi + 1u;             // b / wi == (b * recip) >> 16 for b < 1024, wi <= 32
                    const uint32_t dj = (b * recip) >> 16, di = b - dj * wi;
                    const int i = (int)(org & 0xFFFFu) + (int)di, j = (int)(org >> 16) + (int)dj;
                    const float4 A2 = s_A2[lo], Bu = s_Bu[lo], Bv = s_Bv[lo], box = s_box[lo];
                    const float FI = (float)i, FJ = (float)j;
                    pass[q] = (__builtin_fmaf(FJ, Bv.x, __builtin_fmaf(FI, Bu.x, A2.x)) >= 0.0f) &&
                              (__builtin_fmaf(FJ, Bv.y, __builtin_fmaf(FI, Bu.y, A2.y)) >= 0.0f) &&
                              (__builtin_fmaf(FJ, Bv.z, __builtin_fmaf(FI, Bu.z, A2.z)) >= 0.0f) &&
                              (__builtin_fmaf(FJ, Bv.w, __builtin_fmaf(FI, Bu.w, A2.w)) >= 0.0f) &&
                              (FI + 1.0f >= box.x) && (FI <= box.y) && (FJ + 1.0f >= box.z) && (FJ <= box.w);   // = cell_may_hit(.., A2, i, j, 1)
                    key[q] = (gr.fbase + (uint32_t)j * (uint32_t)nbu + (uint32_t)i) * gr.nshell + sh;
                    val[q] = s_tri[lo];
                }
            }
            // one LDS atomic per wave and round hands out the slots of all its passing tests
            unsigned long long m[TPT];
            uint32_t off[TPT], cnt = 0;
#pragma unroll
            for (int q = 0; q < TPT; q++) { m[q] = __ballot(pass[q]); off[q] = cnt; cnt += (uint32_t)__popcll(m[q]); }
            if (cnt) {
                uint32_t at0 = 0;
                if (lane == 0) at0 = atomicAdd(&s_fill, cnt);
                at0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)at0);
#pragma unroll
                for (int q = 0; q < TPT; q++)
                    if (pass[q]) {
                        const uint32_t at = at0 + off[q] + (uint32_t)__popcll(m[q] & ((1ull << lane) - 1ull));
                        s_keys[at] = key[q]; s_vals[at] = val[q];         // fits: the round started with room for WG * TPT
                    }
            }
        }

        STAMP(st_rounds)
        // ---- huge items: the wave walks them one at a time, pairs into the same LDS buffer ----
        __syncthreads();                              // no flattened round (which counts on its WG * TPT free slots) is still appending
        {
            BinLargeSink sink = { s_keys, s_vals, &s_fill, &s_valid, &bs.counters[0], out, bs.bucket_cnt, bs.bucket_shift };
            for (unsigned long long ml = __ballot(kind == 2); ml;) {
                const int src = __builtin_ctzll(ml);
                ml &= ml - 1ull;
                BinItem u;                            // the item, made wave-uniform
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    u.A0[k] = bcastf(it.A0[k], src); u.A1[k] = bcastf(it.A1[k], src); u.A2[k] = bcastf(it.A2[k], src);
                    u.Bu[k] = bcastf(it.Bu[k], src); u.Bv[k] = bcastf(it.Bv[k], src);
                }
                u.lou = bcastf(it.lou, src); u.hiu = bcastf(it.hiu, src); u.lov = bcastf(it.lov, src); u.hiv = bcastf(it.hiv, src);
                const uint32_t utri = (uint32_t)__builtin_amdgcn_readlane((int)tri, src);
                const uint32_t ushell = (uint32_t)__builtin_amdgcn_readlane((int)shell, src);
                const unsigned long long ucells = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(cells >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)cells, src);
                bin_walk_large(u, utri, ushell, gr, lane, sink, ucells);
            }
        }
        STAMP(st_huge)
    }
#ifdef MIRT_BIN_STAMPS
    st_t = __builtin_amdgcn_s_memtime();
#endif
    flush();
    STAMP(st_flush)
#ifdef MIRT_BIN_STAMPS
    if (threadIdx.x == 0 && (blockIdx.x % 97) == 0)
        printf("wg %d: total %lld | setup %lld prefix %lld rounds %lld (%d) huge %lld flush %lld\n", (int)blockIdx.x, (long long)__builtin_amdgcn_s_memtime() - st_begin,
               st_setup, st_prefix, st_rounds, st_nrounds, st_huge, st_flush);
#endif
}

// Two shapes of workgroup: 512 threads -- a work item's 256 triangles are set up by the first four waves, the flattened tests run on
// all eight -- and 256, where every wave sets up and tests.  The smaller one holds no idle waves during the set-up, which leaves the
// frames in flight beside it more of the chip (100 k triangles at 1080p: 63.8 against 65.7 us per frame with four in flight, 122
// against 118 us with one); the larger one halves the rounds of the tests, which is what a 1 M-triangle frame wants (1.15 against
// 1.23 ms at 8K).  mirt_capi.hip picks by the size of the scene and the frames in flight (MIRT_BIN_WG overrides).
template __global__ void k_bin_pairs<512>(const float *, const OriginRow *, const OriginRow *, int, BinSet, BinPairs);
template __global__ void k_bin_pairs<256>(const float *, const OriginRow *, const OriginRow *, int, BinSet, BinPairs);

}  // namespace mirt
