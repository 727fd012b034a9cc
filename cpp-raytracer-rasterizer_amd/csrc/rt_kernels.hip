// rt_kernels.hip -- hand-written gfx950 kernels for the ray tracer's Draw() loop
// (raytracer/Source/raytracer.cpp:547-606 -> ClosestIntersection :202-257 -> DirectLight :265-327 ->
//  CalculateDOF/PutPixelSDL :608-656, SDLauxiliary.h:70-81).
//
// One thread owns one pixel (P pixels when P > 1), the triangle list is staged into LDS once per workgroup
// chunk and every lane walks it with wave-uniform (broadcast) LDS reads; results leave through coalesced
// 32-bit framebuffer stores.  No MFMA: this is not a contraction.  Built with -ffp-contract=off.
#include "rt_common.hpp"

#include <float.h>

namespace mirt {

// ---- k_prep_origin: origin tables for the camera and every light --------------------------------------
// grid.y = number of origins to build starting at origin o0 (0 = camera, 1 + k = light position k), one thread per
// triangle.  Also raises *unsafe when an entry is outside the pre-reject filter's proven range (then the trace kernels
// skip the filter), and -- being the first kernel of a frame -- zeroes the frame's hit counters and the binning pass's
// pair counter (both nullable), which saves a memset node each.
__global__ __launch_bounds__(256) void k_prep_origin(const float *__restrict__ tris15, int n,
                                                     const float *__restrict__ origins /* (1+nl) x 3, or NULL: origin0 */, v3 origin0, int o0,
                                                     OriginRow *__restrict__ cam_tab,
                                                     OriginRow *__restrict__ light_tab,
                                                     uint32_t *__restrict__ unsafe,
                                                     unsigned long long *__restrict__ zero_hits, uint32_t *__restrict__ zero_counter)
{
    if (blockIdx.y == 0 && blockIdx.x == 0) {
        if (zero_hits)
            for (int g = threadIdx.x; g < HIT_SHARDS * HIT_SHARD_STRIDE; g += blockDim.x) zero_hits[g] = 0ull;
        // [0] pairs of the binning pass, [16..79] tile pairs per (XCD group, list-length class) (k_tile_order); the words between:
        // debug statistics
        if (zero_counter && (threadIdx.x == 0 || (threadIdx.x >= 16 && threadIdx.x < 80))) zero_counter[threadIdx.x] = 0u;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int o = o0 + (int)blockIdx.y;
    if (i >= n) return;
    const v3 S = origins ? ld3(origins + 3 * o) : origin0;
    const OriginRow r = make_origin_row(tris15 + (size_t)15 * i, S);
    OriginRow *dst = (o == 0) ? cam_tab : light_tab + (size_t)(o - 1) * n;
    dst[i] = r;
    if (unsafe && !origin_row_safe(r)) atomicOr(unsafe, 1u);
}

// ---- k_rt_brute: fused primary + shadow + shade + resolve, every ray tests every triangle ------------
//
// Workgroup = 256 threads = 4 wave64; wave w of block (bx, by) owns row y0 + 4*by + w, pixels
// x = bx*64*P + p*64 + lane, so each framebuffer store instruction writes 64 consecutive words (256 B).
// The origin table is staged through LDS in chunks of CHUNK rows (48 B each); all 64 lanes read the same
// row => three conflict-free ds_read_b128 broadcasts per triangle, reused for the lane's P rays.
constexpr int RT_CHUNK = RT_CHUNK_ROWS;

// (bx, by) = the block of P*64 x 4 pixels; CHUNK = origin rows staged in LDS at a time
template <int P, bool FILTER, bool AA, int CHUNK = RT_CHUNK>
__device__ __forceinline__ void brute_body(const RtFrame &f, float4 *s_tab, int bx, int by)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int y = f.y0 + by * 4 + wave;
    const bool row_ok = y < f.y1;
    const v3 cam = ld3(f.cam);
    const float halfW = (float)f.W / 2.0f, halfH = (float)f.H / 2.0f;
    const int rs = AA ? f.aa : 1;                         // realSamples (:549-554); compile-time 1 without supersampling

    int xs[P];
    bool ok[P];
    v3 pos[P], avg[P];
    float best_d[P], x1[P];
    int best_i[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        xs[p] = (bx * P + p) * 64 + lane;
        ok[p] = row_ok && xs[p] < f.W;
        best_d[p] = FLT_MAX;                              // Update() reset (:335-339), once per frame
        best_i[p] = -1;
        pos[p] = avg[p] = V3(0.0f, 0.0f, 0.0f);
    }

    float y1 = aa_start(y, rs);                           // :566-569
    for (int z = 0; z < rs; z++) {
#pragma unroll
        for (int p = 0; p < P; p++) x1[p] = aa_start(xs[p], rs);          // :573-576
        for (int z2 = 0; z2 < rs; z2++) {
            RayDirs<P> nd;
            bool any[P];
#pragma unroll
            for (int p = 0; p < P; p++) {
                // d = (x1 - W/2, y1 - H/2, focalLength); dir = cameraRot * d   (raytracer.cpp:579-580)
                const v3 d = V3(x1[p] - halfW, y1 - halfH, f.focal);
                nd.set(p, neg3(mat3_mul_vec(f.rot, d)));  // negD = -dir (:229)
                any[p] = false;
            }

            // ---------------- primary sub-ray: closest hit so far, ties -> later index (:243) ----------------
            for (int base = 0; base < f.n; base += CHUNK) {
                const int cnt = min(CHUNK, f.n - base);
                __syncthreads();
                {
                    const float4 *src = reinterpret_cast<const float4 *>(f.cam_tab + base);
                    for (int k = threadIdx.x; k < cnt * 3; k += 256) s_tab[k] = src[k];
                }
                __syncthreads();
                // (an explicit one-row-ahead software pipeline costs 4 extra register moves per test here and the
                // other resident waves already hide the LDS latency: measured 205 ms vs 239 ms on the 100k soup)
#pragma unroll 2
                for (int j = 0; j < cnt; j++) {
                    const float4 r0 = s_tab[3 * j], r1 = s_tab[3 * j + 1], r2 = s_tab[3 * j + 2];
                    TestDots d[P];
                    bool maybe[P];
                    test_rays<P, FILTER>(r0, r1, r2, nd, d, maybe);
#pragma unroll
                    for (int p = 0; p < P; p++) {
                        if (maybe[p]) {
                            v3 hp;
                            float dist;
                            if (exact_hit(d[p], r0.w, f.tris15 + (size_t)15 * (base + j), cam, &hp, &dist)) {
                                any[p] = true;
                                if (best_d[p] >= dist) { best_d[p] = dist; best_i[p] = base + j; pos[p] = hp; }
                            }
                        }
                    }
                }
            }

            // ---------------- DirectLight: one shadow ray per light position per sub-ray that hit ----------------
            v3 result[P], result2[P], nDir[P], tcol[P];
            bool hit[P];
#pragma unroll
            for (int p = 0; p < P; p++) {
                hit[p] = ok[p] && any[p];
                result[p] = result2[p] = V3(0.0f, 0.0f, 0.0f);
                const float *t = f.tris15 + (size_t)15 * (best_i[p] >= 0 ? best_i[p] : 0);
                nDir[p] = normalize3(ld3(t + 9));         // glm::normalize(triangles[idx].normal) (:300)
                tcol[p] = ld3(t + 12);
            }
            {
                unsigned long long m = 0;
#pragma unroll
                for (int p = 0; p < P; p++) m += __popcll(__ballot(hit[p]));
                count_hits(f, m);
            }

            for (int k = 0; k < f.nlights; k++) {
                const v3 L = ld3(f.lpos[k]);
                v3 D[P];
                RayDirs<P> rd;
                float thr[P];
                bool live[P];      // still needs shadow testing
                bool any_live = false;
#pragma unroll
                for (int p = 0; p < P; p++) {
                    float r;
                    v3 rdp;
                    D[p] = light_term(f, k, pos[p], nDir[p], &rdp, &r);
                    rd.set(p, rdp);
                    thr[p] = r * 0.99f;                    // j.distance < r*0.99f (:313)
                    live[p] = hit[p];
                    any_live |= live[p];
                }
                const OriginRow *tab = f.light_tab + (size_t)k * f.n;
                // every wave of the block must take part in the staging barriers, so the chunk loop is
                // unconditional; a wave with nothing left to test just skips the inner loop.
                for (int base = 0; base < f.n; base += CHUNK) {
                    const int cnt = min(CHUNK, f.n - base);
                    __syncthreads();
                    {
                        const float4 *src = reinterpret_cast<const float4 *>(tab + base);
                        for (int q = threadIdx.x; q < cnt * 3; q += 256) s_tab[q] = src[q];
                    }
                    __syncthreads();
                    if (!__any(any_live)) continue;
#pragma unroll 2
                    for (int j = 0; j < cnt; j++) {
                        const float4 r0 = s_tab[3 * j], r1 = s_tab[3 * j + 1], r2 = s_tab[3 * j + 2];
                        // shadow ray: start = light, dir = -rDir, so negD = rDir (:310, :229)
                        TestDots d[P];
                        bool maybe[P];
                        test_rays<P, FILTER>(r0, r1, r2, rd, d, maybe);
#pragma unroll
                        for (int p = 0; p < P; p++) {
                            if (live[p] && maybe[p]) {
                                v3 hp;
                                float dist;
                                if (exact_hit(d[p], r0.w, f.tris15 + (size_t)15 * (base + j), L, &hp, &dist)) {
                                    // min over accepted hits < thr  <=>  some accepted hit < thr (any-hit is exact)
                                    if (dist < thr[p]) live[p] = false, D[p] = V3(0.0f, 0.0f, 0.0f);
                                }
                            }
                        }
                    }
                    any_live = false;
#pragma unroll
                    for (int p = 0; p < P; p++) any_live |= live[p];
                }
#pragma unroll
                for (int p = 0; p < P; p++) {
                    result[p] = add3(result[p], D[p]);             // result += D   (:319)
                    if ((k + 1) % f.samples == 0) result2[p] = add3(result2[p], result[p]);   // after each light's samples (:322)
                }
            }

            const v3 N = ld3(f.indirect);
#pragma unroll
            for (int p = 0; p < P; p++) {
                if (hit[p]) {
                    const v3 Dl = mul3(result2[p], tcol[p]);       // DirectLight returns result2 * color (:325-326)
                    const v3 T = add3(Dl, N);                      // (:584-586)
                    avg[p] = add3(avg[p], mul3(tcol[p], T));       // (:587-591)
                    x1[p] += aa_step(rs);                          // (:593) only after a hit
                }
            }
        }
        y1 += aa_step(rs);                                         // (:596)
    }

    // ---------------- resolve ----------------
#pragma unroll
    for (int p = 0; p < P; p++) {
        if (!ok[p]) continue;
        const v3 out = div3s(avg[p], (float)(rs * rs));            // /= realSamples^2 (:599)
        const int x = xs[p];
        const size_t px = (size_t)y * f.W + x;
        if (f.rgb) st3(f.rgb + 3 * px, out);
        if (f.index) f.index[px] = best_i[p];
        if (f.fd) f.fd[px] = best_i[p] >= 0 ? best_d[p] - f.focal_plane : 0.0f;   // focalDistances (:248-249)
        store_intersection(f, px, best_i[p], best_d[p], pos[p]);
        // CalculateDOF draws interior pixels only (:618-620); the border keeps its old value
        if (x >= 1 && x < f.W - 1 && y >= 1 && y < f.H - 1)
            f.xrgb[(size_t)(y - f.row_origin) * f.pitch_words + x] = pack_xrgb(out);
    }
}

// ---- k_rt_small: the whole scene lives in LDS -----------------------------------------------------------
//
// For scenes like the reference's own Cornell box (30 triangles) one workgroup builds every table it needs
// itself -- the origin rows for the camera and each light and the {v0, e1, e2} the exact path reads -- so the
// frame is ONE kernel launch with no global loads inside the loops and a single barrier; waves that miss
// everything retire early.  LDS: n * 48 B * (2 + nlights).
struct SmallGeo { float4 g0, g1, g2; };   // {v0.xyz, e1.x}, {e1.y, e1.z, e2.x, e2.y}, {e2.z, -, -, -}

__device__ __forceinline__ bool exact_hit_lds(const TestDots &d, float e1e2b, const float4 *geo, v3 start, v3 *pos, float *dist)
{
    const float t = e1e2b / d.den, u = d.pu / d.den, v = d.qv / d.den;      // raytracer.cpp:237
    if (u + v <= 1.0f && u >= 0.0f && v >= 0.0f && t >= 0.0f) {             // :239
        const float4 g0 = geo[0], g1 = geo[1], g2 = geo[2];
        const v3 v0 = V3(g0.x, g0.y, g0.z), e1 = V3(g0.w, g1.x, g1.y), e2 = V3(g1.z, g1.w, g2.x);
        const v3 p = add3(add3(v0, scale3(e1, u)), scale3(e2, v));          // :241
        *pos = p;
        *dist = distance3(start, p);                                         // :242
        return true;
    }
    return false;
}

template <int P, bool FILTER, bool AA>
__device__ __forceinline__ void small_body(const RtFrame &f, const float4 *s_cam, const float4 *s_light, const float4 *s_geo)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int y = f.y0 + (int)blockIdx.y * 4 + wave;
    if (y >= f.y1) return;
    const v3 cam = ld3(f.cam);
    const float halfW = (float)f.W / 2.0f, halfH = (float)f.H / 2.0f;
    const int n = f.n;
    const int rs = AA ? f.aa : 1;                                            // realSamples (:549-554)

    int xs[P];
    bool ok[P];
    v3 pos[P], avg[P];
    float best_d[P], x1[P];
    int best_i[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        xs[p] = ((int)blockIdx.x * P + p) * 64 + lane;
        ok[p] = xs[p] < f.W;
        best_d[p] = FLT_MAX;                                                 // Update() reset, once per frame
        best_i[p] = -1;
        pos[p] = avg[p] = V3(0.0f, 0.0f, 0.0f);
    }

    float y1 = aa_start(y, rs);                                              // :566-569
    for (int z = 0; z < rs; z++) {
#pragma unroll
        for (int p = 0; p < P; p++) x1[p] = aa_start(xs[p], rs);            // :573-576
        for (int z2 = 0; z2 < rs; z2++) {
            RayDirs<P> nd;
            bool any[P];
#pragma unroll
            for (int p = 0; p < P; p++) {
                const v3 d = V3(x1[p] - halfW, y1 - halfH, f.focal);        // raytracer.cpp:579
                nd.set(p, neg3(mat3_mul_vec(f.rot, d)));                     // :580, :229
                any[p] = false;
            }
#pragma unroll 2
            for (int j = 0; j < n; j++) {
                const float4 r0 = s_cam[3 * j], r1 = s_cam[3 * j + 1], r2 = s_cam[3 * j + 2];
                TestDots d[P];
                bool maybe[P];
                test_rays<P, FILTER>(r0, r1, r2, nd, d, maybe);
#pragma unroll
                for (int p = 0; p < P; p++) {
                    if (maybe[p]) {
                        v3 hp;
                        float dist;
                        if (exact_hit_lds(d[p], r0.w, s_geo + 3 * j, cam, &hp, &dist)) {
                            any[p] = true;
                            if (best_d[p] >= dist) { best_d[p] = dist; best_i[p] = j; pos[p] = hp; }   // :243-247
                        }
                    }
                }
            }

            bool hit[P];
            bool any_hit = false;
#pragma unroll
            for (int p = 0; p < P; p++) { hit[p] = ok[p] && any[p]; any_hit |= hit[p]; }
            {
                unsigned long long m = 0;
#pragma unroll
                for (int p = 0; p < P; p++) m += __popcll(__ballot(hit[p]));
                count_hits(f, m);
            }
            if (!__any(any_hit)) continue;

            v3 result[P], result2[P], nDir[P], tcol[P];
#pragma unroll
            for (int p = 0; p < P; p++) {
                result[p] = result2[p] = V3(0.0f, 0.0f, 0.0f);
                const float *t = f.tris15 + (size_t)15 * (best_i[p] >= 0 ? best_i[p] : 0);
                nDir[p] = normalize3(ld3(t + 9));                            // :300
                tcol[p] = ld3(t + 12);
            }
            for (int k = 0; k < f.nlights; k++) {
                const v3 L = ld3(f.lpos[k]);
                const float4 *tab = s_light + (size_t)3 * n * k;
                v3 D[P];
                RayDirs<P> rd;
                float thr[P];
                bool live[P];
#pragma unroll
                for (int p = 0; p < P; p++) {
                    float r;
                    v3 rdp;
                    D[p] = light_term(f, k, pos[p], nDir[p], &rdp, &r);
                    rd.set(p, rdp);
                    thr[p] = r * 0.99f;                                      // :313
                    live[p] = hit[p];
                }
#pragma unroll 2
                for (int j = 0; j < n; j++) {
                    const float4 r0 = tab[3 * j], r1 = tab[3 * j + 1], r2 = tab[3 * j + 2];
                    TestDots d[P];
                    bool maybe[P];
                    test_rays<P, FILTER>(r0, r1, r2, rd, d, maybe);          // negD = rDir (:310, :229)
#pragma unroll
                    for (int p = 0; p < P; p++) {
                        if (live[p] && maybe[p]) {
                            v3 hp;
                            float dist;
                            if (exact_hit_lds(d[p], r0.w, s_geo + 3 * j, L, &hp, &dist))
                                if (dist < thr[p]) live[p] = false, D[p] = V3(0.0f, 0.0f, 0.0f);   // :313-314
                        }
                    }
                }
#pragma unroll
                for (int p = 0; p < P; p++) {
                    result[p] = add3(result[p], D[p]);                       // :319
                    if ((k + 1) % f.samples == 0) result2[p] = add3(result2[p], result[p]);    // :322
                }
            }
            const v3 N = ld3(f.indirect);
#pragma unroll
            for (int p = 0; p < P; p++) {
                if (hit[p]) {
                    const v3 Dl = mul3(result2[p], tcol[p]);                 // :325-326
                    avg[p] = add3(avg[p], mul3(tcol[p], add3(Dl, N)));       // :584-591
                    x1[p] += aa_step(rs);                                    // :593, only after a hit
                }
            }
        }
        y1 += aa_step(rs);                                                   // :596
    }

#pragma unroll
    for (int p = 0; p < P; p++) {
        if (!ok[p]) continue;
        const v3 out = div3s(avg[p], (float)(rs * rs));                      // :599
        const int x = xs[p];
        const size_t px = (size_t)y * f.W + x;
        if (f.rgb) st3(f.rgb + 3 * px, out);
        if (f.index) f.index[px] = best_i[p];
        if (f.fd) f.fd[px] = best_i[p] >= 0 ? best_d[p] - f.focal_plane : 0.0f;   // focalDistances (:248-249)
        store_intersection(f, px, best_i[p], best_d[p], pos[p]);
        if (x >= 1 && x < f.W - 1 && y >= 1 && y < f.H - 1)                  // :618-620
            f.xrgb[(size_t)(y - f.row_origin) * f.pitch_words + x] = pack_xrgb(out);
    }
}

// `host_unsafe` != 0: operands outside the filter's proven range (decided on the host) -> exact-only path.
template <int P>
__global__ __launch_bounds__(256) void k_rt_small(const RtFrame f, int host_unsafe)
{
    extern __shared__ __attribute__((aligned(16))) float4 s_all[];
    const int n = f.n;
    // all LDS comes from the dynamic region (a static __shared__ in front would knock the b128 reads off
    // their 16-byte alignment): [0] = flag word, then the tables
    int *s_unsafe = reinterpret_cast<int *>(s_all);
    float4 *s_cam = s_all + 1;                   // n rows
    float4 *s_geo = s_cam + 3 * n;               // n rows
    float4 *s_light = s_cam + 6 * n;             // nlights x n rows
    if (threadIdx.x == 0) *s_unsafe = host_unsafe;
    __syncthreads();
    bool bad = false;
    for (int i = threadIdx.x; i < n * (2 + f.nlights); i += 256) {
        const int which = i / n, t = i - which * n;
        const float *t15 = f.tris15 + (size_t)15 * t;
        if (which == 1) {
            const v3 v0 = ld3(t15), e1 = sub3(ld3(t15 + 3), v0), e2 = sub3(ld3(t15 + 6), v0);   // :216-217
            s_geo[3 * t] = make_float4(v0.x, v0.y, v0.z, e1.x);
            s_geo[3 * t + 1] = make_float4(e1.y, e1.z, e2.x, e2.y);
            s_geo[3 * t + 2] = make_float4(e2.z, 0.0f, 0.0f, 0.0f);
        } else {
            const v3 S = (which == 0) ? ld3(f.cam) : ld3(f.lpos[which - 2]);
            const OriginRow r = make_origin_row(t15, S);
            float4 *dst = (which == 0) ? s_cam + 3 * t : s_light + 3 * ((size_t)(which - 2) * n + t);
            dst[0] = r.r0; dst[1] = r.r1; dst[2] = r.r2;
            bad |= !origin_row_safe(r);
        }
    }
    if (bad) atomicOr(s_unsafe, 1);
    __syncthreads();
    if (f.aa > 1) {
        if (*s_unsafe == 0) small_body<P, true, true>(f, s_cam, s_light, s_geo);
        else small_body<P, false, true>(f, s_cam, s_light, s_geo);
    } else {
        if (*s_unsafe == 0) small_body<P, true, false>(f, s_cam, s_light, s_geo);
        else small_body<P, false, false>(f, s_cam, s_light, s_geo);
    }
}

template __global__ void k_rt_small<2>(const RtFrame, int);

// ---- k_rt_wave: one WAVE per ray, lanes over triangles, wavefront min-t reduce ----------------------------
//
// For frames with few rays and many triangles (a 32x32 pick query into a 100k-triangle scene) one thread per pixel
// leaves the chip empty.  Here a wave owns one pixel: its 64 lanes stride over the origin table (coalesced 48-byte
// rows straight from global memory), each keeps its own closest hit, and the wave reduces them with the packed
// min-t key (rt_common.hpp: wave_min_key).  Shadow rays are the same sweep with an any-hit ballot and early exit.
// Same filter + exact arithmetic as every other kernel, so results are bit-identical.
template <bool FILTER>
__device__ __forceinline__ void wave_body(const RtFrame &f)
{
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int rows = f.y1 - f.y0;
    if (ray >= (long long)f.W * rows) return;
    const int x = (int)(ray % f.W), y = f.y0 + (int)(ray / f.W);
    const v3 cam = ld3(f.cam);
    const v3 d = V3((float)x - (float)f.W / 2.0f, (float)y - (float)f.H / 2.0f, f.focal);   // raytracer.cpp:579
    const v3 nd = neg3(mat3_mul_vec(f.rot, d));                                                // :580, :229

    unsigned long long key = MIN_T_NONE;
    v3 pos = V3(0.0f, 0.0f, 0.0f);
    for (int i = lane; i < f.n; i += 64) {
        const float4 *src = reinterpret_cast<const float4 *>(f.cam_tab + i);
        const float4 r0 = src[0], r1 = src[1], r2 = src[2];
        const TestDots td = test_dots(r0, r1, r2, nd);
        if (!FILTER || maybe_hit(td)) {
            v3 hp;
            float dist;
            if (exact_hit(td, r0.w, f.tris15 + (size_t)15 * i, cam, &hp, &dist)) {
                const unsigned long long k = min_t_key(dist, i);
                if (k < key) { key = k; pos = hp; }            // lane-local: min distance, then max index
            }
        }
    }
    const unsigned long long best = wave_min_key(key);
    const bool hit = best != MIN_T_NONE;
    const int best_i = hit ? min_t_index(best) : -1;
    // the lane that holds the winner broadcasts its hit point
    const int owner = __builtin_ctzll(__ballot(key == best) | (1ull << 63));
    pos.x = __shfl(pos.x, owner); pos.y = __shfl(pos.y, owner); pos.z = __shfl(pos.z, owner);
    if (lane == 0 && hit) count_hits(f, 1);

    v3 avg = V3(0.0f, 0.0f, 0.0f);
    if (hit) {
        const float *t = f.tris15 + (size_t)15 * best_i;
        const v3 nDir = normalize3(ld3(t + 9));                       // :300
        const v3 tcol = ld3(t + 12);
        v3 result = V3(0.0f, 0.0f, 0.0f), result2 = V3(0.0f, 0.0f, 0.0f);
        for (int k = 0; k < f.nlights; k++) {
            const v3 L = ld3(f.lpos[k]);
            v3 rd;
            float r;
            v3 D = light_term(f, k, pos, nDir, &rd, &r);
            const float thr = r * 0.99f;                              // :313
            const OriginRow *tab = f.light_tab + (size_t)k * f.n;
            bool occluded = false;
            for (int base = 0; base < f.n && !occluded; base += 64) {
                const int i = base + lane;
                bool occ = false;
                if (i < f.n) {
                    const float4 *src = reinterpret_cast<const float4 *>(tab + i);
                    const float4 r0 = src[0], r1 = src[1], r2 = src[2];
                    const TestDots td = test_dots(r0, r1, r2, rd);    // negD = rDir (:310, :229)
                    if (!FILTER || maybe_hit(td)) {
                        v3 hp;
                        float dist;
                        occ = exact_hit(td, r0.w, f.tris15 + (size_t)15 * i, L, &hp, &dist) && dist < thr;
                    }
                }
                occluded = __any(occ);                                // any-hit is exact (SURVEY A-5)
            }
            if (occluded) D = V3(0.0f, 0.0f, 0.0f);                   // :313-314
            result = add3(result, D);                                 // :319
            if ((k + 1) % f.samples == 0) result2 = add3(result2, result);                 // :322
        }
        const v3 Dl = mul3(result2, tcol);                            // :325-326
        avg = add3(avg, mul3(tcol, add3(Dl, ld3(f.indirect))));       // :584-591
    }
    avg = div3s(avg, 1.0f);                                           // :599
    if (lane != 0) return;
    const size_t px = (size_t)y * f.W + x;
    if (f.rgb) st3(f.rgb + 3 * px, avg);
    if (f.index) f.index[px] = best_i;
    if (f.fd) f.fd[px] = hit ? min_t_dist(best) - f.focal_plane : 0.0f;   // focalDistances (:248-249)
    store_intersection(f, px, best_i, min_t_dist(best), pos);
    if (x >= 1 && x < f.W - 1 && y >= 1 && y < f.H - 1)               // :618-620
        f.xrgb[(size_t)(y - f.row_origin) * f.pitch_words + x] = pack_xrgb(avg);
}

__global__ __launch_bounds__(256) void k_rt_wave(const RtFrame f)
{
    if (__builtin_amdgcn_readfirstlane(*f.unsafe) == 0u)
        wave_body<true>(f);
    else
        wave_body<false>(f);
}

template <int P>
__global__ __launch_bounds__(256) void k_rt_brute(const RtFrame f)
{
    extern __shared__ __attribute__((aligned(16))) float4 s_tab[];
    const bool safe = __builtin_amdgcn_readfirstlane(*f.unsafe) == 0u;
    if (f.aa > 1) {
        if (safe) brute_body<P, true, true>(f, s_tab, (int)blockIdx.x, (int)blockIdx.y);
        else brute_body<P, false, true>(f, s_tab, (int)blockIdx.x, (int)blockIdx.y);
    } else {
        if (safe) brute_body<P, true, false>(f, s_tab, (int)blockIdx.x, (int)blockIdx.y);
        else brute_body<P, false, false>(f, s_tab, (int)blockIdx.x, (int)blockIdx.y);
    }
}

template __global__ void k_rt_brute<2>(const RtFrame);

}  // namespace mirt
