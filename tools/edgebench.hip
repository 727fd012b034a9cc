#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../cpp-raytracer-rasterizer_amd/csrc/raster_common.hpp"      // edge_advance
__global__ __launch_bounds__(256) void k_chain(float *out, long long *cyc, int n, int dstep, int mode)
{
    extern __shared__ float s[];
    const int tid = threadIdx.x;
    long long t0 = __builtin_amdgcn_s_memtime();
    if (tid < 12) {
        float cur = (float)tid, step = 1.0f / 3.0f;
        float *dst = s + (dstep > 0 ? 0 : (n - 1) * 15) + tid;
        if (mode == 0) {
#pragma unroll 8
            for (int k = 0; k < n; k++) { *dst = cur; cur += step; dst += dstep; }
        } else if (mode == 1) {
#pragma unroll 8
            for (int k = 0; k < n; k++) { dst[k * 15] = cur; cur += step; }
        } else if (mode == 2) {
#pragma unroll 8
            for (int k = 0; k < n; k++) { cur += step; }
            *dst = cur;
        } else if (mode == 3) {
            // field-major layout: 4 consecutive steps of one chain are 16 contiguous bytes
            float4 *d4 = reinterpret_cast<float4 *>(s + tid * n);
#pragma unroll 2
            for (int k = 0; k < n; k += 4) {
                float4 v; v.x = cur; cur += step; v.y = cur; cur += step; v.z = cur; cur += step; v.w = cur; cur += step;
                d4[k >> 2] = v;
            }
        } else if (mode == 4) {
            float2 *d2 = reinterpret_cast<float2 *>(s + tid * n);
#pragma unroll 4
            for (int k = 0; k < n; k += 2) {
                float2 v; v.x = cur; cur += step; v.y = cur; cur += step;
                d2[k >> 1] = v;
            }
        } else if (mode == 6 || mode == 7) {
            float4 *d4 = reinterpret_cast<float4 *>(s + tid * n);
            if (mode == 6) {
                for (int k = 0; k < n; k += 16) {
                    float4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) { v[q].x = cur; cur += step; v[q].y = cur; cur += step; v[q].z = cur; cur += step; v[q].w = cur; cur += step; }
#pragma unroll
                    for (int q = 0; q < 4; q++) d4[(k >> 2) + q] = v[q];
                }
            } else {
                for (int k = 0; k < n; k += 32) {
                    float4 v[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) { v[q].x = cur; cur += step; v[q].y = cur; cur += step; v[q].z = cur; cur += step; v[q].w = cur; cur += step; }
#pragma unroll
                    for (int q = 0; q < 8; q++) d4[(k >> 2) + q] = v[q];
                }
            }
        } else if (mode == 5) {
            // two waves' worth of trick: keep 8 values, write two b128
            float4 *d4 = reinterpret_cast<float4 *>(s + tid * n);
            for (int k = 0; k < n; k += 8) {
                float4 v, w; v.x = cur; cur += step; v.y = cur; cur += step; v.z = cur; cur += step; v.w = cur; cur += step;
                w.x = cur; cur += step; w.y = cur; cur += step; w.z = cur; cur += step; w.w = cur; cur += step;
                d4[k >> 2] = v; d4[(k >> 2) + 1] = w;
            }
        }
    }
    if (mode == 8 || mode == 9) {
        // (measured: 21.9 ticks per step against 12.2 for the lone wave of mode 7 -- four chain-walking waves slow each other down)
        // every wave of the workgroup (one per SIMD) walks the WHOLE chain and stores only its share of the groups:
        // the adds are redundant work on SIMDs that idle anyway, the stores (two issue slots each) are split four ways
        const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        if (lane < 12) {
            float cur = (float)lane, step = 1.0f / 3.0f;
            float4 *d4 = reinterpret_cast<float4 *>(s + lane * n);
            const int per = mode == 8 ? 4 : 8;                   // groups of 4 steps per loop iteration; wave wv stores groups with (g % 4) == wv
            for (int k = 0; k < n; k += 4 * per) {
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    if (q >= per) break;
                    float4 v; v.x = cur; cur += step; v.y = cur; cur += step; v.z = cur; cur += step; v.w = cur; cur += step;
                    if ((q & 3) == wv) d4[(k >> 2) + q] = v;
                }
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    if (tid < 12) out[blockIdx.x * 16 + tid] = s[tid + 15 * (n / 2)] + s[tid * n + 7];
}
// edge_advance(cur, step, n) against n additions, one thread per random (cur, step, n): how often the prediction is the sum itself.
// (k_raster_edges_lds checks every prediction it uses, so a miss costs time, not pixels; this says how rare misses are.)
__device__ unsigned lcg(unsigned &x) { x = x * 1664525u + 1013904223u; return x; }
__global__ void k_advance_check(unsigned seed, int cases_per_thread, unsigned long long *tally /* [0] cases, [1] misses, [2..] first missed case */)
{
    unsigned x = seed ^ (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
    for (int c = 0; c < cases_per_thread; c++) {
        const unsigned kind = lcg(x) % 5u;
        float cur, step;
        const int n = (int)(lcg(x) % 4400u);
        const float u1 = (float)(lcg(x) >> 8) * (1.0f / 16777216.0f), u2 = (float)(lcg(x) >> 8) * (1.0f / 16777216.0f);
        if (kind == 0) { cur = (float)(int)(u1 * 8000.0f - 2000.0f); step = (float)(int)(u2 * 8000.0f - 4000.0f) / (float)max(n, 1); }          // x: integer ends
        else if (kind == 1) { cur = 0.05f + u1; step = (0.05f + u2 - cur) / (float)max(n, 1); }                                              // zinv
        else if (kind == 2) { cur = u1 * 4.0f - 2.0f; step = (u2 * 4.0f - 2.0f - cur) / (float)max(n, 1); }                                  // a coordinate, through zero
        else if (kind == 3) { cur = __uint_as_float(lcg(x)); step = __uint_as_float(lcg(x)); }                                             // any bits at all
        else { cur = u1 * 1024.0f; step = __uint_as_float((lcg(x) & 0x807FFFFFu) | ((100u + lcg(x) % 40u) << 23)); }                        // steps of every size
        float seq = cur;
        for (int k = 0; k < n; k++) seq += step;
        const float got = mirt::edge_advance(cur, step, n);
        atomicAdd(&tally[0], 1ull);
        if (__float_as_uint(seq) != __float_as_uint(got) && !(seq != seq && got != got)) {
            if (atomicAdd(&tally[1], 1ull) == 0ull) { tally[2] = __float_as_uint(cur); tally[3] = __float_as_uint(step); tally[4] = (unsigned long long)n; tally[5] = __float_as_uint(seq); tally[6] = __float_as_uint(got); }
        }
    }
}

int main()
{
    {
        unsigned long long *tally;
        hipMalloc(&tally, 8 * 8); hipMemset(tally, 0, 8 * 8);
        hipLaunchKernelGGL(k_advance_check, dim3(1024), dim3(256), 0, 0, 12345u, 16, tally);
        unsigned long long h[8];
        hipMemcpy(h, tally, sizeof h, hipMemcpyDeviceToHost);
        printf("edge_advance against the additions themselves: %llu random (start, step, count <= 4400) cases, %llu predictions off", h[0], h[1]);
        if (h[1]) printf(" (first: start %08llx step %08llx n %llu: sum %08llx predicted %08llx)", h[2], h[3], h[4], h[5], h[6]);
        printf("\n");
        hipFree(tally);
    }
    float *out; long long *cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 256 * 8);
    hipFuncSetAttribute((const void *)k_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 2560 * 60);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 5; mode < 10; mode++)
        for (int wgs : {20}) {
            const int n = 2160;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_chain, dim3(wgs), dim3(256), n * 60, 0, out, cyc, n, mode == 0 && rep == 1 ? -15 : 15, mode);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                long long c[256]; hipMemcpy(c, cyc, sizeof(long long) * wgs, hipMemcpyDeviceToHost);
                if (rep == 2) printf("mode %d wgs %3d: %.1f us, chain %lld memtime ticks (%.1f per step)\n", mode, wgs, ms * 1e3, c[0], (double)c[0] / n);
            }
        }
    return 0;
}
