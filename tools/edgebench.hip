#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
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
int main()
{
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
