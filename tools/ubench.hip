// tools/ubench.hip -- VALU / LDS rate probes for gfx950 used to size the ray-triangle inner loop
// (which mix of v_mul/v_add, packed v_pk_mul/v_pk_add, IEEE divide and LDS broadcast reads the chip sustains).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/ubench.hip -o tools/ubench (__graft_entry__.build() does); run on
// the GPU box.  (Without -fno-slp-vectorize the compiler pairs the "scalar" chains into v_pk_* itself and the first two lines measure packed code.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2_t __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_valu(float *out, int iters, float a, float b, unsigned long long *clk)
{
    // the shader clock under this load: s_memtime (shader-clock ticks) against s_memrealtime (100 MHz) over the loop
    const unsigned long long tick0 = __builtin_amdgcn_s_memtime(), real0 = __builtin_amdgcn_s_memrealtime();
    // 8 independent chains per lane
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float2_t p0 = { x0, x1 }, p1 = { x2, x3 }, p2 = { x4, x5 }, p3 = { x6, x7 };
    float2_t pa = { a, a }, pb = { b, b };
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {          // 8 mul + 8 add (unfused)
            x0 = x0 * a; x1 = x1 * a; x2 = x2 * a; x3 = x3 * a; x4 = x4 * a; x5 = x5 * a; x6 = x6 * a; x7 = x7 * a;
            x0 = x0 + b; x1 = x1 + b; x2 = x2 + b; x3 = x3 + b; x4 = x4 + b; x5 = x5 + b; x6 = x6 + b; x7 = x7 + b;
        } else if (MODE == 1) {   // 8 fma
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        } else if (MODE == 2) {   // 4 pk_mul + 4 pk_add (8 mul + 8 add worth of flops)
            p0 = p0 * pa; p1 = p1 * pa; p2 = p2 * pa; p3 = p3 * pa;
            p0 = p0 + pb; p1 = p1 + pb; p2 = p2 + pb; p3 = p3 + pb;
        } else if (MODE == 3) {   // 8 IEEE divides
            x0 = a / x0; x1 = a / x1; x2 = a / x2; x3 = a / x3; x4 = a / x4; x5 = a / x5; x6 = a / x6; x7 = a / x7;
        } else if (MODE == 4) {   // 8 x (xor + min3-ish filter ops): and, xor, xor, add, fma, min3, cmp ~ 7 ops
            unsigned s = __float_as_uint(x0) & 0x80000000u;
            float aa = __uint_as_float(__float_as_uint(x1) ^ s), bb = __uint_as_float(__float_as_uint(x2) ^ s);
            float sl = __builtin_fmaf(fabsf(x0), a, -(aa + bb));
            x3 += (fminf(fminf(aa, bb), sl) >= b) ? 1.0f : 0.0f;
            x0 = x0 * a; x1 = x1 + b; x2 = x2 + a;
        }
    }
    if (MODE == 2) { x0 = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y; x1 = x2 = x3 = x4 = x5 = x6 = x7 = 0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (clk && threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - tick0; clk[1] = __builtin_amdgcn_s_memrealtime() - real0; }
}

// LDS broadcast read rate: every lane reads the same 48 bytes (3 x b128), NREAD rows per iteration
__global__ __launch_bounds__(256) void k_lds(float *out, int iters)
{
    __shared__ float4 tab[3 * 1024];
    for (int i = threadIdx.x; i < 3 * 1024; i += 256) tab[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    float acc = 0;
    for (int it = 0; it < iters; it++)
        for (int j = 0; j < 1024; j++) {
            float4 a = tab[3 * j], b = tab[3 * j + 1], c = tab[3 * j + 2];
            acc += a.x + b.y + c.z;
        }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename F>
float time_ms(F launch, int reps = 5)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(e0, 0);
        launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    const int cus = p.multiProcessorCount;
    float *out;
    CHECK(hipMalloc((void **)&out, sizeof(float) * 256 * cus * 8));
    const int iters = 20000;
    unsigned long long *clk;                 // [mode][ticks, 100 MHz ticks] of workgroup 0 of the latest launch
    CHECK(hipHostMalloc((void **)&clk, sizeof(unsigned long long) * 16));
    for (int bpc : { 1, 2, 4, 8 }) {       // blocks of 256 threads per CU = waves per SIMD
        const int grid = cus * bpc;
        const double lanes = (double)grid * 256;
        struct { const char *name; double ops_per_iter; float ms; } r[] = {
            { "mul+add x8 (16 VALU)", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<0>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 0); }) },
            { "fma x8 (8 VALU)", 8, time_ms([&] { hipLaunchKernelGGL(k_valu<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 1); }) },
            { "pk_mul+pk_add x4 (8 VALU)", 8, time_ms([&] { hipLaunchKernelGGL(k_valu<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 2); }) },
            { "IEEE div x8", 8, time_ms([&] { hipLaunchKernelGGL(k_valu<3>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 3); }) },
            { "filter mix (~10 VALU)", 10, time_ms([&] { hipLaunchKernelGGL(k_valu<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 4); }) },
        };
        CHECK(hipDeviceSynchronize());
        int mode = 0;
        for (auto &x : r) {
            const double inst = lanes / 64.0 * iters * x.ops_per_iter;      // wave-instructions
            const double per_simd_cycle = inst / (cus * 4.0) / (x.ms * 1e-3 * 2.4e9);
            const double ghz = clk[2 * mode + 1] ? 0.1 * (double)clk[2 * mode] / (double)clk[2 * mode + 1] : 0.0;
            printf("waves/SIMD=%d  %-28s %8.3f ms  %7.3f wave-instr/clk/SIMD (@2.4GHz)  %6.2f T lane-ops/s   s_memtime / s_memrealtime: %.3f GHz\n", bpc, x.name, x.ms,
                   per_simd_cycle, lanes * iters * x.ops_per_iter / (x.ms * 1e-3) / 1e12, ghz);
            mode++;
        }
        const int lit = 20;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_lds, dim3(grid), dim3(256), 0, 0, out, lit); });
        const double rows = lanes / 64.0 * lit * 1024.0;
        printf("waves/SIMD=%d  %-28s %8.3f ms  %7.3f rows(48B bcast)/clk/CU  %6.2f G rows/s\n", bpc, "LDS 3xb128 broadcast", ms,
               rows / cus / (ms * 1e-3 * 2.4e9), rows / (ms * 1e-3) / 1e9);
    }
    return 0;
}
