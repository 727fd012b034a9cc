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
    float2_t p0 = { x0, x1 }, p1 = { x2, x3 }, p2 = { x4, x5 }, p3 = { x6, x7 }, p4 = { x1, x0 }, p5 = { x3, x2 }, p6 = { x5, x4 }, p7 = { x7, x6 };
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
        // one instruction class at a time, written as inline assembly (the compiler fuses or folds the C forms): 16 per iteration on 8 chains
#define OP16(text) \
            asm volatile(text : "+v"(x0) : "v"(a), "v"(b)); asm volatile(text : "+v"(x1) : "v"(a), "v"(b)); asm volatile(text : "+v"(x2) : "v"(a), "v"(b)); asm volatile(text : "+v"(x3) : "v"(a), "v"(b)); \
            asm volatile(text : "+v"(x4) : "v"(a), "v"(b)); asm volatile(text : "+v"(x5) : "v"(a), "v"(b)); asm volatile(text : "+v"(x6) : "v"(a), "v"(b)); asm volatile(text : "+v"(x7) : "v"(a), "v"(b)); \
            asm volatile(text : "+v"(x0) : "v"(b), "v"(a)); asm volatile(text : "+v"(x1) : "v"(b), "v"(a)); asm volatile(text : "+v"(x2) : "v"(b), "v"(a)); asm volatile(text : "+v"(x3) : "v"(b), "v"(a)); \
            asm volatile(text : "+v"(x4) : "v"(b), "v"(a)); asm volatile(text : "+v"(x5) : "v"(b), "v"(a)); asm volatile(text : "+v"(x6) : "v"(b), "v"(a)); asm volatile(text : "+v"(x7) : "v"(b), "v"(a));
        else if (MODE == 5) { OP16("v_xor_b32 %0, %1, %0") }
        else if (MODE == 6) { OP16("v_cndmask_b32 %0, %0, %1, vcc") }
        else if (MODE == 7) { OP16("v_min_f32 %0, %1, %0") }
        else if (MODE == 8) { OP16("v_add_u32 %0, %1, %0") }
        else if (MODE == 9) { OP16("v_mul_f32 %0, %1, %0") }
        else if (MODE == 10) { OP16("v_fma_f32 %0, %1, %0, %2") }
        else if (MODE == 11) { OP16("v_cmp_lt_f32 vcc, %1, %0") }
        else if (MODE == 12) { OP16("v_mov_b32 %0, %1") }
        else if (MODE == 13) { OP16("v_add_f32 %0, %1, %0") }
        else if (MODE == 17) { OP16("v_sub_f32 %0, %1, %0") }
        else if (MODE == 18) { OP16("v_max_f32 %0, %1, %0") }
        else if (MODE == 19) { OP16("v_and_b32 %0, %1, %0") }
        else if (MODE == 20) { OP16("v_lshlrev_b32 %0, 1, %0") }
        else if (MODE == 21) { OP16("v_mul_f32 %0, 2.0, %0") }
        else if (MODE == 22) { OP16("v_cvt_i32_f32 %0, %0") }
        else if (MODE == 23) { OP16("v_mul_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0") }      // (32 per iteration: dependent multiply-add pairs)
        else if (MODE == 26) { OP16("v_fmac_f32 %0, %1, %2") }
        else if (MODE == 27) { OP16("v_rcp_f32 %0, %0") }
        else if (MODE == 28) { OP16("v_sqrt_f32 %0, %0") }
        else if (MODE == 29) {                                                                  // (the mask in a scalar-register pair)
            const unsigned long long mask = 0x5555555555555555ull ^ (unsigned long long)iters;
#define CM(x) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "s"(mask)); 
            CM(x0) CM(x1) CM(x2) CM(x3) CM(x4) CM(x5) CM(x6) CM(x7) CM(x0) CM(x1) CM(x2) CM(x3) CM(x4) CM(x5) CM(x6) CM(x7)
#undef CM
        }
        else if (MODE == 30) { OP16("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") }
        else if (MODE == 31) { OP16("v_div_fixup_f32 %0, %0, %1, %2") }
        else if (MODE == 32) { OP16("v_lshl_add_u32 %0, %0, 1, %1") }
        else if (MODE == 33) { OP16("v_min3_f32 %0, %0, %1, %2") }
        else if (MODE == 34) { OP16("v_med3_f32 %0, %0, %1, %2") }
        else if (MODE == 35) { OP16("v_mul_lo_u32 %0, %0, %1") }
        else if (MODE == 36) { OP16("v_mad_u32_u24 %0, %0, %1, %2") }
        else if (MODE == 37) { OP16("v_bfe_u32 %0, %0, 1, 8") }
        else if (MODE == 38) { OP16("v_cvt_f32_i32 %0, %0") }
#undef OP16
        // ... with a scalar-register operand (what the compiler makes of a uniform multiplier)
#define OPS16(text) \
            asm volatile(text : "+v"(x0) : "s"(a), "s"(b)); asm volatile(text : "+v"(x1) : "s"(a), "s"(b)); asm volatile(text : "+v"(x2) : "s"(a), "s"(b)); asm volatile(text : "+v"(x3) : "s"(a), "s"(b)); \
            asm volatile(text : "+v"(x4) : "s"(a), "s"(b)); asm volatile(text : "+v"(x5) : "s"(a), "s"(b)); asm volatile(text : "+v"(x6) : "s"(a), "s"(b)); asm volatile(text : "+v"(x7) : "s"(a), "s"(b)); \
            asm volatile(text : "+v"(x0) : "s"(b), "s"(a)); asm volatile(text : "+v"(x1) : "s"(b), "s"(a)); asm volatile(text : "+v"(x2) : "s"(b), "s"(a)); asm volatile(text : "+v"(x3) : "s"(b), "s"(a)); \
            asm volatile(text : "+v"(x4) : "s"(b), "s"(a)); asm volatile(text : "+v"(x5) : "s"(b), "s"(a)); asm volatile(text : "+v"(x6) : "s"(b), "s"(a)); asm volatile(text : "+v"(x7) : "s"(b), "s"(a));
        else if (MODE == 14) { OPS16("v_mul_f32 %0, %1, %0") }
        else if (MODE == 24) { OPS16("v_fma_f32 %0, %0, %1, %1") }
#undef OPS16
        // packed forms, every operand a vector-register pair
#define OPP8(text) \
            asm volatile(text : "+v"(p0) : "v"(pa), "v"(pb)); asm volatile(text : "+v"(p1) : "v"(pa), "v"(pb)); asm volatile(text : "+v"(p2) : "v"(pa), "v"(pb)); asm volatile(text : "+v"(p3) : "v"(pa), "v"(pb)); \
            asm volatile(text : "+v"(p4) : "v"(pa), "v"(pb)); asm volatile(text : "+v"(p5) : "v"(pa), "v"(pb)); asm volatile(text : "+v"(p6) : "v"(pa), "v"(pb)); asm volatile(text : "+v"(p7) : "v"(pa), "v"(pb));
        else if (MODE == 15) { OPP8("v_pk_mul_f32 %0, %1, %0") OPP8("v_pk_mul_f32 %0, %2, %0") }
        else if (MODE == 16) { OPP8("v_pk_fma_f32 %0, %1, %0, %2") OPP8("v_pk_fma_f32 %0, %2, %0, %1") }
        else if (MODE == 25) { OPP8("v_pk_add_f32 %0, %1, %0") OPP8("v_pk_add_f32 %0, %2, %0") }
#undef OPP8
    }
    if (MODE == 2 || MODE == 15 || MODE == 16 || MODE == 25) { x0 = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y; x1 = x2 = x3 = x4 = x5 = x6 = x7 = 0; }
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

// what the two counters tick at: one wave spins until s_memtime has advanced by `ticks`, the host times it with events
__global__ void k_spin(unsigned long long ticks, unsigned long long *out)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
    out[0] = __builtin_amdgcn_s_memtime() - t0; out[1] = __builtin_amdgcn_s_memrealtime() - r0;
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
    CHECK(hipHostMalloc((void **)&clk, sizeof(unsigned long long) * 128));
    {
        const unsigned long long ticks = 100000000ull;
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, 0, 1000ull, clk);      // (warm-up)
        CHECK(hipDeviceSynchronize());
        const float ms = time_ms([&] { hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, 0, ticks, clk); }, 1);
        CHECK(hipDeviceSynchronize());
        printf("counters: %llu s_memtime ticks and %llu s_memrealtime ticks in %.3f ms by hipEvents -> s_memtime %.1f MHz, s_memrealtime %.1f MHz\n",
               clk[0], clk[1], ms, clk[0] / (ms * 1e3), clk[1] / (ms * 1e3));
    }
    for (int bpc : { 1, 2, 4, 8 }) {       // blocks of 256 threads per CU = waves per SIMD
        const int grid = cus * bpc;
        const double lanes = (double)grid * 256;
        struct { const char *name; double ops_per_iter; float ms; } r[] = {
            { "mul+add x8 (16 VALU)", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<0>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 0); }) },
            { "fma x8 (8 VALU)", 8, time_ms([&] { hipLaunchKernelGGL(k_valu<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 1); }) },
            { "pk_mul+pk_add x4 (8 VALU)", 8, time_ms([&] { hipLaunchKernelGGL(k_valu<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 2); }) },
            { "IEEE div x8", 8, time_ms([&] { hipLaunchKernelGGL(k_valu<3>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 3); }) },
            { "filter mix (11 VALU)", 11, time_ms([&] { hipLaunchKernelGGL(k_valu<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 4); }) },
            { "v_xor_b32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<5>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 5); }) },
            { "v_cndmask_b32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<6>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 6); }) },
            { "v_min_f32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<7>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 7); }) },
            { "v_add_u32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 8); }) },
            { "v_mul_f32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<9>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 9); }) },
            { "v_fma_f32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<10>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 10); }) },
            { "v_cmp_lt_f32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<11>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 11); }) },
            { "v_mov_b32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<12>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 12); }) },
            { "v_add_f32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<13>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 13); }) },
            { "v_sub_f32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<17>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 14); }) },
            { "v_max_f32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<18>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 15); }) },
            { "v_and_b32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<19>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 16); }) },
            { "v_lshlrev_b32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<20>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 17); }) },
            { "v_mul_f32 by 2.0 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<21>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 18); }) },
            { "v_cvt_i32_f32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<22>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 19); }) },
            { "v_mul_f32 -> v_add_f32 x16 (32)", 32, time_ms([&] { hipLaunchKernelGGL(k_valu<23>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 20); }) },
            { "v_mul_f32 by an SGPR x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<14>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 21); }) },
            { "v_fma_f32 with SGPRs x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<24>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 22); }) },
            { "v_pk_mul_f32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<15>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 23); }) },
            { "v_pk_fma_f32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<16>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 24); }) },
            { "v_pk_add_f32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<25>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 25); }) },
            { "v_fmac_f32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<26>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 26); }) },
            { "v_rcp_f32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<27>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 27); }) },
            { "v_sqrt_f32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<28>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 28); }) },
            { "v_cndmask_b32 (SGPR mask) x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<29>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 29); }) },
            { "v_mov_b32 dpp row_shr x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<30>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 30); }) },
            { "v_div_fixup_f32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<31>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 31); }) },
            { "v_lshl_add_u32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<32>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 32); }) },
            { "v_min3_f32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<33>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 33); }) },
            { "v_med3_f32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<34>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 34); }) },
            { "v_mul_lo_u32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<35>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 35); }) },
            { "v_mad_u32_u24 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<36>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 36); }) },
            { "v_bfe_u32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<37>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 37); }) },
            { "v_cvt_f32_i32 x16", 16, time_ms([&] { hipLaunchKernelGGL(k_valu<38>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, clk + 2 * 38); }) },
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
