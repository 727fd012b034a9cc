// tools/graphbench.hip -- what a hipGraph would buy the per-frame launch chains ("next"): a chain of K dependent short
// kernels (each reads what the previous one wrote) launched into a stream one by one, against the same chain captured once and replayed,
// with and without a kernel-node parameter update per replay (a frame's descriptor changes every frame).  Reports host time per chain
// (the launching thread's cost) and device time per chain (events around many chains).
//   hipcc --offload-arch=gfx950 -O2 tools/graphbench.hip -o tools/graphbench
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Desc { float v[16]; int n; };                 // a frame descriptor in kernel arguments, as the library passes it

__global__ void k_link(const float *__restrict__ in, float *__restrict__ out, const Desc d, int spin)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float x = in[i % d.n] + d.v[i & 15];
    for (int s = 0; s < spin; s++) x = x * 1.0001f + 0.5f;          // (a few hundred ns of dependent arithmetic per 100 spins)
    out[i % d.n] = x;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 6, reps = 2000;
    const int blocks = 256, threads = 256, n = blocks * threads;
    float *buf[2];
    CHECK(hipMalloc(&buf[0], n * sizeof(float)));
    CHECK(hipMalloc(&buf[1], n * sizeof(float)));
    CHECK(hipMemset(buf[0], 0, n * sizeof(float)));
    CHECK(hipMemset(buf[1], 0, n * sizeof(float)));
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    Desc d;
    for (int i = 0; i < 16; i++) d.v[i] = 0.25f * i;
    d.n = n;
    for (int spin : { 0, 2000 }) {
        auto chain = [&](hipStream_t st) {
            for (int k = 0; k < K; k++) hipLaunchKernelGGL(k_link, dim3(blocks), dim3(threads), 0, st, buf[k & 1], buf[(k + 1) & 1], d, spin);
        };
        // ---- stream launches ----
        for (int i = 0; i < 200; i++) chain(s);
        CHECK(hipStreamSynchronize(s));
        double t0 = now();
        CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; i++) chain(s);
        CHECK(hipEventRecord(e1, s));
        const double host_stream = (now() - t0) / reps;
        CHECK(hipStreamSynchronize(s));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double dev_stream = ms * 1e-3 / reps;
        // ---- the chain captured once ----
        hipGraph_t g;
        hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        chain(s);
        CHECK(hipStreamEndCapture(s, &g));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        size_t nn = 0;
        CHECK(hipGraphGetNodes(g, nullptr, &nn));
        std::vector<hipGraphNode_t> nodes(nn);
        CHECK(hipGraphGetNodes(g, nodes.data(), &nn));
        for (int i = 0; i < 200; i++) CHECK(hipGraphLaunch(ge, s));
        CHECK(hipStreamSynchronize(s));
        t0 = now();
        CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; i++) CHECK(hipGraphLaunch(ge, s));
        CHECK(hipEventRecord(e1, s));
        const double host_graph = (now() - t0) / reps;
        CHECK(hipStreamSynchronize(s));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double dev_graph = ms * 1e-3 / reps;
        // ---- replayed with every node's arguments rewritten first (a new frame descriptor per replay) ----
        double host_upd = 0, dev_upd = 0;
        {
            std::vector<hipKernelNodeParams> kp(nn);
            bool ok = true;
            for (size_t i = 0; i < nn && ok; i++) ok = hipGraphKernelNodeGetParams(nodes[i], &kp[i]) == hipSuccess;
            if (ok) {
                t0 = now();
                CHECK(hipEventRecord(e0, s));
                for (int r = 0; r < reps; r++) {
                    d.v[0] = (float)r;
                    for (size_t i = 0; i < nn; i++) {
                        const int k = (int)i;
                        const float *in = buf[k & 1];
                        float *out = buf[(k + 1) & 1];
                        void *args[] = { (void *)&in, (void *)&out, (void *)&d, (void *)&spin };
                        hipKernelNodeParams p = kp[i];
                        p.kernelParams = args;
                        p.extra = nullptr;
                        CHECK(hipGraphExecKernelNodeSetParams(ge, nodes[i], &p));
                    }
                    CHECK(hipGraphLaunch(ge, s));
                }
                CHECK(hipEventRecord(e1, s));
                host_upd = (now() - t0) / reps;
                CHECK(hipStreamSynchronize(s));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                dev_upd = ms * 1e-3 / reps;
            }
        }
        printf("chain of %d dependent kernels (%d x %d threads, spin %d): per chain  stream launches host %.2f us device %.2f us | graph replay host %.2f us device %.2f us | graph with %d node updates host %.2f us device %.2f us\n",
               K, blocks, threads, spin, host_stream * 1e6, dev_stream * 1e6, host_graph * 1e6, dev_graph * 1e6, (int)nn, host_upd * 1e6, dev_upd * 1e6);
        CHECK(hipGraphExecDestroy(ge));
        CHECK(hipGraphDestroy(g));
    }
    return 0;
}
