// Fixed cost of a step: back-to-back launches of an (almost) empty kernel with the step kernel's grid
// (2048 workgroups x 256 threads per 65 536-environment step, split over 1, 2 or 3 streams).
// MI355X, ROCm 7.2: 2.6 us per step with one launch, 5.7 us with two launches on two streams, 9.2 us with three:
// dispatch is ~2.8 us per launch and does not parallelise across streams.  (depth 1 adds one load per thread.)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/launch.hip -o /tmp/launch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void chain(const unsigned *p, unsigned *out, int depth, unsigned mask)
{
    unsigned idx = (blockIdx.x * 256u + threadIdx.x) & mask;
    for (int d = 0; d < depth; ++d) idx = p[idx] & mask;          // dependent loads
    if (idx == 0xffffffffu) out[0] = idx;
}

int main()
{
    const unsigned N = 1u << 22;
    unsigned *p, *out;
    CHECK(hipMalloc((void **)&p, N * 4));
    CHECK(hipMalloc((void **)&out, 4));
    unsigned *h = (unsigned *)malloc(N * 4);
    for (unsigned i = 0; i < N; ++i) h[i] = (i * 2654435761u) >> 10;
    CHECK(hipMemcpy(p, h, N * 4, hipMemcpyHostToDevice));
    hipStream_t st[3];
    for (int i = 0; i < 3; ++i) CHECK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    hipEvent_t e0, e1, ej[3];
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CHECK(hipEventCreateWithFlags(&ej[i], hipEventDisableTiming));
    const int iters = 500;
    for (int depth = 0; depth <= 1; ++depth)
        for (int parts = 1; parts <= 3; ++parts) {
            const int wgs = 2048 / parts;                         // 65536 envs / 8 per wave / 4 waves per WG, split
            auto step = [&]() { for (int q = 0; q < parts; ++q) hipLaunchKernelGGL(chain, dim3(wgs), dim3(256), 0, st[q], p, out, depth, N - 1); };
            for (int i = 0; i < 50; ++i) step();
            for (int q = 0; q < parts; ++q) CHECK(hipStreamSynchronize(st[q]));
            CHECK(hipEventRecord(e0, st[0]));
            for (int q = 1; q < parts; ++q) CHECK(hipStreamWaitEvent(st[q], e0, 0));
            for (int i = 0; i < iters; ++i) step();
            for (int q = 1; q < parts; ++q) { CHECK(hipEventRecord(ej[q], st[q])); CHECK(hipStreamWaitEvent(st[0], ej[q], 0)); }
            CHECK(hipEventRecord(e1, st[0]));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("dependent loads per wave %d, launches per step %d: %.2f us per step\n", depth, parts, ms * 1e3f / iters);
            fflush(stdout);
        }
    return 0;
}
