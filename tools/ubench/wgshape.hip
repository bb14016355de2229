// GPU-side cost of launching one generation of waves, by workgroup shape: every wave just waits ~10 us of wall clock,
// launches are queued back to back on one stream (the host stays ahead), so (time per launch - 10 us) is what the
// dispatch, ramp and drain of 8 192 waves cost.  hipcc --offload-arch=gfx950 -O3 tools/ubench/wgshape.hip -o /tmp/wgshape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LDS>
__global__ void waiter(unsigned *out, unsigned ticks)
{
    __shared__ unsigned pad[LDS / 4 + 1];
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (ticks == 0xffffffffu) { pad[threadIdx.x % (LDS / 4 + 1)] = 1; out[0] = pad[0]; }
}

int main()
{
    unsigned *out;
    CHECK(hipMalloc((void **)&out, 4));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 300;
    const unsigned ticks = 1000;                       // 10 us at 100 MHz
    struct { int wgs, threads; } shapes[] = { { 2048, 256 }, { 1024, 512 }, { 512, 1024 }, { 4096, 128 }, { 8192, 64 }, { 1024, 256 }, { 4096, 256 } };
    for (auto sh : shapes) {
        for (int lds = 0; lds < 2; ++lds) {
            auto launch = [&]() {
                if (lds) hipLaunchKernelGGL(waiter<8192>, dim3(sh.wgs), dim3(sh.threads), 0, st, out, ticks);
                else hipLaunchKernelGGL(waiter<16>, dim3(sh.wgs), dim3(sh.threads), 0, st, out, ticks);
            };
            for (int i = 0; i < 20; ++i) launch();
            CHECK(hipStreamSynchronize(st));
            CHECK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) launch();
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("%5d workgroups x %4d threads (%5d waves), %s LDS per workgroup: %.2f us per launch (10.00 of it waiting)\n",
                   sh.wgs, sh.threads, sh.wgs * sh.threads / 64, lds ? "8 KB" : "no", ms * 1e3f / iters);
            fflush(stdout);
        }
    }
    return 0;
}
