// Floor of a synchronous call on this runtime: launch of a one-lane kernel + wait, nothing else.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_sync_floor tools/launch_sync_floor.hip && /tmp/launch_sync_floor
// (a) hipStreamSynchronize after every launch, (b) the kernel writes a pinned flag and the host spins on it,
// (c) a kernel that also writes 590 KB to pinned host memory (reward + done of 65 536 environments) + hipStreamSynchronize.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>

__global__ void k_empty(uint32_t *) {}
__global__ void k_flag(uint32_t *flag, uint32_t v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void k_out(double *reward, uint8_t *done, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        reward[i] = (double)i;
        done[i] = (uint8_t)i;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    uint32_t *flag;
    CK(hipHostMalloc((void **)&flag, 64, hipHostMallocDefault));
    *flag = 0;
    const int n = 65536;
    double *reward;
    uint8_t *done;
    CK(hipHostMalloc((void **)&reward, n * sizeof(double), hipHostMallocDefault));
    CK(hipHostMalloc((void **)&done, n, hipHostMallocDefault));
    const int iters = 5000;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    for (int rep = 0; rep < 2; ++rep) {
        auto t0 = now();
        for (int i = 0; i < iters; ++i) {
            hipLaunchKernelGGL(k_empty, dim3(1), dim3(1), 0, s, flag);
            CK(hipStreamSynchronize(s));
        }
        auto t1 = now();
        uint32_t seq = *flag;
        for (int i = 0; i < iters; ++i) {
            ++seq;
            hipLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, s, flag, seq);
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) __builtin_ia32_pause();
        }
        auto t2 = now();
        CK(hipStreamSynchronize(s));
        auto t3 = now();
        for (int i = 0; i < iters; ++i) {
            hipLaunchKernelGGL(k_out, dim3(n / 256), dim3(256), 0, s, reward, done, n);
            CK(hipStreamSynchronize(s));
        }
        auto t4 = now();
        printf("rep %d: launch + hipStreamSynchronize %.2f us   launch + pinned flag spin %.2f us   590 KB to pinned memory + hipStreamSynchronize %.2f us\n",
               rep, us(t0, t1) / iters, us(t1, t2) / iters, us(t3, t4) / iters);
    }
    return 0;
}
