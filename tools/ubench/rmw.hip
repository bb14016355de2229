// Practical ceiling of the step kernel's memory pattern on MI355X: read every environment's cells,
// change them, write them back IN PLACE, with no CA work at all.
//   pattern 0: the step kernel's layout - a wave owns 8 consecutive environments of 784 B, lanes 0..48 move
//              16 B each, one environment per iteration, next environment prefetched
//   pattern 1: plain grid-stride 16 B per thread over the same bytes
//   pattern 2: half a grid row (7 cells, 28 B) per lane, 2 environments per iteration (lanes 0..27, 30..57)
//   pattern 3: a grid row (14 cells, 56 B) per lane, 4 environments per iteration (lanes 15h..15h+13)
// over a 51 MB state (65536 x 14 x 14, fits the 256 MB memory-side cache) and over a 4 GB state.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/rmw.hip -o /tmp/rmw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr unsigned ENV_BYTES = 784, LANES = 49, G = 8;

__global__ __launch_bounds__(256) void rmw_env(char *p, unsigned n_env)
{
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    const unsigned e0 = wave * G;
    if (e0 >= n_env || lane >= LANES) return;
    const unsigned ng = min(G, n_env - e0);
    uint4 nxt = *reinterpret_cast<const uint4 *>(p + (size_t)e0 * ENV_BYTES + lane * 16u);
    for (unsigned k = 0; k < ng; ++k) {
        const uint4 c = nxt;
        const unsigned en = e0 + min(k + 1u, ng - 1u);
        nxt = *reinterpret_cast<const uint4 *>(p + (size_t)en * ENV_BYTES + lane * 16u);
        *reinterpret_cast<uint4 *>(p + (size_t)(e0 + k) * ENV_BYTES + lane * 16u) = make_uint4(c.x + 1u, c.y + 1u, c.z + 1u, c.w + 1u);
    }
}

// CPL dwords per lane, LPE lanes per environment, EPW environments per iteration, one dead lane group between envs
template <int CPL, int LPE, int EPW, int ENVS_PER_WAVE>
__global__ __launch_bounds__(256) void rmw_rows(char *p, unsigned n_env)
{
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    const unsigned e0 = wave * ENVS_PER_WAVE;
    constexpr unsigned PITCH = LPE + 64 / LPE / 8 + 1;       // 15 for LPE 14, 30 for LPE 28 (dead lanes between envs)
    const unsigned h = lane / PITCH, l = lane % PITCH;
    if (e0 >= n_env || l >= LPE || h >= EPW) return;
    const unsigned off = h * ENV_BYTES + l * (CPL * 4u);
    unsigned c[CPL], n[CPL];
    auto load = [&](unsigned e, unsigned *r) {
        const char *q = p + (size_t)e * ENV_BYTES + off;
        if (CPL == 7) {
            const uint4 a = *reinterpret_cast<const uint4 *>(q);
            const uint3 b = *reinterpret_cast<const uint3 *>(q + 16);
            r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z;
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const uint4 a = *reinterpret_cast<const uint4 *>(q + 16 * i);
                r[4 * i] = a.x; r[4 * i + 1] = a.y; r[4 * i + 2] = a.z; r[4 * i + 3] = a.w;
            }
            const uint2 b = *reinterpret_cast<const uint2 *>(q + 48);
            r[12] = b.x; r[13] = b.y;
        }
    };
    load(e0, n);
    for (unsigned k = 0; k < ENVS_PER_WAVE; k += EPW) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) c[i] = n[i] + 1u;
        load(e0 + min(k + EPW, (unsigned)ENVS_PER_WAVE - EPW), n);
        char *q = p + (size_t)(e0 + k) * ENV_BYTES + off;
        if (CPL == 7) {
            *reinterpret_cast<uint4 *>(q) = make_uint4(c[0], c[1], c[2], c[3]);
            *reinterpret_cast<uint3 *>(q + 16) = make_uint3(c[4], c[5], c[6]);
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) *reinterpret_cast<uint4 *>(q + 16 * i) = make_uint4(c[4 * i], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3]);
            *reinterpret_cast<uint2 *>(q + 48) = make_uint2(c[12], c[13]);
        }
    }
}

// pattern 5: half-row loads as pattern 2, but the stores leave through an LDS transpose as dense 16-byte pieces
__global__ __launch_bounds__(256) void rmw_rows_lds(char *p, unsigned n_env)
{
    constexpr int CPL = 7, LPE = 28, EPW = 2, ENVS_PER_WAVE = 8;
    __shared__ __attribute__((aligned(16))) unsigned lds[4][EPW * 196 + 8];
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wv = threadIdx.x >> 6;
    const unsigned wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wv);
    const unsigned e0 = wave * ENVS_PER_WAVE;
    const unsigned h = lane / 30u, l = lane % 30u;
    if (e0 >= n_env || l >= LPE || h >= EPW) return;
    const unsigned off = h * ENV_BYTES + l * (CPL * 4u);
    const unsigned cell0 = h * 196u + l * 7u;                  // dword index of this lane's first cell in the pair image
    const unsigned rank = h * 28u + l;                          // dense index among the 56 working lanes
    unsigned c[CPL], n[CPL];
    auto load = [&](unsigned e, unsigned *r) {
        const char *q = p + (size_t)e * ENV_BYTES + off;
        const uint4 a = *reinterpret_cast<const uint4 *>(q);
        const uint3 b = *reinterpret_cast<const uint3 *>(q + 16);
        r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z;
    };
    load(e0, n);
    for (unsigned k = 0; k < ENVS_PER_WAVE; k += EPW) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) c[i] = n[i] + 1u;
        load(e0 + min(k + EPW, (unsigned)ENVS_PER_WAVE - EPW), n);
#pragma unroll
        for (int i = 0; i < CPL; ++i) lds[wv][cell0 + i] = c[i];
        char *q = p + (size_t)(e0 + k) * ENV_BYTES;            // 1568 contiguous bytes = 98 pieces of 16
        const uint4 v0 = *reinterpret_cast<const uint4 *>(&lds[wv][rank * 4u]);
        *reinterpret_cast<uint4 *>(q + rank * 16u) = v0;
        if (rank < 42u) {
            const uint4 v1 = *reinterpret_cast<const uint4 *>(&lds[wv][(rank + 56u) * 4u]);
            *reinterpret_cast<uint4 *>(q + (rank + 56u) * 16u) = v1;
        }
    }
}

__global__ __launch_bounds__(256) void rmw_flat(uint4 *p, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        uint4 c = p[i];
        p[i] = make_uint4(c.x + 1u, c.y + 1u, c.z + 1u, c.w + 1u);
    }
}

static float run(int pattern, char *buf, size_t n_env, int parts, int iters, hipStream_t *st)
{
    hipEvent_t e0, e1, ej[4];
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 4; ++i) CHECK(hipEventCreateWithFlags(&ej[i], hipEventDisableTiming));
    const size_t per = (n_env + parts - 1) / parts;
    auto launch = [&](int it) {
        for (int q = 0; q < parts; ++q) {
            const size_t lo = q * per, n = (lo + per <= n_env ? per : n_env - lo);
            char *b = buf + lo * ENV_BYTES;
            if (pattern == 0) {
                const unsigned waves = (unsigned)((n + G - 1) / G);
                hipLaunchKernelGGL(rmw_env, dim3((waves + 3) / 4), dim3(256), 0, st[q], b, (unsigned)n);
            } else if (pattern == 2) {
                const unsigned waves = (unsigned)(n / 8);
                hipLaunchKernelGGL((rmw_rows<7, 28, 2, 8>), dim3((waves + 3) / 4), dim3(256), 0, st[q], b, (unsigned)n);
            } else if (pattern == 3) {
                const unsigned waves = (unsigned)(n / 16);
                hipLaunchKernelGGL((rmw_rows<14, 14, 4, 16>), dim3((waves + 3) / 4), dim3(256), 0, st[q], b, (unsigned)n);
            } else if (pattern == 5) {
                const unsigned waves = (unsigned)(n / 8);
                hipLaunchKernelGGL(rmw_rows_lds, dim3((waves + 3) / 4), dim3(256), 0, st[q], b, (unsigned)n);
            } else if (pattern == 4) {
                const unsigned waves = (unsigned)(n / 8);
                hipLaunchKernelGGL((rmw_rows<14, 14, 4, 8>), dim3((waves + 3) / 4), dim3(256), 0, st[q], b, (unsigned)n);
            } else {
                const size_t n16 = n * ENV_BYTES / 16;
                hipLaunchKernelGGL(rmw_flat, dim3(2048), dim3(256), 0, st[q], (uint4 *)b, n16);
            }
        }
    };
    for (int it = 0; it < 20; ++it) launch(it);
    for (int q = 0; q < parts; ++q) CHECK(hipStreamSynchronize(st[q]));
    CHECK(hipEventRecord(e0, st[0]));
    for (int q = 1; q < parts; ++q) CHECK(hipStreamWaitEvent(st[q], e0, 0));
    for (int it = 0; it < iters; ++it) launch(it);
    for (int q = 1; q < parts; ++q) { CHECK(hipEventRecord(ej[q], st[q])); CHECK(hipStreamWaitEvent(st[0], ej[q], 0)); }
    CHECK(hipEventRecord(e1, st[0]));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

int main()
{
    hipStream_t st[4];
    for (int i = 0; i < 4; ++i) CHECK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    const size_t sizes[2] = { 65536, 65536ull * 80 };        // 51 MB and 4.1 GB
    for (size_t n_env : sizes) {
        char *buf;
        CHECK(hipMalloc((void **)&buf, n_env * ENV_BYTES));
        CHECK(hipMemset(buf, 0, n_env * ENV_BYTES));
        for (int pattern = 0; pattern < 6; ++pattern)
            for (int parts = 1; parts <= 2; ++parts) {
                const int iters = n_env > 100000 ? 10 : 300;
                const float us = run(pattern, buf, n_env, parts, iters, st);
                const double gbs = 2.0 * n_env * ENV_BYTES / (us * 1e-6) / 1e9;
                printf("envs=%zu pattern=%s parts=%d  %.2f us/pass  %.0f GB/s (read+write)\n", n_env,
                       pattern == 0 ? "env-wave" : pattern == 1 ? "flat" : pattern == 2 ? "halfrow-2env" : pattern == 3 ? "row-4env-16" : pattern == 4 ? "row-4env-8" : "halfrow-lds-store", parts, us, gbs);
                fflush(stdout);
            }
        CHECK(hipFree(buf));
    }
    return 0;
}
