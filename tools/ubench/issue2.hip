// Issue-rate microbenchmark, second edition (round 4): what ONE SIMD and ONE CU of gfx950 issue per cycle for the
// instruction kinds the wave-family kernels are made of, at 1, 2, 4 and 8 waves per SIMD (every CU busy).
//   valu      independent v_add_u32
//   valu_sgpr v_cmp -> SGPR pair, v_cndmask / v_addc_co_u32 reading an SGPR pair (independent of each other)
//   salu      s_and_b64 / s_or_b64 / s_lshl_b64 on independent pairs
//   lane      v_readlane_b32 / v_writelane_b32 (the scalar spills of wf_step_rows)
//   mixN      N VALU + (4 - N) SALU instructions interleaved in one wave (independent)
//   chain     v_cmp -> s_and_b64 -> v_cndmask -> v_add (dependent: the CA idiom)
//   split     half the waves of a SIMD run `valu`, the other half `salu`
// Output per row: ns and shader cycles per instruction PER SIMD (all its waves together) and instructions per cycle PER CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/issue2.hip -o /tmp/issue2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define N_OUTER 4000
#define REP16(x) x x x x x x x x x x x x x x x x

enum { VALU, VALU_SGPR, SALU, LANE, MIX3, MIX2, MIX1, CHAIN, SPLIT, NKIND };
static const char *NAMES[NKIND] = { "valu", "valu_sgpr", "salu", "lane", "mix 3v+1s", "mix 2v+2s", "mix 1v+3s", "chain", "split v|s" };

template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned long long *out, unsigned *sink)
{
    unsigned v0 = threadIdx.x, v1 = threadIdx.x * 3, v2 = 7, v3 = 11;
    unsigned long long m0 = 0x5555555555555555ull + blockIdx.x, m1 = 3 + blockIdx.x, m2 = 5, m3 = 9;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool second = ((wave >> 2) & 1) != 0;                   // waves 4-7, 12-15: the "other half" of each SIMD
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int o = 0; o < N_OUTER; ++o) {
        if (KIND == VALU || (KIND == SPLIT && !second)) {
            REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
        } else if (KIND == SALU || (KIND == SPLIT && second)) {
            REP16(asm volatile("s_and_b64 %0, %0, %1\n s_or_b64 %1, %1, %2\n s_lshl_b64 %2, %2, 1\n s_andn2_b64 %3, %3, %0"
                               : "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : : "scc");)
        } else if (KIND == VALU_SGPR) {
            REP16(asm volatile("v_cmp_gt_u32_e64 %4, %0, %1\n v_cndmask_b32_e64 %2, %2, %3, %5\n v_addc_co_u32 %3, vcc, %3, %0, %6\n v_cmp_eq_u32_e64 %7, %1, %2"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : : "vcc");)
        } else if (KIND == LANE) {
            unsigned s;
            REP16(asm volatile("v_readlane_b32 %4, %0, 3\n v_writelane_b32 %1, %4, 5\n v_readlane_b32 %4, %2, 7\n v_writelane_b32 %3, %4, 9"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "=s"(s));)
        } else if (KIND == MIX3) {
            REP16(asm volatile("v_add_u32 %0, %0, %1\n s_and_b64 %4, %4, %5\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(m0), "+s"(m1) : : "scc");)
        } else if (KIND == MIX2) {
            REP16(asm volatile("v_add_u32 %0, %0, %1\n s_and_b64 %4, %4, %5\n v_add_u32 %2, %2, %3\n s_or_b64 %6, %6, %7"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : : "scc");)
        } else if (KIND == MIX1) {
            REP16(asm volatile("v_add_u32 %0, %0, %1\n s_and_b64 %4, %4, %5\n s_lshl_b64 %5, %5, 1\n s_or_b64 %6, %6, %7"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : : "scc");)
        } else if (KIND == CHAIN) {
            REP16(asm volatile("v_cmp_gt_u32_e64 %2, %0, %1\n s_and_b64 %2, %2, %3\n v_cndmask_b32_e64 %1, 0, 1, %2\n v_add_u32 %0, %0, %1"
                               : "+v"(v0), "+v"(v1), "+s"(m0), "+s"(m1) : : "scc");)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
    if (v0 + v1 + v2 + v3 + (unsigned)(m0 + m1 + m2 + m3) == 0x12345) sink[0] = 1;
}

template <int KIND>
void run(unsigned long long *d, unsigned *sink)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 1; w <= 8; w *= 2) {                 // waves per SIMD
        const int per_block = w <= 4 ? w : 4, blocks = 256 * (w / per_block), threads = 64 * 4 * per_block;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, sink);
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, sink);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2];
        CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
        const double ghz = (double)h[0] / ((double)h[1] * 10.0);           // shader ticks per ns (s_memrealtime: 100 MHz)
        const double instr = (double)N_OUTER * 64.0;                        // per wave
        const double ns_simd = ms * 1e6 / (instr * w);                      // per instruction, all waves of a SIMD together
        printf("%-10s waves/SIMD=%d  %7.3f ms  clock %.2f GHz  per SIMD: %.3f ns = %.2f cycles per instruction;  per CU: %.2f instructions per cycle\n",
               NAMES[KIND], w, ms, ghz, ns_simd, ns_simd * ghz, 4.0 / (ns_simd * ghz));
        fflush(stdout);
    }
}

int main()
{
    unsigned long long *d; unsigned *sink;
    CHECK(hipMalloc(&d, 16 * 1024)); CHECK(hipMalloc(&sink, 4));
    run<VALU>(d, sink); run<VALU_SGPR>(d, sink); run<SALU>(d, sink); run<LANE>(d, sink);
    run<MIX3>(d, sink); run<MIX2>(d, sink); run<MIX1>(d, sink); run<CHAIN>(d, sink); run<SPLIT>(d, sink);
    return 0;
}
