// Issue-rate microbenchmark for gfx950 (round 1 finding, MI355X: VALU-only 1.19 ns per instruction per SIMD at >= 2
// waves/SIMD; 2 VALU waves + 2 SALU waves per SIMD 1.15 ns per instruction -> the two pipes do NOT co-issue across
// waves here, every instruction costs ~2.5 cycles of its SIMD; v_cmp->s_and->v_cndmask->v_add chains 1.43 ns at 4
// waves/SIMD.  The KIND 1/2 (pure SALU asm) variants mis-measure and are kept only for completeness.)
// how many cycles per instruction does one SIMD spend on
// VALU / SALU streams, alone and mixed, at 1..8 waves per SIMD.  Build: hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define N_INNER 64
#define N_OUTER 20000

template <int KIND>
__global__ void k(unsigned long long *out, unsigned *sink, int waves_valu)
{
    // one block = waves on ONE CU; with blockDim = 64*w*4 the w waves/SIMD share each SIMD
    unsigned v0 = threadIdx.x, v1 = threadIdx.x * 3, v2 = 7, v3 = 11;
    unsigned s0 = blockIdx.x, s1 = 3, s2 = 5, s3 = 9;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool do_valu = KIND == 0 || KIND == 2 || (KIND == 3 && (wave / 4) % 2 == 0);
    const bool do_salu = KIND == 1 || KIND == 2 || (KIND == 3 && (wave / 4) % 2 == 1);
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int o = 0; o < N_OUTER; ++o) {
        if (KIND == 4) {
            // dependent chain: v_cmp -> s_and -> v_cndmask -> v_add (the CA kernels' idiom)
#pragma unroll
            for (int i = 0; i < N_INNER / 4; ++i) {
                unsigned long long m = __ballot(v0 > v1);
                m &= 0x5555555555555555ull + s0;
                v0 += __builtin_amdgcn_inverse_ballot_w64(m) ? 1u : 0u;
                asm volatile("" : "+v"(v0));
            }
        } else {
            if (do_valu) {
#pragma unroll
                for (int i = 0; i < N_INNER / 4; ++i) {
                    asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0"
                                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
                }
            }
            if (do_salu) {
#pragma unroll
                for (int i = 0; i < N_INNER / 4; ++i) {
                    asm volatile("s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0"
                                 : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 64 + wave] = t1 - t0;
    if (v0 + v1 + v2 + v3 + s0 + s1 + s2 + s3 == 0x12345) sink[0] = 1;
}

template <int KIND>
void run(const char *name, int per_wave_instrs)
{
    unsigned long long *d; unsigned *sink;
    hipMalloc(&d, 64 * 8 * 4096); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 1; w <= 4; w *= 2) {           // waves per SIMD (one 256*w-thread block per CU, 256 CUs)
        const int threads = 64 * 4 * w;
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, sink, 0);   // warm-up / clocks
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, sink, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_wave = (double)N_OUTER * per_wave_instrs;
        const double ns = ms * 1e6;
        // per SIMD: w waves, each instr_per_wave instructions, in ns
        printf("%-28s waves/SIMD=%d  %8.3f ms  ns per instr per wave=%6.3f  ns per instr per SIMD=%6.3f\n", name, w, ms,
               ns / instr_per_wave, ns / (instr_per_wave * w));
    }
    hipFree(d); hipFree(sink);
}

int main()
{
    run<0>("VALU only (v_add chain4)", N_INNER);
    run<1>("SALU only (s_add chain4)", N_INNER);
    run<2>("VALU+SALU same wave", 2 * N_INNER);
    run<3>("VALU waves + SALU waves", N_INNER);
    run<4>("cmp->s_and->cndmask->add", N_INNER);   // 16 chains of 4 instrs
    return 0;
}
