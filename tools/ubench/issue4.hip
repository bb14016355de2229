// Issue-rate microbenchmark, fourth edition (round 4): ONE instruction kind per row
// issue2 showed: plain VALU 2.3 cycles per instruction per SIMD at 8 waves per SIMD, but v_cmp -> SGPR pair / v_cndmask and
// v_addc reading an SGPR pair / v_readlane / v_writelane 4.0 - the rate of the CU's ONE scalar unit (SALU 4.3 per SIMD =
// 0.93 per cycle per CU).  This one separates the forms (VCC against an arbitrary SGPR pair, SDWA, SGPR as a 32-bit source)
// and asks whether such instructions and SALU instructions ADD on that port (same wave, and split over the waves).
// Rows: 4 and 8 waves per SIMD, every CU busy; cycles per instruction per SIMD (all its waves together).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/issue3.hip -o /tmp/issue3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define N_OUTER 3000
#define REP16(x) x x x x x x x x x x x x x x x x
#define OPS "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3), "+s"(s0), "+s"(s1)
// %0-%3 VGPRs, %4-%7 SGPR pairs, %8-%9 SGPRs
#define KINDS(X) \
    X(0, "v_add_u32", "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0") \
    X(1, "v_sub_u32", "v_sub_u32 %0, %0, %1\n v_sub_u32 %1, %1, %2\n v_sub_u32 %2, %2, %3\n v_sub_u32 %3, %3, %0") \
    X(2, "v_and_b32", "v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %0") \
    X(3, "v_or_b32", "v_or_b32 %0, %0, %1\n v_or_b32 %1, %1, %2\n v_or_b32 %2, %2, %3\n v_or_b32 %3, %3, %0") \
    X(4, "v_xor_b32", "v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0") \
    X(5, "v_lshlrev_b32 (const shift)", "v_lshlrev_b32 %0, 3, %1\n v_lshlrev_b32 %1, 3, %2\n v_lshlrev_b32 %2, 3, %3\n v_lshlrev_b32 %3, 3, %0") \
    X(6, "v_lshrrev_b32 (vgpr shift)", "v_lshrrev_b32 %0, %1, %0\n v_lshrrev_b32 %1, %2, %1\n v_lshrrev_b32 %2, %3, %2\n v_lshrrev_b32 %3, %0, %3") \
    X(7, "v_mov_b32 v, v", "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0") \
    X(8, "v_mov_b32 v, s", "v_mov_b32 %0, %8\n v_mov_b32 %1, %9\n v_mov_b32 %2, %8\n v_mov_b32 %3, %9") \
    X(9, "v_mov_b32 v, literal", "v_mov_b32 %0, 0x12345\n v_mov_b32 %1, 0x23456\n v_mov_b32 %2, 0x34567\n v_mov_b32 %3, 0x45678") \
    X(10, "v_add_u32 v, literal, v", "v_add_u32 %0, 0x12345, %1\n v_add_u32 %1, 0x23456, %2\n v_add_u32 %2, 0x34567, %3\n v_add_u32 %3, 0x45678, %0") \
    X(11, "v_add_u32 v, inline const, v", "v_add_u32 %0, 7, %1\n v_add_u32 %1, 9, %2\n v_add_u32 %2, 11, %3\n v_add_u32 %3, 13, %0") \
    X(12, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 3, %1\n v_lshl_add_u32 %1, %1, 3, %2\n v_lshl_add_u32 %2, %2, 3, %3\n v_lshl_add_u32 %3, %3, 3, %0") \
    X(13, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2\n v_and_or_b32 %1, %1, %2, %3\n v_and_or_b32 %2, %2, %3, %0\n v_and_or_b32 %3, %3, %0, %1") \
    X(14, "v_or3_b32", "v_or3_b32 %0, %0, %1, %2\n v_or3_b32 %1, %1, %2, %3\n v_or3_b32 %2, %2, %3, %0\n v_or3_b32 %3, %3, %0, %1") \
    X(15, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %0\n v_add3_u32 %3, %3, %0, %1") \
    X(16, "v_bfe_u32", "v_bfe_u32 %0, %1, 8, 8\n v_bfe_u32 %1, %2, 8, 8\n v_bfe_u32 %2, %3, 8, 8\n v_bfe_u32 %3, %0, 8, 8") \
    X(17, "v_bfi_b32", "v_bfi_b32 %0, %0, %1, %2\n v_bfi_b32 %1, %1, %2, %3\n v_bfi_b32 %2, %2, %3, %0\n v_bfi_b32 %3, %3, %0, %1") \
    X(18, "v_bitop3_b32", "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x80\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0x80\n v_bitop3_b32 %2, %2, %3, %0 bitop3:0x80\n v_bitop3_b32 %3, %3, %0, %1 bitop3:0x80") \
    X(19, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %1, %1, %2\n v_pk_add_u16 %2, %2, %3\n v_pk_add_u16 %3, %3, %0") \
    X(20, "v_min_u32", "v_min_u32 %0, %0, %1\n v_min_u32 %1, %1, %2\n v_min_u32 %2, %2, %3\n v_min_u32 %3, %3, %0") \
    X(21, "v_min_u32_sdwa", "v_min_u32_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_min_u32_sdwa %1, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_min_u32_sdwa %2, %2, %3 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_min_u32_sdwa %3, %3, %0 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD") \
    X(22, "v_sub_u32_sdwa (byte src)", "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_sub_u32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_sub_u32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_sub_u32_sdwa %3, %3, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2") \
    X(23, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %0\n v_mad_u32_u24 %3, %3, %0, %1") \
    X(24, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %3, %3, %0") \
    X(25, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0") \
    X(26, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %0\n v_perm_b32 %3, %3, %0, %1") \
    X(27, "v_alignbyte_b32", "v_alignbyte_b32 %0, %0, %1, 1\n v_alignbyte_b32 %1, %1, %2, 1\n v_alignbyte_b32 %2, %2, %3, 1\n v_alignbyte_b32 %3, %3, %0, 1") \
    X(28, "v_sad_u8", "v_sad_u8 %0, %0, %1, %2\n v_sad_u8 %1, %1, %2, %3\n v_sad_u8 %2, %2, %3, %0\n v_sad_u8 %3, %3, %0, %1") \
    X(29, "v_dot4_u32_u8", "v_dot4_u32_u8 %0, %0, %1, %2\n v_dot4_u32_u8 %1, %1, %2, %3\n v_dot4_u32_u8 %2, %2, %3, %0\n v_dot4_u32_u8 %3, %3, %0, %1") \
    X(30, "v_cndmask_b32_e32 <- vcc (independent dst)", "v_cndmask_b32_e32 %0, %1, %2, vcc\n v_cndmask_b32_e32 %1, %2, %3, vcc\n v_cndmask_b32_e32 %2, %3, %0, vcc\n v_cndmask_b32_e32 %3, %0, %1, vcc") \
    X(31, "v_cndmask_b32_e64 0/1 <- sgpr pair", "v_cndmask_b32_e64 %0, 0, 1, %4\n v_cndmask_b32_e64 %1, 0, 1, %5\n v_cndmask_b32_e64 %2, 0, 1, %6\n v_cndmask_b32_e64 %3, 0, 1, %7") \
    X(32, "v_add_co_u32 -> vcc", "v_add_co_u32 %0, vcc, %0, %1\n v_add_co_u32 %1, vcc, %1, %2\n v_add_co_u32 %2, vcc, %2, %3\n v_add_co_u32 %3, vcc, %3, %0") \
    X(33, "v_cmp_lt_i16_e64 -> sgpr", "v_cmp_lt_i16_e64 %4, %0, %1\n v_cmp_lt_i16_e64 %5, %1, %2\n v_cmp_lt_i16_e64 %6, %2, %3\n v_cmp_lt_i16_e64 %7, %3, %0") \
    X(34, "v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 %0, -1, %1\n v_mbcnt_lo_u32_b32 %1, -1, %2\n v_mbcnt_lo_u32_b32 %2, -1, %3\n v_mbcnt_lo_u32_b32 %3, -1, %0") \
    X(35, "v_add_f32", "v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %0") \
    X(36, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1") \
    X(37, "v_add_u32 dpp quad_perm", "v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
    X(38, "s_and_b64", "s_and_b64 %4, %4, %5\n s_and_b64 %5, %5, %6\n s_and_b64 %6, %6, %7\n s_and_b64 %7, %7, %4") \
    X(39, "s_lshl_b64", "s_lshl_b64 %4, %4, 1\n s_lshl_b64 %5, %5, 1\n s_lshl_b64 %6, %6, 1\n s_lshl_b64 %7, %7, 1") \
    X(40, "s_mov_b64 literal-free", "s_mov_b64 %4, %5\n s_mov_b64 %5, %6\n s_mov_b64 %6, %7\n s_mov_b64 %7, %4") \
    X(41, "s_nop 0", "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0")

#define DECL(n, name, body) if constexpr (KIND == n) { REP16(asm volatile(body : OPS : : "vcc", "scc");) }
#define NAME(n, name, body) name,
static const char *NAMES[] = { KINDS(NAME) };
constexpr int NK = sizeof(NAMES) / sizeof(NAMES[0]);

// the whole timed loop of one kind (kept in one function: no SGPR value is merged across the split's branch)
template <int KIND>
__device__ __forceinline__ void loop(unsigned *sink)
{
    unsigned v0 = threadIdx.x, v1 = threadIdx.x * 3, v2 = 7, v3 = 11;
    unsigned long long m0 = 0x5555555555555555ull + blockIdx.x, m1 = 3 + blockIdx.x, m2 = 5, m3 = 9;
    unsigned s0 = blockIdx.x, s1 = 17;
    asm volatile("v_cmp_gt_u32_e32 vcc, %0, %1" : : "v"(v0), "v"(v1) : "vcc");
    for (int o = 0; o < N_OUTER; ++o) {
        KINDS(DECL)
    }
    if (v0 + v1 + v2 + v3 + (unsigned)(m0 + m1 + m2 + m3) + s0 + s1 == 0x12345) sink[0] = 1;
}

// KA != KB: waves 4-7 / 12-15 of a block (the second wave of each SIMD pair) run kind KB instead of KA
template <int KA, int KB>
__global__ __launch_bounds__(1024) void k(unsigned long long *out, unsigned *sink)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool second = ((wave >> 2) & 1) != 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (KA == KB || !second) loop<KA>(sink);
    else loop<KB>(sink);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int ka, int kb>
void run(unsigned long long *d, unsigned *sink)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 4; w <= 8; w *= 2) {                 // waves per SIMD
        const int blocks = 256 * (w / 4), threads = 1024;
        hipLaunchKernelGGL((k<ka, kb>), dim3(blocks), dim3(threads), 0, 0, d, sink);
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k<ka, kb>), dim3(blocks), dim3(threads), 0, 0, d, sink);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2];
        CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
        const double ghz = (double)h[0] / ((double)h[1] * 10.0);
        const double instr = (double)N_OUTER * 64.0;
        const double ns_simd = ms * 1e6 / (instr * w);
        if (ka == kb) printf("%-52s", NAMES[ka]);
        else printf("SPLIT %-22s | %-21s", NAMES[ka], NAMES[kb]);
        printf(" w=%d %7.3f ms %.2f GHz  %.3f ns = %.2f cycles per instruction per SIMD; %.2f per cycle per CU\n",
               w, ms, ghz, ns_simd, ns_simd * ghz, 4.0 / (ns_simd * ghz));
        fflush(stdout);
    }
}

int main()
{
    unsigned long long *d; unsigned *sink;
    CHECK(hipMalloc(&d, 16 * 1024)); CHECK(hipMalloc(&sink, 4));
#define RUN(n, name, body) run<n, n>(d, sink);
    KINDS(RUN)
    return 0;
}
