// valu_rate2.hip -- per-instruction issue cost on gfx950, 4 waves per SIMD (see valu_rate.hip).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
// one asm statement = 8 independent instances of the instruction (operands %0..%7 in/out,
// %8 %9 extra vector inputs)
#define DEFK(NAME, ASM8)                                                                       \
    __global__ void __launch_bounds__(1024) NAME(float *out, int iters) {                      \
        unsigned a0 = threadIdx.x * 3 + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4,  \
                 a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = a0 | 0x41C64E6Du, c = threadIdx.x & 31; \
        for (int it = 0; it < iters; ++it) {                                                   \
            REP8(asm volatile(ASM8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), \
                              "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s10", "s11");)      \
        }                                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;     \
    }
#define I8(op, fmt) op " %0, " fmt(0) "\n" op " %1, " fmt(1) "\n" op " %2, " fmt(2) "\n" op " %3, " fmt(3) "\n" \
                    op " %4, " fmt(4) "\n" op " %5, " fmt(5) "\n" op " %6, " fmt(6) "\n" op " %7, " fmt(7)
#define F2(i) "%" #i ", %8"
#define F2r(i) "%8, %" #i
#define F3(i) "%" #i ", %8, %9"
#define F1(i) "%" #i
DEFK(k_mul_lo, I8("v_mul_lo_u32", F2))
DEFK(k_mul_u24, I8("v_mul_u32_u24", F2))
DEFK(k_mad_u24, I8("v_mad_u32_u24", F3))
DEFK(k_mul_hi, I8("v_mul_hi_u32", F2))
DEFK(k_lshl32, I8("v_lshlrev_b32", F2r))
DEFK(k_max_i32, I8("v_max_i32", F2))
DEFK(k_bfi, I8("v_bfi_b32", F3))
DEFK(k_bfe_u, "v_bfe_u32 %0, %0, 3, 5\n v_bfe_u32 %1, %1, 3, 5\n v_bfe_u32 %2, %2, 3, 5\n v_bfe_u32 %3, %3, 3, 5\n v_bfe_u32 %4, %4, 3, 5\n v_bfe_u32 %5, %5, 3, 5\n v_bfe_u32 %6, %6, 3, 5\n v_bfe_u32 %7, %7, 3, 5")
DEFK(k_or3, I8("v_or3_b32", F3))
DEFK(k_and_or, I8("v_and_or_b32", F3))
DEFK(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_lshl_add_u32 %2, %2, 2, %8\n v_lshl_add_u32 %3, %3, 2, %8\n v_lshl_add_u32 %4, %4, 2, %8\n v_lshl_add_u32 %5, %5, 2, %8\n v_lshl_add_u32 %6, %6, 2, %8\n v_lshl_add_u32 %7, %7, 2, %8")
DEFK(k_add3, I8("v_add3_u32", F3))
DEFK(k_add_u32, I8("v_add_u32", F2))
DEFK(k_bcnt, I8("v_bcnt_u32_b32", F2))
DEFK(k_cvt_ubyte, "v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte1 %1, %8\n v_cvt_f32_ubyte2 %2, %8\n v_cvt_f32_ubyte3 %3, %8\n v_cvt_f32_ubyte0 %4, %9\n v_cvt_f32_ubyte1 %5, %9\n v_cvt_f32_ubyte2 %6, %9\n v_cvt_f32_ubyte3 %7, %9")
DEFK(k_mov, I8("v_mov_b32", F1))
DEFK(k_cmp_vcc, "v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %8\n v_cmp_lt_u32 vcc, %3, %8\n v_cmp_lt_u32 vcc, %4, %8\n v_cmp_lt_u32 vcc, %5, %8\n v_cmp_lt_u32 vcc, %6, %8\n v_cmp_lt_u32 vcc, %7, %8")
DEFK(k_cmp_sgpr, "v_cmp_lt_u32 s[10:11], %0, %8\n v_cmp_lt_u32 s[10:11], %1, %8\n v_cmp_lt_u32 s[10:11], %2, %8\n v_cmp_lt_u32 s[10:11], %3, %8\n v_cmp_lt_u32 s[10:11], %4, %8\n v_cmp_lt_u32 s[10:11], %5, %8\n v_cmp_lt_u32 s[10:11], %6, %8\n v_cmp_lt_u32 s[10:11], %7, %8")
DEFK(k_cndmask_vcc, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc")
DEFK(k_cndmask_sgpr, "v_cndmask_b32 %0, %0, %8, s[10:11]\n v_cndmask_b32 %1, %1, %8, s[10:11]\n v_cndmask_b32 %2, %2, %8, s[10:11]\n v_cndmask_b32 %3, %3, %8, s[10:11]\n v_cndmask_b32 %4, %4, %8, s[10:11]\n v_cndmask_b32 %5, %5, %8, s[10:11]\n v_cndmask_b32 %6, %6, %8, s[10:11]\n v_cndmask_b32 %7, %7, %8, s[10:11]")
DEFK(k_addc, "v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n v_addc_co_u32 %3, vcc, 0, %3, vcc\n v_addc_co_u32 %4, vcc, 0, %4, vcc\n v_addc_co_u32 %5, vcc, 0, %5, vcc\n v_addc_co_u32 %6, vcc, 0, %6, vcc\n v_addc_co_u32 %7, vcc, 0, %7, vcc")
DEFK(k_and_sdwa, "v_and_b32_sdwa %0, %0, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_and_b32_sdwa %1, %1, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_and_b32_sdwa %2, %2, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_and_b32_sdwa %3, %3, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_and_b32_sdwa %4, %4, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_and_b32_sdwa %5, %5, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_and_b32_sdwa %6, %6, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_and_b32_sdwa %7, %7, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3")
DEFK(k_mul_f32, I8("v_mul_f32", F2))
DEFK(k_sub_f32, I8("v_sub_f32", F2))
DEFK(k_fmac, I8("v_fmac_f32", F2r))
DEFK(k_ashr, "v_ashrrev_i32 %0, 31, %0\n v_ashrrev_i32 %1, 31, %1\n v_ashrrev_i32 %2, 31, %2\n v_ashrrev_i32 %3, 31, %3\n v_ashrrev_i32 %4, 31, %4\n v_ashrrev_i32 %5, 31, %5\n v_ashrrev_i32 %6, 31, %6\n v_ashrrev_i32 %7, 31, %7")
DEFK(k_perm, I8("v_perm_b32", F3))
DEFK(k_cvt_f16, I8("v_cvt_f32_f16", F1))

#define DEFK64(NAME, ASM4)                                                                     \
    __global__ void __launch_bounds__(1024) NAME(float *out, int iters) {                      \
        unsigned long long a0 = threadIdx.x * 3 + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;     \
        unsigned c = threadIdx.x & 31;                                                         \
        for (int it = 0; it < iters; ++it) {                                                   \
            REP8(asm volatile(ASM4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)         \
        }                                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3);                \
    }
DEFK64(k_lshr64, "v_lshrrev_b64 %0, %4, %0\n v_lshrrev_b64 %1, %4, %1\n v_lshrrev_b64 %2, %4, %2\n v_lshrrev_b64 %3, %4, %3\n v_lshrrev_b64 %0, %4, %0\n v_lshrrev_b64 %1, %4, %1\n v_lshrrev_b64 %2, %4, %2\n v_lshrrev_b64 %3, %4, %3")
DEFK64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %1\n v_lshl_add_u64 %1, %1, 1, %2\n v_lshl_add_u64 %2, %2, 1, %3\n v_lshl_add_u64 %3, %3, 1, %0\n v_lshl_add_u64 %0, %0, 1, %1\n v_lshl_add_u64 %1, %1, 1, %2\n v_lshl_add_u64 %2, %2, 1, %3\n v_lshl_add_u64 %3, %3, 1, %0")

// packed float32 (operands are aligned register pairs): S1's vote chain
#define DEFKPK(NAME, ASM8)                                                                     \
    __global__ void __launch_bounds__(1024) NAME(float *out, int iters) {                      \
        double a0 = threadIdx.x * 3 + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.5, c = 0.25; \
        for (int it = 0; it < iters; ++it) {                                                   \
            REP8(asm volatile(ASM8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)  \
        }                                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3);                \
    }
DEFKPK(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5")
DEFKPK(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4")
DEFKPK(k_pk_add_f32, "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4")
DEFKPK(k_pk_add_clamp, "v_pk_add_f32 %0, %0, %4 clamp\n v_pk_add_f32 %1, %1, %4 clamp\n v_pk_add_f32 %2, %2, %4 clamp\n v_pk_add_f32 %3, %3, %4 clamp\n v_pk_add_f32 %0, %0, %4 clamp\n v_pk_add_f32 %1, %1, %4 clamp\n v_pk_add_f32 %2, %2, %4 clamp\n v_pk_add_f32 %3, %3, %4 clamp")
DEFK(k_fma_f32, I8("v_fma_f32", F3))
DEFK(k_fma_mix, "v_fma_mix_f32 %0, %0, %8, %9 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %1, %1, %8, %9 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %2, %2, %8, %9 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %3, %3, %8, %9 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %4, %4, %8, %9 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %5, %5, %8, %9 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %6, %6, %8, %9 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %7, %7, %8, %9 op_sel_hi:[0,1,0]")
DEFK(k_pk_mad_u16, I8("v_pk_mad_u16", F3))

typedef void (*kern_t)(float *, int);
static void run(const char *name, kern_t k, float *out, bool pair) {
    const int iters = 2000, wps = 4, threads = 1024, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<blocks, threads>>>(out, 10);
    (void)hipEventRecord(e0);
    k<<<blocks, threads>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * 64 * wps;
    printf("%-18s %.2f cycles per wave-instr per SIMD (4 waves/SIMD, 2.4 GHz assumed)\n", name, ms * 1e6 / per_simd * 2.4);
}
#define RUN(n) run(#n, n, out, false)
int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    float *out; (void)hipMalloc(&out, 256 * 1024 * 4);
    RUN(k_mov); RUN(k_add_u32); RUN(k_mul_f32); RUN(k_sub_f32); RUN(k_fmac); RUN(k_lshl32); RUN(k_ashr);
    RUN(k_max_i32); RUN(k_bfi); RUN(k_bfe_u); RUN(k_or3); RUN(k_and_or); RUN(k_lshl_add); RUN(k_add3);
    RUN(k_bcnt); RUN(k_perm); RUN(k_cvt_ubyte); RUN(k_cvt_f16); RUN(k_and_sdwa);
    RUN(k_cmp_vcc); RUN(k_cmp_sgpr); RUN(k_cndmask_vcc); RUN(k_cndmask_sgpr); RUN(k_addc);
    RUN(k_fma_f32); RUN(k_fma_mix); RUN(k_pk_fma_f32); RUN(k_pk_mul_f32); RUN(k_pk_add_f32); RUN(k_pk_add_clamp); RUN(k_pk_mad_u16);
    RUN(k_mul_lo); RUN(k_mul_hi); RUN(k_mul_u24); RUN(k_mad_u24); RUN(k_lshr64); RUN(k_lshl_add_u64);
    return 0;
}
