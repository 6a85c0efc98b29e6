// issue_mix.hip -- does a SIMD issue instructions of DIFFERENT types (VALU / LDS / SALU / s_waitcnt)
// from its waves in parallel, or does every instruction cost an issue slot of its own?  gfx950, 4 waves
// per SIMD (1024 threads per block, one block per CU), every stream independent.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define BODY_BEGIN(NAME)                                                                        \
    __global__ void __launch_bounds__(1024) NAME(float *out, int iters) {                       \
        __shared__ float lds[4096];                                                             \
        lds[threadIdx.x] = (float)threadIdx.x; lds[threadIdx.x + 1024] = 1.0f;                   \
        __syncthreads();                                                                        \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, r0 = 0, r1 = 0, r2 = 0, r3 = 0; \
        unsigned c = 0x28002800u, addr = (threadIdx.x & 63) * 4;                                \
        for (int it = 0; it < iters; ++it) {
#define BODY_END                                                                                \
        }                                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + r0 + r1 + r2 + r3;      \
    }
#define MIX4 "v_fma_mix_f32 %0, %4, %8, %0 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %1, %5, %8, %1 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n v_fma_mix_f32 %2, %6, %8, %2 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %3, %7, %8, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n"
#define FMA4 "v_fma_f32 %0, %4, %8, %0\n v_fma_f32 %1, %5, %8, %1\n v_fma_f32 %2, %6, %8, %2\n v_fma_f32 %3, %7, %8, %3\n"
#define FMAC4 "v_fmac_f32 %0, %4, %8\n v_fmac_f32 %1, %5, %8\n v_fmac_f32 %2, %6, %8\n v_fmac_f32 %3, %7, %8\n"
#define DS4 "ds_read_b32 %4, %9\n ds_read_b32 %5, %9 offset:256\n ds_read_b32 %6, %9 offset:512\n ds_read_b32 %7, %9 offset:768\n"
#define NOP4 "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
#define WAIT4 "s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n"
#define W1 "s_waitcnt lgkmcnt(0)\n"
#define ASM(S) asm volatile(S : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c), "v"(addr) : "memory");

BODY_BEGIN(k_mix) REP16(ASM(MIX4)) BODY_END
BODY_BEGIN(k_fma) REP16(ASM(FMA4)) BODY_END
BODY_BEGIN(k_fmac) REP16(ASM(FMAC4)) BODY_END
BODY_BEGIN(k_ds) REP16(ASM(DS4 W1)) BODY_END
BODY_BEGIN(k_mix_ds) REP16(ASM(DS4 W1 MIX4)) BODY_END
BODY_BEGIN(k_mix_nop) REP16(ASM(NOP4 MIX4)) BODY_END
BODY_BEGIN(k_mix_wait) REP16(ASM(WAIT4 MIX4)) BODY_END
BODY_BEGIN(k_fmac_ds) REP16(ASM(DS4 W1 FMAC4)) BODY_END

typedef void (*kern_t)(float *, int);
static void run(const char *name, kern_t k, float *out, int per_body) {
    const int iters = 500, wps = 4, threads = 1024, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<blocks, threads>>>(out, 10);
    (void)hipEventRecord(e0);
    k<<<blocks, threads>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double groups = (double)iters * 16 * wps;     // groups per SIMD
    printf("%-12s %6.2f cycles per group of %d instructions per SIMD (4 waves/SIMD, 2.4 GHz assumed)\n", name,
           ms * 1e6 / groups * 2.4, per_body);
}
int main(int argc, char **argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    float *out; (void)hipMalloc(&out, 256 * 1024 * 4);
    if (argc > 1) {      // one kernel by name (a hang is then attributable)
        struct { const char *n; kern_t k; int b; } all[] = {{"mix", k_mix, 4}, {"fma", k_fma, 4}, {"fmac", k_fmac, 4},
            {"ds+wait", k_ds, 5}, {"mix+ds+wait", k_mix_ds, 9}, {"mix+nop", k_mix_nop, 8},
            {"mix+4wait", k_mix_wait, 8}, {"fmac+ds+wait", k_fmac_ds, 9}};
        for (auto &e : all) if (!strcmp(e.n, argv[1])) run(e.n, e.k, out, e.b);
        return 0;
    }
    run("mix", k_mix, out, 4); run("fma", k_fma, out, 4); run("fmac", k_fmac, out, 4);
    run("ds+wait", k_ds, out, 5);
    run("mix+ds+wait", k_mix_ds, out, 9);
    run("mix+nop", k_mix_nop, out, 8); run("mix+4wait", k_mix_wait, out, 8);
    run("fmac+ds+wait", k_fmac_ds, out, 9);
    return 0;
}
