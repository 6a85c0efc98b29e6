// cndmask.hip -- compare + select through VCC (VOP2 v_cndmask_b32_e32) against compare + select through
// an SGPR pair (VOP3 v_cndmask_b32_e64) on gfx950, 4 waves per SIMD, independent streams.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#define REP4(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
#define REP16(...) REP4(REP4(__VA_ARGS__))
#define KERNEL(NAME, ...)                                                                          \
    __global__ void __launch_bounds__(1024) NAME(float *out, int iters) {                          \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 100.5f, c = 2.0f;         \
        for (int it = 0; it < iters; ++it) { REP16(__VA_ARGS__) }                                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;                             \
    }
// four compare + select pairs through VCC (the wait state the compiler puts between them included)
#define VCC1(A) "v_cmp_gt_f32 vcc, " A ", %4\n s_nop 1\n v_cndmask_b32 " A ", " A ", %5, vcc\n"
KERNEL(k_vcc, asm volatile(VCC1("%0") VCC1("%1") VCC1("%2") VCC1("%3") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");)
// the same through SGPR pairs
#define SG1(A, S) "v_cmp_gt_f32 " S ", " A ", %8\n s_nop 1\n v_cndmask_b32 " A ", " A ", %9, " S "\n"
KERNEL(k_sgpr, { unsigned long long m0, m1, m2, m3; asm volatile(SG1("%0", "%4") SG1("%1", "%5") SG1("%2", "%6") SG1("%3", "%7") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3) : "v"(b), "v"(c)); })
// compares only / selects only
KERNEL(k_cmp_vcc, asm volatile("v_cmp_gt_f32 vcc, %0, %4\n v_cmp_gt_f32 vcc, %1, %4\n v_cmp_gt_f32 vcc, %2, %4\n v_cmp_gt_f32 vcc, %3, %4\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");)
KERNEL(k_sel_vcc, asm volatile("v_cndmask_b32 %0, %0, %5, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %5, vcc\n v_cndmask_b32 %3, %3, %5, vcc\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");)
// the clamp form S1's chain uses instead of a select: max(min(x * big, 1), 0)-style indicator
KERNEL(k_mul_clamp, asm volatile("v_mul_f32 %0, %0, %5 clamp\n v_mul_f32 %1, %1, %5 clamp\n v_mul_f32 %2, %2, %5 clamp\n v_mul_f32 %3, %3, %5 clamp\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)

typedef void (*kern_t)(float *, int);
static void run(const char *name, kern_t k, float *out, const char *what) {
    const int iters = 500, wps = 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<256, 1024>>>(out, 10);
    (void)hipEventRecord(e0);
    k<<<256, 1024>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-12s %6.2f cycles per %s per SIMD (4 waves/SIMD, 2.4 GHz assumed)\n", name, ms * 1e6 / ((double)iters * 16 * 4 * wps) * 2.4, what);
}
int main(int argc, char **argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    float *out; (void)hipMalloc(&out, 256 * 1024 * 4);
    struct { const char *n; kern_t k; const char *w; } all[] = {{"cmp+sel vcc", k_vcc, "compare + select"}, {"cmp+sel sgpr", k_sgpr, "compare + select"},
        {"cmp vcc", k_cmp_vcc, "compare"}, {"sel vcc", k_sel_vcc, "select"}, {"mul clamp", k_mul_clamp, "v_mul_f32 clamp"}};
    for (auto &e : all) if (argc < 2 || !strcmp(e.n, argv[1])) run(e.n, e.k, out, e.w);
    return 0;
}
