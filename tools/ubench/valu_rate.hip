// valu_rate.hip -- issue-rate microbenchmark for gfx950 (MI355X): wave-instructions per cycle per
// SIMD for the instruction kinds the S1/S2/S5 kernels are made of, at 1..8 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void __launch_bounds__(1024) k(float *out, int iters, long long *cyc) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    float b = 1.0001f, c = 0.5f;
    double db = 1.0001;
    int m = threadIdx.x;
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    __syncthreads();
    unsigned addr = (KIND == 11 ? (threadIdx.x * 2654435761u >> 20) & 4095u : threadIdx.x & 4095u) * 4u;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (KIND == 1) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(db));) }
        if (KIND == 2) { REP8(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db));) }
        if (KIND == 3) { REP8(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db));) }
        if (KIND == 4) { REP8(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
        if (KIND == 5) { REP8(asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));) }
        if (KIND == 6) { REP8(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));) }
        if (KIND == 7) { REP8(asm volatile("v_bfe_i32 %0, %8, 3, 1\n v_bfe_i32 %1, %8, 4, 1\n v_bfe_i32 %2, %8, 5, 1\n v_bfe_i32 %3, %8, 6, 1\n v_bfe_i32 %4, %8, 7, 1\n v_bfe_i32 %5, %8, 8, 1\n v_bfe_i32 %6, %8, 9, 1\n v_bfe_i32 %7, %8, 10, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));) }
        if (KIND == 8) { REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (KIND == 9) { REP8(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(b));) }
        if (KIND == 10 || KIND == 11) { REP8(asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(addr));) }
        if (KIND == 12) { REP8(asm volatile("v_cmp_neq_f32 vcc, 0, %0\n v_cndmask_b32 %1, 0, %2, vcc\n v_cmp_neq_f32 vcc, 0, %3\n v_cndmask_b32 %4, 0, %5, vcc\n v_cmp_neq_f32 vcc, 0, %0\n v_cndmask_b32 %6, 0, %2, vcc\n v_cmp_neq_f32 vcc, 0, %3\n v_cndmask_b32 %7, 0, %5, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");) }
        if (KIND == 13) { REP8(asm volatile("v_max_i32 %0, %0, %8\n v_bfi_b32 %1, %8, %1, %2\n v_max_i32 %2, %2, %8\n v_bfi_b32 %3, %8, %3, %4\n v_max_i32 %4, %4, %8\n v_bfi_b32 %5, %8, %5, %6\n v_max_i32 %6, %6, %8\n v_bfi_b32 %7, %8, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));) }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
static void run(const char *name, float *out, long long *cyc) {
    const int iters = 2000;
    for (int wps = 1; wps <= 8; wps *= 2) {   // waves per SIMD: block = wps*4 waves, one block per CU
        const int threads = 64 * 4 * wps > 1024 ? 1024 : 64 * 4 * wps;
        const int blocks = 256 * (64 * 4 * wps / threads);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<blocks, threads>>>(out, 10, cyc);
        hipEventRecord(e0);
        k<KIND><<<blocks, threads>>>(out, iters, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double inst_per_wave = (double)iters * 64;
        // wall-clock based: wave-instructions per SIMD per ns
        const double per_simd = inst_per_wave * wps;
        printf("%-22s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz); clock64 delta %lld\n",
               name, wps, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, c);
    }
}

int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 8 * 1024 * 4); hipMalloc(&cyc, 8);
    run<0>("v_fma_f32", out, cyc);
    run<8>("v_add_f32", out, cyc);
    run<9>("v_add_f32 dependent", out, cyc);
    run<1>("v_pk_fma_f32", out, cyc);
    run<2>("v_add_f64", out, cyc);
    run<3>("v_mul_f64", out, cyc);
    run<4>("v_cvt_f64_f32", out, cyc);
    run<5>("v_cvt_f32_f64", out, cyc);
    run<6>("v_and_b32", out, cyc);
    run<7>("v_bfe_i32", out, cyc);
    run<12>("v_cmp+v_cndmask", out, cyc);
    run<13>("v_max_i32+v_bfi", out, cyc);
    run<10>("ds_read_b32 linear", out, cyc);
    run<11>("ds_read_b32 random", out, cyc);
    return 0;
}
