// ppp_decode.hip -- ppp+dec: the TAIL of the patch decoder, fused with the scatter into the
// float16 prediction block that S1 reads.
//
// Reference: experiments/flylight/setups/setup01/torch_model.py:497-544 (Autoencoder.forward:
// from_code, then per stage [Upsample -> ConvPass], crop) and decode.py:43-65 (decode_sample: the
// decoded patch of every foreground voxel is scattered into output[:, z, y, x], one voxel at a
// time, in a float32 (C, Z, Y, X) array that is written as float16).  With the shipped decoder
// (default_train_code.toml [model.autoencoder]: num_fmaps [64, 128], 3^3 kernels, resize_conv
// upsampling x2 twice, 7^3 patches) the last stage is
//     U  = nearest-neighbour upsampling x2 of X            X: (F = 64, 4, 4, 4) per voxel
//     Y1 = relu(conv3(U; W1 [1][F][3][3][3]) + b1)         (1, 8, 8, 8)       up[1]
//     Y2 = conv3(Y1; W2) + b2,  Y3 = conv3(Y2; W3) + b3    no activation      up_conv[1]
//     patch = Y3[0:7, 0:7, 0:7]                            centre crop 8 -> 7 (offset 0)
// ("same" zero padding everywhere).  Everything before it -- the 1x1 from_code and the first stage,
// 28 of the decoder's 30 M multiply-adds per voxel, dense 64 / 128-channel convolutions at 4^3 --
// stays with the library GEMMs (MIOpen through torch); this kernel takes X and writes float16
// patch values straight into pred[r][dst voxel]: no (B, 1, 8, 8, 8) float32 intermediates, no
// float32 (C, Z, Y, X) array, no per-voxel scatter.
//
// The convolution over the UPSAMPLED image collapses: U[c][i] = X[c][i >> 1], so
//     Y1[o] = relu(b1 + sum_t Z[(o + t - 1) >> 1][t]),   Z[v][t] = sum_c X[c][v] * W1[c][t]
// -- a (64 voxels) x (64 channels) x (27 taps) product per patch instead of 512 x 64 x 27: 6.8
// times fewer multiply-adds.  Z is a GEMM and runs on the matrix cores in float32
// (v_mfma_f32_32x32x2_f32: exact f32 products, a k-ordered fma chain over the channels); the
// gathers of Y1 and the two single-channel convolutions are vector work on LDS.
//
// PARITY: the decoder's arithmetic is unpinned by the reference (funlib.learn.torch absent, no
// checkpoint); this kernel is checked against the torch restatement of the same layers
// (patchperpix_amd/decode.py PatchDecoder) within a float tolerance (summation order differs from
// MIOpen's), tests/test_decode.py.
#include "ppp_kernels.hpp"

namespace ppp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int DT_F = 64, DT_S = 4, DT_O = 2 * DT_S, DT_NV = DT_S * DT_S * DT_S;   // 64 source voxels
static constexpr int DT_NO = DT_O * DT_O * DT_O;                                       // 512 outputs
static constexpr int DT_WAVES = 4, DT_GROUP = 64;       // patches per workgroup (one output tile)
static constexpr int DT_ZS = 33;                        // padded tap stride of Z in LDS

template <typename T, int P>
__global__ void __launch_bounds__(64 * DT_WAVES)
    decode_tail_kernel(const float *__restrict__ X, const long long B, const float *__restrict__ W1,
                       const float b1, const float *__restrict__ W2, const float b2,
                       const float *__restrict__ W3, const float b3, const long long *__restrict__ dst,
                       T *__restrict__ pred, const long long V) {
    constexpr int C = P * P * P, OFF = (DT_O - P) / 2;
    // 102 KB of LDS (one workgroup per CU; the kernel is a sliver of the decoder's time):
    // dynamic, the static limit is 64 KB
    extern __shared__ float dt_lds[];
    float (*w1s)[32] = reinterpret_cast<float (*)[32]>(dt_lds);                        // [channel][tap], taps 27..31 zero
    float (*zs)[DT_NV * DT_ZS] = reinterpret_cast<float (*)[DT_NV * DT_ZS]>(dt_lds + DT_F * 32);
    float (*ya)[DT_NO] = reinterpret_cast<float (*)[DT_NO]>(dt_lds + DT_F * 32 + DT_WAVES * DT_NV * DT_ZS);
    float (*yb)[DT_NO] = ya + DT_WAVES;
    float (*w23)[27] = reinterpret_cast<float (*)[27]>(&yb[DT_WAVES][0]);
    __half (*tile)[DT_GROUP] = reinterpret_cast<__half (*)[DT_GROUP]>(&w23[2][2]);   // [patch value r][patch of the group]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < DT_F * 32; e += 64 * DT_WAVES) {
        const int c = e >> 5, t = e & 31;
        w1s[c][t] = t < 27 ? W1[c * 27 + t] : 0.0f;
    }
    if (tid < 27) { w23[0][tid] = W2[tid]; w23[1][tid] = W3[tid]; }
    __syncthreads();
    const long long g0 = (long long)blockIdx.x * DT_GROUP;
    float *z = zs[wave];
    for (int pi = wave; pi < DT_GROUP; pi += DT_WAVES) {
        const long long b = g0 + pi;
        if (b >= B) break;                          // (wave-uniform)
        const float *xb = X + b * (long long)(DT_F * DT_NV);
        // ---- Z = X^T W1 on the matrix cores: two 32-voxel tiles x 32 taps, k = channels
        f32x16 acc0 = {0}, acc1 = {0};
        const int r = lane & 31, h = lane >> 5;
#pragma unroll 8
        for (int ks = 0; ks < DT_F / 2; ++ks) {
            const int c = 2 * ks + h;
            const float bw = w1s[c][r];
            const float a0 = xb[c * DT_NV + r], a1 = xb[c * DT_NV + 32 + r];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bw, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bw, acc1, 0, 0, 0);
        }
        // C/D map: column (tap) = lane & 31, row (voxel) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int v = (q & 3) + 8 * (q >> 2) + 4 * h;
            z[v * DT_ZS + r] = acc0[q];
            z[(32 + v) * DT_ZS + r] = acc1[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- Y1 = relu(b1 + gathers of Z), then the two single-channel 3^3 convolutions
        float *y1 = ya[wave], *y2 = yb[wave];
#pragma unroll
        for (int i = 0; i < DT_NO / 64; ++i) {
            const int o = lane + 64 * i;
            const int oz = o / (DT_O * DT_O), oy = (o / DT_O) % DT_O, ox = o % DT_O;
            float s = b1;
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                const int iz = oz + t / 9 - 1, iy = oy + (t / 3) % 3 - 1, ix = ox + t % 3 - 1;
                if (iz >= 0 && iz < DT_O && iy >= 0 && iy < DT_O && ix >= 0 && ix < DT_O)
                    s += z[(((iz >> 1) * DT_S + (iy >> 1)) * DT_S + (ix >> 1)) * DT_ZS + t];
            }
            y1[o] = s > 0.0f ? s : 0.0f;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        auto conv1 = [&](const float *src, float *dstv, const float *w, const float bias, const bool last) {
#pragma unroll
            for (int i = 0; i < DT_NO / 64; ++i) {
                const int o = lane + 64 * i;
                const int oz = o / (DT_O * DT_O), oy = (o / DT_O) % DT_O, ox = o % DT_O;
                float s = bias;
#pragma unroll
                for (int t = 0; t < 27; ++t) {
                    const int iz = oz + t / 9 - 1, iy = oy + (t / 3) % 3 - 1, ix = ox + t % 3 - 1;
                    if (iz >= 0 && iz < DT_O && iy >= 0 && iy < DT_O && ix >= 0 && ix < DT_O)
                        s = __builtin_fmaf(w[t], src[(iz * DT_O + iy) * DT_O + ix], s);
                }
                if (!last) dstv[o] = s;
                else {
                    // centre crop and the float16 the prediction is stored in (decode.py:104-109)
                    const int pz = oz - OFF, py = oy - OFF, px = ox - OFF;
                    if (pz >= 0 && pz < P && py >= 0 && py < P && px >= 0 && px < P)
                        tile[(pz * P + py) * P + px][pi] = __float2half_rn(s);
                }
            }
        };
        conv1(y1, y2, w23[0], b2, false);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        conv1(y2, nullptr, w23[1], b3, true);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- the group's patches into pred[r][dst]: lanes = patches (consecutive foreground voxels
    // are consecutive addresses), one patch value r per wave and step
    const long long b = g0 + lane;
    const bool ok = b < B;
    const long long d = ok ? dst[b] : 0;
    for (int r = wave; r < C; r += DT_WAVES) {
        if (ok) {
            const __half hv = tile[r][lane];
            if constexpr (sizeof(T) == 2) pred[(long long)r * V + d] = hv;
            else pred[(long long)r * V + d] = __half2float(hv);
        }
    }
}

hipError_t launch_decode_tail(const float *X, long long B, int F, int S, const float *W1, float b1,
                              const float *W2, float b2, const float *W3, float b3, const long long *dst,
                              void *pred, int dtype, const Geo &G, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (F != DT_F || S != DT_S || G.pz != 7 || G.py != 7 || G.px != 7) return hipErrorNotSupported;
    const long long groups = (B + DT_GROUP - 1) / DT_GROUP;
    PPP_GRID_CHECK(groups, 64 * DT_WAVES);
    const size_t lds = (size_t)(DT_F * 32 + DT_WAVES * DT_NV * DT_ZS + 2 * DT_WAVES * DT_NO + 2 * 27 + 2) * 4 +
                       (size_t)343 * DT_GROUP * 2;
    hipError_t e;
    if (dtype == PPP_F16) {
        if ((e = hipFuncSetAttribute((const void *)decode_tail_kernel<__half, 7>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        decode_tail_kernel<__half, 7><<<dim3((unsigned)groups), dim3(64 * DT_WAVES), lds, s>>>(
            X, B, W1, b1, W2, b2, W3, b3, dst, (__half *)pred, G.V);
    } else {
        if ((e = hipFuncSetAttribute((const void *)decode_tail_kernel<float, 7>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        decode_tail_kernel<float, 7><<<dim3((unsigned)groups), dim3(64 * DT_WAVES), lds, s>>>(
            X, B, W1, b1, W2, b2, W3, b3, dst, (float *)pred, G.V);
    }
    return hipGetLastError();
}

}  // namespace ppp
