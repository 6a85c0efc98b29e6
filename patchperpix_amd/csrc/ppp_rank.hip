// ppp_rank.hip -- S2: per-patch agreement score against the consensus.
//
// Reference: cuda/rankPatches.cu:1-161.  One thread per patch centre; the float
// accumulation runs in the reference's loop order (r1 raster, r2 raster), so the score is
// bit-identical given the same consensus.  Lanes of a wave are consecutive centres along
// x, so every consensus gather (same plane, consecutive base voxels) is coalesced.
#include <cstdlib>

#include <stdlib.h>

#include "ppp_kernels.hpp"

namespace ppp {

template <typename T>
__global__ void __launch_bounds__(256)
    rank_kernel(const T *__restrict__ pred, const float *__restrict__ cons,
                const uint8_t *__restrict__ ov, float *__restrict__ score, const ppp_box sb,
                const Geo G) {
    const int sX = sb.x1 - sb.x0, sY = sb.y1 - sb.y0, sZ = sb.z1 - sb.z0;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)sX * sY * sZ) return;
    const int cx = sb.x0 + (int)(i % sX);
    const long long t = i / sX;
    const int cy = sb.y0 + (int)(t % sY);
    const int cz = sb.z0 + (int)(t / sY);
    const long long lc = vox(G, cz, cy, cx);
    if (!interior(G, cz, cy, cx)) {
        score[lc] = G.norm_rank ? -1.0f : -9999999.0f;
        return;
    }
    const T *mid = pred + (long long)G.mid * G.V;
    if (!(ldf(mid, lc) > G.th_gt)) {
        score[lc] = 0.0f;  // the reference leaves the zero from allocation
        return;
    }
    float acc = 0.0f;
    unsigned fg_cnt = 0;
    int a = 0;
    for (int z1o = 0; z1o < G.pz; ++z1o)
        for (int y1o = 0; y1o < G.py; ++y1o)
            for (int x1o = 0; x1o < G.px; ++x1o, ++a) {
                const float v1 = ldf(pred, (long long)a * G.V + lc);
                if (!(v1 > G.th_gt)) continue;
                const int z1 = cz + z1o - G.rz, y1 = cy + y1o - G.ry, x1 = cx + x1o - G.rx;
                const long long u1 = vox(G, z1, y1, x1);
                if (!(ldf(mid, u1) > G.th_gt)) continue;
                if (G.use_overlap && ov[u1] != 0) continue;
                int b = 0;
                for (int z2o = 0; z2o < G.pz; ++z2o)
                    for (int y2o = 0; y2o < G.py; ++y2o)
                        for (int x2o = 0; x2o < G.px; ++x2o, ++b) {
                            if (a == b) continue;
                            const int z2 = cz + z2o - G.rz, y2 = cy + y2o - G.ry,
                                      x2 = cx + x2o - G.rx;
                            const long long u2 = vox(G, z2, y2, x2);
                            if (!(ldf(mid, u2) > G.th_gt)) continue;
                            if (G.use_overlap && ov[u2] != 0) continue;
                            const float v2 = ldf(pred, (long long)b * G.V + lc);
                            if (v2 > G.th_gt) {
                                if (b <= a) continue;
                                const float v3 =
                                    cons[cons_at(G, z2o - z1o, y2o - y1o, x2o - x1o, z1, y1, x1)];
                                if (G.count_pos_neg) acc += (v3 != 0.0f) ? copysignf(1.0f, v3) : -1.0f;
                                else acc += v3;
                            } else if (v2 < G.bg_lt) {
                                const float v3 =
                                    (b <= a)
                                        ? cons[cons_at(G, z1o - z2o, y1o - y2o, x1o - x2o, z2, y2, x2)]
                                        : cons[cons_at(G, z2o - z1o, y2o - y1o, x2o - x1o, z1, y1, x1)];
                                if (G.count_pos_neg) acc -= (v3 != 0.0f) ? copysignf(1.0f, v3) : 1.0f;
                                else acc -= v3;
                            }
                            fg_cnt += 1;
                        }
            }
    score[lc] = G.norm_rank ? acc / (float)(fg_cnt > 1u ? fg_cnt : 1u) : acc;
}

hipError_t launch_rank(const void *pred, int dtype, const float *cons, const uint8_t *ov,
                       float *score, const ppp_box &sb, const Geo &G, hipStream_t s) {
    const long long n = (long long)(sb.x1 - sb.x0) * (sb.y1 - sb.y0) * (sb.z1 - sb.z0);
    if (n <= 0) return hipSuccess;
    // specialised kernel for px in {3,5,7,9}; PPP_RANK_GENERIC=1 forces the generic one
    static EnvSwitch generic_sw("PPP_RANK_GENERIC");
    const bool force_generic = generic_sw.get() != nullptr;
    if (!force_generic) {
        const hipError_t e2 = launch_rank_v2(pred, dtype, cons, ov, score, sb, G, s);
        if (e2 != hipErrorNotSupported) return e2;
    }
    PPP_GRID_CHECK((n + 255) / 256, 256);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == PPP_F16)
        rank_kernel<__half><<<grid, dim3(256), 0, s>>>((const __half *)pred, cons, ov, score, sb, G);
    else
        rank_kernel<float><<<grid, dim3(256), 0, s>>>((const float *)pred, cons, ov, score, sb, G);
    return hipGetLastError();
}

}  // namespace ppp
