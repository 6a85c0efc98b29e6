// ppp_patch_graph.hip -- S5: affinity of a pair of selected patches from the consensus.
//
// Reference: cuda/computePatchGraph.cu:3-136 (one thread per patch pair).  The float sum
// and the LCG thinning of the patch intersection are order dependent, so each pair is
// evaluated by one lane in the reference's loop order (bit-identical result).
#include "ppp_kernels.hpp"

namespace ppp {

template <typename T>
__global__ void __launch_bounds__(64)
    patch_graph_kernel(const T *__restrict__ pred, const float *__restrict__ cons,
                       const uint32_t *__restrict__ pairs, const uint64_t n,
                       float *__restrict__ aff, const Geo G) {
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    const int az = (int)pairs[id * 6 + 0], ay = (int)pairs[id * 6 + 1], ax = (int)pairs[id * 6 + 2];
    const int bz = (int)pairs[id * 6 + 3], by = (int)pairs[id * 6 + 4], bx = (int)pairs[id * 6 + 5];
    uint32_t rnd = (uint32_t)az * (uint32_t)bz * (uint32_t)ay * (uint32_t)by * (uint32_t)ax *
                   (uint32_t)bx;
    const long long ca = vox(G, az, ay, ax), cb = vox(G, bz, by, bx);
    const T *mid = pred + (long long)G.mid * G.V;
    float acc = 0.0f;
    unsigned fg_cnt = 0;
    int a = 0;
    for (int z1o = 0; z1o < G.pz; ++z1o)
        for (int y1o = 0; y1o < G.py; ++y1o)
            for (int x1o = 0; x1o < G.px; ++x1o, ++a) {
                const int z1 = az + z1o - G.rz, y1 = ay + y1o - G.ry, x1 = ax + x1o - G.rx;
                const long long u1 = vox(G, z1, y1, x1);
                if (!(ldf(mid, u1) > G.th_gt)) continue;
                if (!(ldf(pred, (long long)a * G.V + ca) > G.th_gt)) continue;
                const bool in_b = abs(x1 - bx) <= G.rx && abs(y1 - by) <= G.ry && abs(z1 - bz) <= G.rz;
                int b = 0;
                for (int z2o = 0; z2o < G.pz; ++z2o)
                    for (int y2o = 0; y2o < G.py; ++y2o)
                        for (int x2o = 0; x2o < G.px; ++x2o, ++b) {
                            const int z2 = bz + z2o - G.rz, y2 = by + y2o - G.ry,
                                      x2 = bx + x2o - G.rx;
                            const long long u2 = vox(G, z2, y2, x2);
                            if (!(ldf(mid, u2) > G.th_gt)) continue;
                            if (!(ldf(pred, (long long)b * G.V + cb) > G.th_gt)) continue;
                            if (in_b && abs(x2 - ax) <= G.rx && abs(y2 - ay) <= G.ry &&
                                abs(z2 - az) <= G.rz) {
                                rnd = rnd * 1103515245U;
                                const float rnd_t = (float)rnd / 4294967296.0f;
                                if ((double)rnd_t > 0.2) continue;
                            }
                            int dz, dy, dx, ez, ey, ex;  // offset, base voxel
                            if (u1 <= u2) {
                                dz = z2 - z1; dy = y2 - y1; dx = x2 - x1; ez = z1; ey = y1; ex = x1;
                            } else {
                                dz = z1 - z2; dy = y1 - y2; dx = x1 - x2; ez = z2; ey = y2; ex = x2;
                            }
                            // reference bound: 0 <= d + p - 1 < 2p  (allows d = +p)
                            if (dz < -(G.pz - 1) || dz > G.pz || dy < -(G.py - 1) || dy > G.py ||
                                dx < -(G.px - 1) || dx > G.px)
                                continue;
                            // planes with a component == +p, and the zero offset, are never
                            // written by S1: they read as 0 but still count
                            const bool stored = dz < G.pz && dy < G.py && dx < G.px &&
                                                (dz | dy | dx) != 0;
                            if (stored) acc += cons[cons_at(G, dz, dy, dx, ez, ey, ex)];
                            else acc += 0.0f;
                            fg_cnt += 1;
                        }
            }
    aff[id] = G.norm_aff ? acc / (float)(fg_cnt > 1u ? fg_cnt : 1u) : acc;
}

hipError_t launch_patch_graph(const void *pred, int dtype, const float *cons,
                              const uint32_t *pairs, uint64_t n, float *aff, const Geo &G,
                              hipStream_t s) {
    if (n == 0) return hipSuccess;
    const dim3 grid((unsigned)((n + 63) / 64));
    if (dtype == PPP_F16)
        patch_graph_kernel<__half><<<grid, dim3(64), 0, s>>>((const __half *)pred, cons, pairs, n, aff, G);
    else
        patch_graph_kernel<float><<<grid, dim3(64), 0, s>>>((const float *)pred, cons, pairs, n, aff, G);
    return hipGetLastError();
}

}  // namespace ppp
