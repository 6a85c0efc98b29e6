// ppp_consensus.hip -- S1: patch-pair scoring + consensus vote, GATHER form.
//
// Reference: cuda/fillConsensusArray.cu:5-218 (scatter with float atomicAdd, run twice
// for value and count) + cuda/normConsensusArray.cu:5-43.
//
// The reference scatters: one thread per patch centre c adds a vote to key
// (offset d, earlier voxel u) for every ordered pixel pair of its patch.  For a fixed
// key, a centre contributes AT MOST ONE vote (the key fixes the unordered voxel pair
// {u, w = u + d}; the pos / neg cases on (pred[r_u][c], pred[r_w][c]) are mutually
// exclusive).  So
//
//     cons[d][u] = sum over centres c with u, w in win(c) of  vote(c, u, w)
//
// and this kernel computes it directly: one thread per key, looping over the centres in
// RASTER ORDER of c -- a legal serialisation of the reference's atomics, no atomics, no
// zero-fill, count and normalisation fused, bit-reproducible.
//
// Centre c = u - k + rad where k in [0,p)^3 is the patch offset of u, and k + d the
// patch offset of w; raster-ascending c  <=>  lexicographically descending k.
#include <cstdlib>

#include "ppp_kernels.hpp"

namespace ppp {

template <typename T>
__global__ void __launch_bounds__(256)
    consensus_gather_kernel(const T *__restrict__ pred, const uint8_t *__restrict__ ov,
                            float *__restrict__ cons, float *__restrict__ cnt_out, const Geo G) {
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= G.BV) return;
    // plane -> offset d (signed mixed radix, see ppp_mi355x.h)
    const int L = (int)blockIdx.y + 1;
    const int Ls = L + (G.py - 1) * G.wx + (G.px - 1);
    const int dx = Ls % G.wx - (G.px - 1);
    const int q = Ls / G.wx;
    const int dy = q % G.wy - (G.py - 1);
    const int dz = q / G.wy;
    // base voxel u (global coords) and partner w
    const int ux = G.bx0 + (int)(b % G.bX);
    const long long t = b / G.bX;
    const int uy = G.by0 + (int)(t % G.bY);
    const int uz = G.bz0 + (int)(t / G.bY);
    const int wz = uz + dz, wy = uy + dy, wx = ux + dx;

    float acc = 0.0f, cnt = 0.0f;
    const T *mid = pred + (long long)G.mid * G.V;
    bool ok = wz < G.Z && wy >= 0 && wy < G.Y && wx >= 0 && wx < G.X;
    long long lu = vox(G, uz, uy, ux), lw = 0;
    if (ok) {
        lw = vox(G, wz, wy, wx);
        ok = ldf(mid, lu) > G.th_gt && ldf(mid, lw) > G.th_gt;
        if (ok && G.use_overlap) ok = ov[lu] == 0 && ov[lw] == 0;
    }
    if (ok) {
        const int kz_hi = min(G.pz - 1, G.pz - 1 - dz), kz_lo = max(0, -dz);
        const int ky_hi = min(G.py - 1, G.py - 1 - dy), ky_lo = max(0, -dy);
        const int kx_hi = min(G.px - 1, G.px - 1 - dx), kx_lo = max(0, -dx);
        for (int kz = kz_hi; kz >= kz_lo; --kz) {
            const int cz = uz - kz + G.rz;
            if (cz < G.rz || cz >= G.Z - G.rz) continue;
            for (int ky = ky_hi; ky >= ky_lo; --ky) {
                const int cy = uy - ky + G.ry;
                if (cy < G.ry || cy >= G.Y - G.ry) continue;
                for (int kx = kx_hi; kx >= kx_lo; --kx) {
                    const int cx = ux - kx + G.rx;
                    if (cx < G.rx || cx >= G.X - G.rx) continue;
                    const long long lc = vox(G, cz, cy, cx);
                    if (!(ldf(mid, lc) > G.th_gt)) continue;
                    const int ru = (kz * G.py + ky) * G.px + kx;
                    const int rw = ((kz + dz) * G.py + (ky + dy)) * G.px + (kx + dx);
                    const float a = ldf(pred, (long long)ru * G.V + lc);  // about u
                    const float bb = ldf(pred, (long long)rw * G.V + lc); // about w
                    if (a > G.th_gt) {
                        if (bb > G.th_gt) {
                            acc = acc + vote_value(G, a * bb);
                            cnt = cnt + 1.0f;
                        } else if (bb < G.bg_lt) {
                            acc = acc + (-vote_value(G, a * (1.0f - bb)));
                            cnt = cnt + 1.0f;
                        }
                    } else if (bb > G.th_gt && a < G.bg_lt) {
                        acc = acc + (-vote_value(G, bb * (1.0f - a)));
                        cnt = cnt + 1.0f;
                    }
                }
            }
        }
    }
    const long long o = cons_at(G, dz, dy, dx, uz, uy, ux);
    if (cons) cons[o] = (G.normalise && cnt != 0.0f) ? acc / cnt : acc;
    if (cnt_out) cnt_out[o] = cnt;
}

static const char *g_s1_kernel = "none";
const char *last_consensus_kernel() { return g_s1_kernel; }

hipError_t launch_consensus(const void *pred, int dtype, const uint8_t *ov, float *cons,
                            float *cnt, const Geo &G, hipStream_t s) {
    if (G.layout == PPP_CONS_VOXEL_MAJOR) {
        // only the packed two-slice kernel writes the voxel-major rows directly
        if (!consensus_v3_supported(G)) return hipErrorNotSupported;
#ifdef PPP_BUILD_EXPERIMENTS       // (the two-wave kernel: PPP_BUILD_EXPERIMENTS=1 builds, PPP_S1_V4=1 only)
        if (consensus_v4_supported(G)) {
            g_s1_kernel = "consensus_v4_kernel";
            return launch_consensus_v4(pred, dtype, ov, cons, cnt, G, s);
        }
#endif
        g_s1_kernel = "consensus_v3_kernel";
        return launch_consensus_v3(pred, dtype, ov, cons, cnt, G, s);
    }
    const dim3 block(256);
    PPP_GRID_CHECK((G.BV + 255) / 256, 256);
    const dim3 grid((unsigned)((G.BV + 255) / 256), (unsigned)G.n_planes);
    if (G.layout == PPP_CONS_REFERENCE) {
        // planes the reference never writes stay zero
        const size_t bytes = (size_t)(G.pz > 1 ? 2 * G.pz : 1) * G.nsy * G.nsx * G.V * sizeof(float);
        hipError_t e;
        if (cons && (e = hipMemsetAsync(cons, 0, bytes, s)) != hipSuccess) return e;
        if (cnt && (e = hipMemsetAsync(cnt, 0, bytes, s)) != hipSuccess) return e;
    }
    // specialised kernel for px in {3,5,7,9}; PPP_CONSENSUS_GENERIC=1 forces the generic one
    static EnvSwitch generic_sw("PPP_CONSENSUS_GENERIC");
    const bool force_generic = generic_sw.get() != nullptr;
    if (!force_generic) {
        // packed two-slice kernels (TH = 0.5, normalised product): accumulators split over two
        // waves (px in {5, 7, 9}) or in one wave; else the general v2
#ifdef PPP_BUILD_EXPERIMENTS
        const hipError_t e4 = launch_consensus_v4(pred, dtype, ov, cons, cnt, G, s);
        if (e4 != hipErrorNotSupported) { g_s1_kernel = "consensus_v4_kernel"; return e4; }
#endif
        const hipError_t e3 = launch_consensus_v3(pred, dtype, ov, cons, cnt, G, s);
        if (e3 != hipErrorNotSupported) { g_s1_kernel = "consensus_v3_kernel"; return e3; }
        const hipError_t e2 = launch_consensus_v2(pred, dtype, ov, cons, cnt, G, s);
        if (e2 != hipErrorNotSupported) {
            g_s1_kernel = G.px == 25 ? "consensus_wide_kernel" : "consensus_v2_kernel";   // (the 25-wide one lives in the v2 file)
            return e2;
        }
    }
    g_s1_kernel = "consensus_gather_kernel";
    if (dtype == PPP_F16)
        consensus_gather_kernel<__half><<<grid, block, 0, s>>>((const __half *)pred, ov, cons, cnt, G);
    else
        consensus_gather_kernel<float><<<grid, block, 0, s>>>((const float *)pred, ov, cons, cnt, G);
    return hipGetLastError();
}

// S1 for a sub-box of the cons box (G.c*): the packed kernel only
hipError_t launch_consensus_part(const void *pred, int dtype, const uint8_t *ov, float *cons, const Geo &G,
                                 hipStream_t s) {
    g_s1_kernel = "consensus_v3_kernel";
    return launch_consensus_v3(pred, dtype, ov, cons, nullptr, G, s);
}

// ---- compact -> reference layout ------------------------------------------------------
__global__ void cons_expand_kernel(const float *__restrict__ compact, float *__restrict__ ref,
                                   const Geo G) {
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= G.V) return;
    const int L = (int)blockIdx.y + 1;
    const int Ls = L + (G.py - 1) * G.wx + (G.px - 1);
    const int dx = Ls % G.wx - (G.px - 1);
    const int q = Ls / G.wx;
    const int dy = q % G.wy - (G.py - 1);
    const int dz = q / G.wy;
    const long long plane =
        ((long long)(dz + G.pz - 1) * G.nsy + (dy + G.py - 1)) * G.nsx + (dx + G.px - 1);
    ref[plane * G.V + v] = compact[(long long)(L - 1) * G.V + v];
}

hipError_t launch_cons_to_reference(const float *compact, float *ref, const Geo &G,
                                    hipStream_t s) {
    const size_t bytes = (size_t)(G.pz > 1 ? 2 * G.pz : 1) * G.nsy * G.nsx * G.V * sizeof(float);
    hipError_t e = hipMemsetAsync(ref, 0, bytes, s);
    if (e != hipSuccess) return e;
    PPP_GRID_CHECK((G.V + 255) / 256, 256);
    const dim3 grid((unsigned)((G.V + 255) / 256), (unsigned)G.n_planes);
    cons_expand_kernel<<<grid, dim3(256), 0, s>>>(compact, ref, G);
    return hipGetLastError();
}

// ---- compact (plane-major) -> symmetric voxel-major ---------------------------------------
// S[v][L] for all SIGNED offsets q (|q_i| <= p_i-1), L = Lc + (qz*wy + qy)*wx + qx:
//     q > 0 :  cons[q][v]          (v is the earlier voxel)
//     q < 0 :  cons[-q][v + q]     (v is the later voxel)
//     q = 0 :  0
// i.e. the consensus between voxel v and voxel v+q.  The patch-graph kernel reads, for a
// fixed pixel z1, a run of consecutive L -- contiguous in this layout.  64x64 LDS transpose:
// coalesced reads along the voxel axis, coalesced writes along L.
// The planes may be indexed by a box of their own (sz0 ... sX: a cache of compact planes over a
// larger box, ppp_cons_planes_to_rows): the rows' box lies inside it; a negative entry whose
// earlier voxel lies outside the PLANES' box is 0 (nothing reads it, see ppp_consensus_rows).
struct PlaneBox { int z0, y0, x0, Y, X; long long V; };
__global__ void __launch_bounds__(256)
    cons_voxel_major_kernel(const float *__restrict__ compact, float *__restrict__ S, const Geo G,
                            const int W, const PlaneBox Q) {
    __shared__ float tile[64][65];
    const int Lc = (W - 1) / 2;
    const long long v0 = (long long)blockIdx.x * 64;
    const int L0 = blockIdx.y * 64;
    // (wave index made scalar: the offset arithmetic of a tile row is wave-uniform)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long v = v0 + lane;
    int bx = 0, by = 0, bz = 0;
    if (v < G.BV) {
        bx = (int)(v % G.bX);
        const long long t = v / G.bX;
        by = (int)(t % G.bY);
        bz = (int)(t / G.bY);
    }
    for (int i = wave; i < 64; i += 4) {
        const int L = L0 + i;
        float val = 0.0f;
        if (L < W && v < G.BV && L != Lc) {
            int Ls = L - Lc;
            const bool neg = Ls < 0;
            if (neg) Ls = -Ls;
            const int t2 = Ls + (G.py - 1) * G.wx + (G.px - 1);
            const int qx = t2 % G.wx - (G.px - 1);
            const int q2 = t2 / G.wx;
            const int qy = q2 % G.wy - (G.py - 1);
            const int qz = q2 / G.wy;
            // (coordinates in the planes' box)
            const int sz = bz + G.bz0 - Q.z0, sy = by + G.by0 - Q.y0, sx = bx + G.bx0 - Q.x0;
            if (!neg) {
                val = compact[(long long)(Ls - 1) * Q.V + ((long long)sz * Q.Y + sy) * Q.X + sx];
            } else {
                const int ez = sz - qz, ey = sy - qy, ex = sx - qx;  // earlier voxel v + q
                if (ez >= 0 && ey >= 0 && ey < Q.Y && ex >= 0 && ex < Q.X)
                    val = compact[(long long)(Ls - 1) * Q.V + ((long long)ez * Q.Y + ey) * Q.X + ex];
            }
        }
        tile[i][lane] = val;
    }
    __syncthreads();
    for (int i = wave; i < 64; i += 4) {
        const long long vv = v0 + i;
        const int L = L0 + lane;
        if (vv < G.BV && L < W) S[vv * W + L] = tile[lane][i];
    }
}

hipError_t launch_cons_to_voxel_major(const float *compact, float *S, const Geo &G,
                                      hipStream_t s) {
    const int W = (2 * G.pz - 1) * G.wy * G.wx;
    PPP_GRID_CHECK((G.BV + 63) / 64, 256);
    const dim3 grid((unsigned)((G.BV + 63) / 64), (unsigned)((W + 63) / 64));
    const PlaneBox Q = {G.bz0, G.by0, G.bx0, G.bY, G.bX, G.BV};
    cons_voxel_major_kernel<<<grid, dim3(256), 0, s>>>(compact, S, G, W, Q);
    return hipGetLastError();
}

hipError_t launch_cons_planes_to_rows(const float *planes, const ppp_box &pb, float *S, const Geo &G,
                                      hipStream_t s) {
    const int W = (2 * G.pz - 1) * G.wy * G.wx;
    PPP_GRID_CHECK((G.BV + 63) / 64, 256);
    const dim3 grid((unsigned)((G.BV + 63) / 64), (unsigned)((W + 63) / 64));
    const PlaneBox Q = {pb.z0, pb.y0, pb.x0, pb.y1 - pb.y0, pb.x1 - pb.x0,
                        (long long)(pb.z1 - pb.z0) * (pb.y1 - pb.y0) * (pb.x1 - pb.x0)};
    cons_voxel_major_kernel<<<grid, dim3(256), 0, s>>>(planes, S, G, W, Q);
    return hipGetLastError();
}

}  // namespace ppp
