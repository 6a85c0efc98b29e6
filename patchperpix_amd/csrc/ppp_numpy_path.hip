// ppp_numpy_path.hip -- the reference's NumPy-semantics stages (`cuda=False`) on the device.
//
// Reference (a DIFFERENT function from the CUDA kernels' -- integer votes):
//   utilVoteInstances.py:59-92 + get_patch_sets.py:32-79   per centre c (interior, foreground):
//        pf_c = { v in win(c) : pred[r_v][c] >  th      and fg[v] }
//        pb_c = { v in win(c) : pred[r_v][c] <  1 - th  and fg[v] }      (float32 compares)
//   consensus_array.py:18-68      int16 votes: for every centre, +1 on the key of every unordered
//        pair {u, w} of pf_c, -1 on the key of every pair (u in pf_c, w in pb_c); a key is
//        (lexicographically positive offset, earlier voxel) -- utilVoteInstances.py:19-56 -- and a
//        NumPy `a[idx] -= 1` applies once per DISTINCT key of a centre (only matters for th < 0.5,
//        where a pixel can be in both sets; then also the zero-offset key (0, v) exists)
//   ranked_patches.py:76-105      score(c) = #{ff keys with vote > 0} - #{ff keys <= 0}
//                                           + #{fb entries with vote < 0} - #{fb entries >= 0}
//                                 (fb entries: one per ORDERED pair, duplicates counted)
//   aff_patch_graph.py:209-282    weight(A, B) = sum of the votes over all (p in pf_A, q in pf_B)
//        with |p - q| < patchshape on every axis and p != q, pf taken against mask_to_cover; the
//        edge exists iff there is at least one such pair.
// Everything is integer arithmetic: results are exact whatever the order of summation, so the
// reductions use wavefront shuffles.
//
// Layout: votes int16 [plane][Z][Y][X], plane q = linear signed offset (dz * (2py-1) + dy) * (2px-1)
// + dx of the lexicographically positive offsets (1 .. n_planes, the COMPACT order of the float
// consensus) and plane 0 = the zero offset.
#include "ppp_kernels.hpp"

namespace ppp {

struct NpGeo {
    int Z, Y, X, pz, py, px, rz, ry, rx, C, wy, wx, n_planes;
    long long V;
    float th, bg;   // float32(th), float32(1 - th): NumPy compares the float32 patch with these
};

static NpGeo np_geo(const Geo &G, double th) {
    NpGeo g;
    g.Z = G.Z; g.Y = G.Y; g.X = G.X; g.pz = G.pz; g.py = G.py; g.px = G.px;
    g.rz = G.rz; g.ry = G.ry; g.rx = G.rx; g.C = G.C; g.wy = G.wy; g.wx = G.wx;
    g.n_planes = G.n_planes; g.V = G.V;
    g.th = (float)th;
    g.bg = (float)(1.0 - th);
    return g;
}

__device__ __forceinline__ long long nvox(const NpGeo &g, int z, int y, int x) {
    return ((long long)z * g.Y + y) * g.X + x;
}
__device__ __forceinline__ bool np_interior(const NpGeo &g, int z, int y, int x) {
    return z >= g.rz && z < g.Z - g.rz && y >= g.ry && y < g.Y - g.ry && x >= g.rx && x < g.X - g.rx;
}
// key of the voxel pair (p, q), p != q or p == q: plane and base voxel
__device__ __forceinline__ long long np_key(const NpGeo &g, int pz, int py, int px, int qz, int qy, int qx) {
    int dz = qz - pz, dy = qy - py, dx = qx - px;
    int bz = pz, by = py, bx = px;
    if (dz < 0 || (dz == 0 && (dy < 0 || (dy == 0 && dx < 0)))) {   // not positive: key of (q, p)
        dz = -dz; dy = -dy; dx = -dx;
        bz = qz; by = qy; bx = qx;
    }
    const int plane = (dz * g.wy + dy) * g.wx + dx;                  // 0 for p == q
    return (long long)plane * g.V + nvox(g, bz, by, bx);
}

// ---- votes: thread per (plane, base voxel), gathered over the centres that see both voxels ----
template <typename T>
__global__ void __launch_bounds__(256)
    np_consensus_kernel(const T *__restrict__ pred, const uint8_t *__restrict__ fg, int16_t *__restrict__ cons,
                        const NpGeo g) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)(g.n_planes + 1) * g.V;
    if (t >= total) return;
    const int plane = (int)(t / g.V);
    const long long u = t % g.V;
    const int ux = (int)(u % g.X), uy = (int)((u / g.X) % g.Y), uz = (int)(u / ((long long)g.X * g.Y));
    // signed offset of the plane
    int dx = plane % g.wx, dy = (plane / g.wx) % g.wy, dz = plane / (g.wx * g.wy);
    if (dx > g.px - 1) { dx -= g.wx; dy += 1; }
    if (dy > g.py - 1) { dy -= g.wy; dz += 1; }
    const int wz = uz + dz, wy_ = uy + dy, wx_ = ux + dx;
    int votes = 0;
    if (wz < g.Z && wy_ >= 0 && wy_ < g.Y && wx_ >= 0 && wx_ < g.X && fg[u] && fg[nvox(g, wz, wy_, wx_)]) {
        // centres whose window holds u and w
        const int cz0 = max(max(uz, wz) - g.rz, g.rz), cz1 = min(min(uz, wz) + g.rz, g.Z - 1 - g.rz);
        const int cy0 = max(max(uy, wy_) - g.ry, g.ry), cy1 = min(min(uy, wy_) + g.ry, g.Y - 1 - g.ry);
        const int cx0 = max(max(ux, wx_) - g.rx, g.rx), cx1 = min(min(ux, wx_) + g.rx, g.X - 1 - g.rx);
        for (int cz = cz0; cz <= cz1; ++cz)
            for (int cy = cy0; cy <= cy1; ++cy)
                for (int cx = cx0; cx <= cx1; ++cx) {
                    const long long c = nvox(g, cz, cy, cx);
                    if (!fg[c]) continue;
                    const int ku = ((uz - cz + g.rz) * g.py + (uy - cy + g.ry)) * g.px + (ux - cx + g.rx);
                    const float vu = ldf(pred, (long long)ku * g.V + c);
                    const bool pu = vu > g.th, nu = vu < g.bg;
                    if (plane == 0) {
                        votes -= (pu && nu) ? 1 : 0;
                        continue;
                    }
                    const int kw = ((wz - cz + g.rz) * g.py + (wy_ - cy + g.ry)) * g.px + (wx_ - cx + g.rx);
                    const float vw = ldf(pred, (long long)kw * g.V + c);
                    const bool pw = vw > g.th, nw = vw < g.bg;
                    votes += (pu && pw) ? 1 : 0;
                    votes -= ((pu && nw) || (pw && nu)) ? 1 : 0;
                }
    }
    cons[t] = (int16_t)votes;
}

// ---- ranking: one wave per centre ---------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(64)
    np_rank_kernel(const T *__restrict__ pred, const uint8_t *__restrict__ fg, const int16_t *__restrict__ cons,
                   int32_t *__restrict__ score, const NpGeo g) {
    extern __shared__ uint8_t cls[];          // per channel: bit 0 = in pf, bit 1 = in pb
    const long long c = blockIdx.x;
    const int lane = threadIdx.x;
    const int cx = (int)(c % g.X), cy = (int)((c / g.X) % g.Y), cz = (int)(c / ((long long)g.X * g.Y));
    if (!np_interior(g, cz, cy, cx) || !fg[c]) {
        if (lane == 0) score[c] = 0;
        return;
    }
    for (int k = lane; k < g.C; k += 64) {
        const int z = cz + k / (g.py * g.px) - g.rz, y = cy + (k / g.px) % g.py - g.ry, x = cx + k % g.px - g.rx;
        const float v = ldf(pred, (long long)k * g.V + c);
        const bool in = fg[nvox(g, z, y, x)] != 0;
        cls[k] = (uint8_t)((in && v > g.th ? 1 : 0) | (in && v < g.bg ? 2 : 0));
    }
    __syncthreads();
    int s = 0;
    for (int k1 = 0; k1 < g.C; ++k1) {
        if (!(cls[k1] & 1)) continue;          // (wave-uniform)
        const int z1 = cz + k1 / (g.py * g.px) - g.rz, y1 = cy + (k1 / g.px) % g.py - g.ry, x1 = cx + k1 % g.px - g.rx;
        for (int k2 = lane; k2 < g.C; k2 += 64) {
            const uint8_t c2 = cls[k2];
            const bool ff = (c2 & 1) && k2 > k1, fb = (c2 & 2) != 0;
            if (!ff && !fb) continue;
            const int z2 = cz + k2 / (g.py * g.px) - g.rz, y2 = cy + (k2 / g.px) % g.py - g.ry, x2 = cx + k2 % g.px - g.rx;
            const int v = cons[np_key(g, z1, y1, x1, z2, y2, x2)];
            if (ff) s += v > 0 ? 1 : -1;
            if (fb) s += v < 0 ? 1 : -1;
        }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) score[c] = s;
}

// ---- patch graph: one wave per candidate row (A, B) -----------------------------------------------
template <typename T>
__global__ void __launch_bounds__(64)
    np_graph_kernel(const T *__restrict__ pred, const uint8_t *__restrict__ mask, const int16_t *__restrict__ cons,
                    const int32_t *__restrict__ rows, long long *__restrict__ weight, int32_t *__restrict__ count,
                    const NpGeo g) {
    extern __shared__ uint8_t cls[];          // [0, C): pf of A, [C, 2C): pf of B
    const long long r = blockIdx.x;
    const int lane = threadIdx.x;
    const int az = rows[r * 6], ay = rows[r * 6 + 1], ax = rows[r * 6 + 2];
    const int bz = rows[r * 6 + 3], by = rows[r * 6 + 4], bx = rows[r * 6 + 5];
    for (int k = lane; k < 2 * g.C; k += 64) {
        const int kk = k < g.C ? k : k - g.C;
        const int cz = k < g.C ? az : bz, cy = k < g.C ? ay : by, cx = k < g.C ? ax : bx;
        uint8_t in = 0;
        // get_foreground_set: empty unless the window lies inside the volume
        if (np_interior(g, cz, cy, cx)) {
            const int z = cz + kk / (g.py * g.px) - g.rz, y = cy + (kk / g.px) % g.py - g.ry, x = cx + kk % g.px - g.rx;
            in = (mask[nvox(g, z, y, x)] && ldf(pred, (long long)kk * g.V + nvox(g, cz, cy, cx)) > g.th) ? 1 : 0;
        }
        cls[k] = in;
    }
    __syncthreads();
    long long w = 0;
    int n = 0;
    for (int k1 = 0; k1 < g.C; ++k1) {
        if (!cls[k1]) continue;
        const int z1 = az + k1 / (g.py * g.px) - g.rz, y1 = ay + (k1 / g.px) % g.py - g.ry, x1 = ax + k1 % g.px - g.rx;
        for (int k2 = lane; k2 < g.C; k2 += 64) {
            if (!cls[g.C + k2]) continue;
            const int z2 = bz + k2 / (g.py * g.px) - g.rz, y2 = by + (k2 / g.px) % g.py - g.ry, x2 = bx + k2 % g.px - g.rx;
            const int dz = z1 - z2, dy = y1 - y2, dx = x1 - x2;
            if (abs(dz) >= g.pz || abs(dy) >= g.py || abs(dx) >= g.px || (dz == 0 && dy == 0 && dx == 0)) continue;
            w += cons[np_key(g, z1, y1, x1, z2, y2, x2)];
            ++n;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        w += __shfl_xor(w, o);
        n += __shfl_xor(n, o);
    }
    if (lane == 0) { weight[r] = w; count[r] = n; }
}

hipError_t launch_np_consensus(const void *pred, int dtype, const uint8_t *fg, int16_t *cons, const Geo &G,
                               double th, hipStream_t s) {
    const NpGeo g = np_geo(G, th);
    const long long total = (long long)(g.n_planes + 1) * g.V;
    PPP_GRID_CHECK((total + 255) / 256, 256);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == PPP_F16) np_consensus_kernel<__half><<<grid, dim3(256), 0, s>>>((const __half *)pred, fg, cons, g);
    else np_consensus_kernel<float><<<grid, dim3(256), 0, s>>>((const float *)pred, fg, cons, g);
    return hipGetLastError();
}

hipError_t launch_np_rank(const void *pred, int dtype, const uint8_t *fg, const int16_t *cons, int32_t *score,
                          const Geo &G, double th, hipStream_t s) {
    const NpGeo g = np_geo(G, th);
    PPP_GRID_CHECK(g.V, 64);
    const dim3 grid((unsigned)g.V);
    const size_t lds = (size_t)g.C;
    if (dtype == PPP_F16) np_rank_kernel<__half><<<grid, dim3(64), lds, s>>>((const __half *)pred, fg, cons, score, g);
    else np_rank_kernel<float><<<grid, dim3(64), lds, s>>>((const float *)pred, fg, cons, score, g);
    return hipGetLastError();
}

hipError_t launch_np_graph(const void *pred, int dtype, const uint8_t *mask, const int16_t *cons,
                           const int32_t *rows, uint64_t n, long long *weight, int32_t *count, const Geo &G,
                           double th, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const NpGeo g = np_geo(G, th);
    PPP_GRID_CHECK(n, 64);
    const dim3 grid((unsigned)n);
    const size_t lds = 2 * (size_t)g.C;
    if (dtype == PPP_F16)
        np_graph_kernel<__half><<<grid, dim3(64), lds, s>>>((const __half *)pred, mask, cons, rows, weight, count, g);
    else
        np_graph_kernel<float><<<grid, dim3(64), lds, s>>>((const float *)pred, mask, cons, rows, weight, count, g);
    return hipGetLastError();
}

}  // namespace ppp
