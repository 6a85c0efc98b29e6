// ppp_patch_graph_pa.hip -- S5 "per patch": one workgroup = one selected patch A and (a chunk
// of) ALL its pair rows (A, B); lane = one pair.
//
// Why.  aff(A, B) sums consensus[z2 - z1][earlier pixel] over pixels z1 of A and z2 of B in
// the reference's loop order (cuda/computePatchGraph.cu:38-130).  In the voxel-major layout the
// values for one z1 form ONE row S[z1][*], and that row is the same for every pair of the same
// patch A (~390 pairs at 7^3, ~800 at 9^3).  The pair-per-lane kernels in ppp_patch_graph.hip
// let every lane gather from its own rows (~13 cache lines for ~48 floats per pixel): they are
// HBM-bound (5.4 TB fetched on the 140^3 benchmark).  Here the workgroup reads the row of z1
// once, coalesced, into LDS, and all its lanes index it with their own offset
// q = (B - A) + r2 - r1: global traffic drops to (pixels of A) x (row size) per patch.
//
// Per-lane control.  The patch offset d = B - A now differs per lane, so which (r1, r2)
// combinations are in range / in the patch intersection / stored is evaluated per lane -- as
// bit masks over the p candidates of each axis (the conditions are per-axis intervals except
// for the lexicographic orientation test), so one row of candidates costs a handful of vector
// bit operations, and the inner loops walk set bits only.
//
// The per-pair float sum, the candidate order (r1 raster, then z2o, y2o, x2o ascending) and the
// LCG stream are exactly those of the reference: results are bit-identical to the other kernels.
#include "ppp_kernels.hpp"

namespace ppp {

static constexpr int PA_THREADS = 256;

// bits b in [0, n) with lo <= b <= hi
__device__ __forceinline__ uint32_t rmask(int lo, int hi, int n) {
    lo = max(lo, 0);
    hi = min(hi, n - 1);
    return lo > hi ? 0u : (((2u << hi) - 1u) & ~((1u << lo) - 1u));
}

// per-axis candidate masks for one pixel coordinate a of patch A and patch offset dd;
// candidate b of patch B has pixel offset q = dd + b - a
struct AxisMasks {
    uint32_t f;     // forward-orientation range   -(p-1) <= q <= p
    uint32_t bk;    // backward-orientation range  -p <= q <= p-1
    uint32_t pos;   // q > 0
    uint32_t zero;  // q == 0
    uint32_t st;    // |q| < p (stored planes)
    uint32_t in;    // z2 inside the window of A:  |dd + b - r| <= r
    int q0;         // q of candidate b = 0
};
__device__ __forceinline__ AxisMasks axis_masks(int dd, int a, int p) {
    AxisMasks m;
    const int q0 = dd - a, r = p / 2;
    m.q0 = q0;
    m.f = rmask(-(p - 1) - q0, p - q0, p);
    m.bk = rmask(-p - q0, p - 1 - q0, p);
    m.pos = rmask(1 - q0, p, p);
    m.zero = rmask(-q0, -q0, p);
    m.st = rmask(-(p - 1) - q0, p - 1 - q0, p);
    m.in = rmask(-dd, 2 * r - dd, p);
    return m;
}

template <typename T, int PX>
__global__ void __launch_bounds__(PA_THREADS)
    patch_graph_pa_kernel(const T *__restrict__ pred, const float *__restrict__ S,
                          const uint32_t *__restrict__ rows, const uint32_t *__restrict__ order,
                          const long long *__restrict__ group_start,
                          const long long *__restrict__ chunk_offsets, const int n_groups,
                          float *__restrict__ aff, const Geo G) {
    extern __shared__ uint32_t lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int words = (G.C + 31) / 32;
    const int W = (2 * G.pz - 1) * G.wy * G.wx, Lc = (W - 1) / 2;
    float *rowbuf = reinterpret_cast<float *>(lds_raw);          // [W]
    uint32_t *faw = lds_raw + ((W + 3) & ~3);                     // [words]
    uint32_t *fbw = faw + ((words + 3) & ~3);                     // [words][PA_THREADS]

    // ---- which (patch, chunk) is this workgroup?  binary search in the chunk prefix sums
    int g;
    {
        int lo = 0, hi = n_groups;   // find g with chunk_offsets[g] <= blockIdx.x < chunk_offsets[g+1]
        const long long b = blockIdx.x;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (chunk_offsets[mid] <= b) lo = mid; else hi = mid;
        }
        g = lo;
    }
    const long long gs = group_start[g], ge = group_start[g + 1];
    const long long pos = gs + (long long)(blockIdx.x - chunk_offsets[g]) * PA_THREADS + tid;
    const bool live = pos < ge;
    const uint32_t first_row = order[gs];
    const int az = (int)rows[(size_t)first_row * 6 + 0], ay = (int)rows[(size_t)first_row * 6 + 1],
              ax = (int)rows[(size_t)first_row * 6 + 2];
    const T *mid = pred + (long long)G.mid * G.V;

    // ---- foreground bits of patch A (shared) and of this lane's patch B
    for (int base = 0; base < G.C; base += PA_THREADS) {
        const int r = base + tid;
        bool on = false;
        if (r < G.C) {
            const int z = az + r / (G.py * PX) - G.rz, y = ay + (r / PX) % G.py - G.ry,
                      x = ax + r % PX - PX / 2;
            on = ldf(mid, vox(G, z, y, x)) > G.th_gt &&
                 ldf(pred, (long long)r * G.V + vox(G, az, ay, ax)) > G.th_gt;
        }
        const unsigned long long m = __ballot(on);
        if (lane == 0) {
            const int w = (base + (tid & ~63)) >> 5;
            if (w < words) faw[w] = (uint32_t)m;
            if (w + 1 < words) faw[w + 1] = (uint32_t)(m >> 32);
        }
    }
    uint32_t row_id = 0;
    int dz = 0, dy = 0, dx = 0;
    uint32_t rnd = 0;
    if (live) {
        row_id = order[pos];
        const uint32_t *rw = rows + (size_t)row_id * 6;
        const int bz = (int)rw[3], by = (int)rw[4], bx = (int)rw[5];
        dz = bz - az; dy = by - ay; dx = bx - ax;
        rnd = (uint32_t)(az + G.oz) * (uint32_t)(bz + G.oz) * (uint32_t)(ay + G.oy) *
              (uint32_t)(by + G.oy) * (uint32_t)(ax + G.ox) * (uint32_t)(bx + G.ox);
        const long long lb = vox(G, bz, by, bx);
        int r = 0;
        for (int w = 0; w < words; ++w) {
            uint32_t bits = 0;
            for (int b = 0; b < 32 && r < G.C; ++b, ++r) {
                const int z = bz + r / (G.py * PX) - G.rz, y = by + (r / PX) % G.py - G.ry,
                          x = bx + r % PX - PX / 2;
                const bool on = ldf(mid, vox(G, z, y, x)) > G.th_gt &&
                                ldf(pred, (long long)r * G.V + lb) > G.th_gt;
                bits |= (on ? 1u : 0u) << b;
            }
            fbw[w * PA_THREADS + tid] = bits;
        }
    } else {
        for (int w = 0; w < words; ++w) fbw[w * PA_THREADS + tid] = 0u;
    }
    __syncthreads();

    const long long sY = G.bX, sZ = (long long)G.bX * G.bY;
    const long long baseA = ((long long)(az - G.bz0) * G.bY + (ay - G.by0)) * G.bX + (ax - G.bx0);
    float acc = 0.0f;
    unsigned fg_cnt = 0;

    int r1 = 0;
    for (int z1o = 0; z1o < G.pz; ++z1o) {
        const AxisMasks mz = axis_masks(dz, z1o, G.pz);
        for (int y1o = 0; y1o < G.py; ++y1o) {
            const AxisMasks my = axis_masks(dy, y1o, G.py);
            for (int x1o = 0; x1o < PX; ++x1o, ++r1) {
                if (!((faw[r1 >> 5] >> (r1 & 31)) & 1u)) continue;     // uniform: z1 not in F_A
                // ---- stage the consensus row of z1 (shared by every pair of A)
                __syncthreads();                                     // previous row fully consumed
                {
                    const float *src = S + (baseA + (long long)(z1o - G.rz) * sZ +
                                            (long long)(y1o - G.ry) * sY + (x1o - PX / 2)) * W;
                    for (int i = tid; i < W; i += PA_THREADS) rowbuf[i] = src[i];
                }
                __syncthreads();
                if (!live) continue;
                const AxisMasks mx = axis_masks(dx, x1o, PX);
                const bool in_b = abs(x1o - PX / 2 - dx) <= PX / 2 && abs(y1o - G.ry - dy) <= G.ry &&
                                  abs(z1o - G.rz - dz) <= G.rz;
                const uint32_t mx_nonneg = mx.pos | mx.zero;
                uint32_t zset = mz.f | mz.bk;
                while (zset) {
                    const int z2o = __ffs(zset) - 1;
                    zset &= zset - 1;
                    const uint32_t zb = 1u << z2o;
                    const int qz = mz.q0 + z2o;
                    uint32_t yset = my.f | my.bk;
                    while (yset) {
                        const int y2o = __ffs(yset) - 1;
                        yset &= yset - 1;
                        const uint32_t yb = 1u << y2o;
                        const int qy = my.q0 + y2o;
                        // orientation: z1 before z2 in raster order <=> q >= 0 lexicographically
                        const bool fwd_all = (mz.pos & zb) || ((mz.zero & zb) && (my.pos & yb));
                        const bool fwd_x = (mz.zero & zb) && (my.zero & yb);
                        const uint32_t m_fwd = fwd_all ? ~0u : (fwd_x ? mx_nonneg : 0u);
                        const bool zy_f = (mz.f & zb) && (my.f & yb), zy_b = (mz.bk & zb) && (my.bk & yb);
                        const uint32_t m_range = (zy_f ? (mx.f & m_fwd) : 0u) | (zy_b ? (mx.bk & ~m_fwd) : 0u);
                        const uint32_t m_inter = (in_b && (mz.in & zb) && (my.in & yb)) ? mx.in : 0u;
                        uint32_t m_stored = ((mz.st & zb) && (my.st & yb)) ? (mx.st & m_range) : 0u;
                        if (fwd_x) m_stored &= ~mx.zero;                 // q == 0 is never stored
                        uint32_t cand = m_range | m_inter;
                        if (!cand) continue;
                        // foreground bits of patch B on this candidate row
                        const int b0 = (z2o * G.py + y2o) * PX, w0 = b0 >> 5, sh = b0 & 31;
                        uint64_t f64 = fbw[w0 * PA_THREADS + tid];
                        if (sh + PX > 32 && w0 + 1 < words) f64 |= (uint64_t)fbw[(w0 + 1) * PA_THREADS + tid] << 32;
                        const uint32_t fb = (uint32_t)(f64 >> sh);
                        const float *rowq = rowbuf + Lc + (qz * G.wy + qy) * G.wx + mx.q0;
                        while (cand) {
                            const int x2o = __ffs(cand) - 1;
                            cand &= cand - 1;
                            const uint32_t xb = 1u << x2o;
                            bool valid = (fb & xb) != 0u;
                            if (m_inter & xb) {
                                const uint32_t nxt = rnd * 1103515245U;
                                const float rnd_t = (float)nxt / 4294967296.0f;
                                rnd = valid ? nxt : rnd;
                                valid = valid && !(rnd_t > 0.19999998807907104f);   // > 0.2 in double
                            }
                            if (m_range & xb) {
                                if ((m_stored & xb) && valid) acc += rowq[x2o];
                                fg_cnt += valid ? 1u : 0u;
                            }
                        }
                    }
                }
            }
        }
    }
    if (live) aff[row_id] = G.norm_aff ? acc / (float)(fg_cnt > 1u ? fg_cnt : 1u) : acc;
}

hipError_t launch_patch_graph_pa(const void *pred, int dtype, const float *S, const uint32_t *rows,
                                 const uint32_t *order, const long long *group_start,
                                 const long long *chunk_offsets, int n_groups, long long n_blocks,
                                 float *aff, const Geo &G, hipStream_t s) {
    if (n_groups <= 0 || n_blocks <= 0) return hipSuccess;
    const int words = (G.C + 31) / 32;
    const int W = (2 * G.pz - 1) * G.wy * G.wx;
    const size_t lds = (size_t)(((W + 3) & ~3) + ((words + 3) & ~3) + words * PA_THREADS) * 4;
    if (lds > 64 * 1024 || n_blocks >= (1ll << 31)) return hipErrorNotSupported;
#define PPP_PA_CASE(P)                                                                             \
    case P:                                                                                        \
        if (dtype == PPP_F16)                                                                      \
            patch_graph_pa_kernel<__half, P><<<dim3((unsigned)n_blocks), dim3(PA_THREADS), lds, s>>>( \
                (const __half *)pred, S, rows, order, group_start, chunk_offsets, n_groups, aff, G); \
        else                                                                                       \
            patch_graph_pa_kernel<float, P><<<dim3((unsigned)n_blocks), dim3(PA_THREADS), lds, s>>>( \
                (const float *)pred, S, rows, order, group_start, chunk_offsets, n_groups, aff, G); \
        return hipGetLastError();
    switch (G.px) {
        PPP_PA_CASE(3)
        PPP_PA_CASE(5)
        PPP_PA_CASE(7)
        PPP_PA_CASE(9)
    default:
        return hipErrorNotSupported;
    }
#undef PPP_PA_CASE
}

}  // namespace ppp
