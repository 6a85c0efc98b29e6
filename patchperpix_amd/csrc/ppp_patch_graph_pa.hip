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
// for the lexicographic orientation test).  Control flow stays wave-uniform: the (z2o, y2o)
// candidate rows are walked over the union of the lanes' ranges (the rows of a patch are sorted
// by offset, so the lanes of a wave have similar ranges), the PX candidates of a row are
// predicated, and the LCG branch is only entered by waves in which some lane is inside the
// patch intersection on that row.  The row of the next pixel is fetched into registers while
// the current one is consumed (two LDS row buffers, one barrier per pixel).
//
// The per-pair float sum, the candidate order (r1 raster, then z2o, y2o, x2o ascending) and the
// LCG stream are exactly those of the reference: results are bit-identical to the other kernels.
#include "ppp_kernels.hpp"

namespace ppp {

// threads (= pair rows) per workgroup: the [words][threads] foreground bits of the B patches and
// two row buffers must fit 64 KB of LDS
template <int PX> struct PaCfg { static constexpr int THREADS = PX <= 7 ? 512 : 256; };
static constexpr int PA_PAD = 16;      // floats of slack either side of the staged row

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
__global__ void __launch_bounds__(PaCfg<PX>::THREADS)
    patch_graph_pa_kernel(const T *__restrict__ pred, const float *__restrict__ S,
                          const uint32_t *__restrict__ rows, const uint32_t *__restrict__ order,
                          const long long *__restrict__ group_start,
                          const long long *__restrict__ chunk_offsets, const int n_groups,
                          float *__restrict__ aff, const Geo G) {
    extern __shared__ uint32_t lds_raw[];
    constexpr int PA_THREADS = PaCfg<PX>::THREADS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int words = (G.C + 31) / 32;
    const int W = (2 * G.pz - 1) * G.wy * G.wx, Lc = (W - 1) / 2;
    const int WB = (W + 2 * PA_PAD + 3) & ~3;                     // floats per row buffer
    // staged floats per thread (the launcher checks W <= NST * PA_THREADS)
    constexpr int NST = ((2 * PX - 1) * (2 * PX - 1) * (2 * PX - 1) + PA_THREADS - 1) / PA_THREADS;
    float *rowbuf = reinterpret_cast<float *>(lds_raw);           // [2][WB]
    uint32_t *faw = lds_raw + 2 * WB;                             // [words]
    uint16_t *ulist = reinterpret_cast<uint16_t *>(faw + ((words + 3) & ~3));   // [C] pixels of F_A
    uint32_t *fbw = faw + ((words + 3) & ~3) + (((G.C + 1) / 2 + 3) & ~3);          // [words][PA_THREADS]
    __shared__ int s_nu;

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
    int dz = 1 << 20, dy = 1 << 20, dx = 1 << 20;      // idle lanes: every mask comes out empty
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
    // list of the pixels of A that are in F_A, raster order
    if (tid == 0) {
        int n = 0;
        for (int r = 0; r < G.C; ++r)
            if ((faw[r >> 5] >> (r & 31)) & 1u) ulist[n++] = (uint16_t)r;
        s_nu = n;
    }
    __syncthreads();
    const int n_u = s_nu;

    // wave-uniform bounds of the lanes' patch offsets (idle lanes excluded)
    int dz_lo = live ? dz : (1 << 20), dz_hi = live ? dz : -(1 << 20);
    int dy_lo = live ? dy : (1 << 20), dy_hi = live ? dy : -(1 << 20);
    for (int o = 32; o > 0; o >>= 1) {
        dz_lo = min(dz_lo, __shfl_xor(dz_lo, o)); dz_hi = max(dz_hi, __shfl_xor(dz_hi, o));
        dy_lo = min(dy_lo, __shfl_xor(dy_lo, o)); dy_hi = max(dy_hi, __shfl_xor(dy_hi, o));
    }
    dz_lo = __builtin_amdgcn_readfirstlane(dz_lo); dz_hi = __builtin_amdgcn_readfirstlane(dz_hi);
    dy_lo = __builtin_amdgcn_readfirstlane(dy_lo); dy_hi = __builtin_amdgcn_readfirstlane(dy_hi);
    const bool wave_live = dz_lo <= dz_hi;

    const long long sY = G.bX, sZ = (long long)G.bX * G.bY;
    const long long baseA = ((long long)(az - G.bz0) * G.bY + (ay - G.by0)) * G.bX + (ax - G.bx0);
    auto row_of = [&](int r1) -> const float * {
        const int z1o = r1 / (G.py * PX), y1o = (r1 / PX) % G.py, x1o = r1 % PX;
        return S + (baseA + (long long)(z1o - G.rz) * sZ + (long long)(y1o - G.ry) * sY + (x1o - PX / 2)) * W;
    };
    float acc = 0.0f;
    unsigned fg_cnt = 0;

    // ---- stage the first row
    float st[NST];
    if (n_u > 0) {
        const float *src = row_of(ulist[0]);
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int e = tid + i * PA_THREADS;
            if (e < W) rowbuf[PA_PAD + e] = src[e];
        }
    }
    __syncthreads();

    int prev_z1o = -1, prev_y1o = -1;
    AxisMasks mz = axis_masks(dz, 0, G.pz), my = axis_masks(dy, 0, G.py);
    for (int k = 0; k < n_u; ++k) {
        const int r1 = ulist[k];
        const int z1o = r1 / (G.py * PX), y1o = (r1 / PX) % G.py, x1o = r1 % PX;
        const float *cur = rowbuf + (k & 1) * WB + PA_PAD;
        // ---- fetch the next row into registers while this one is consumed
        const bool more = k + 1 < n_u;
        if (more) {
            const float *src = row_of(ulist[k + 1]);
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int e = tid + i * PA_THREADS;
                st[i] = e < W ? src[e] : 0.0f;
            }
        }
        if (wave_live) {
            if (z1o != prev_z1o) { mz = axis_masks(dz, z1o, G.pz); prev_z1o = z1o; prev_y1o = -1; }
            if (y1o != prev_y1o) { my = axis_masks(dy, y1o, G.py); prev_y1o = y1o; }
            const AxisMasks mx = axis_masks(dx, x1o, PX);
            const bool in_b = abs(x1o - PX / 2 - dx) <= PX / 2 && abs(y1o - G.ry - dy) <= G.ry &&
                              abs(z1o - G.rz - dz) <= G.rz;
            const uint32_t mx_nonneg = mx.pos | mx.zero;
            // union of the lanes' candidate ranges: -p <= d + b - a <= p
            const int z_lo = max(0, z1o - dz_hi - G.pz), z_hi = min(G.pz - 1, z1o - dz_lo + G.pz);
            const int y_lo = max(0, y1o - dy_hi - G.py), y_hi = min(G.py - 1, y1o - dy_lo + G.py);
            for (int z2o = z_lo; z2o <= z_hi; ++z2o) {
                const uint32_t zb = 1u << z2o;
                const int qz = mz.q0 + z2o;
                const bool z_f = mz.f & zb, z_bk = mz.bk & zb, z_pos = mz.pos & zb, z_zero = mz.zero & zb,
                           z_st = mz.st & zb, z_in = mz.in & zb;
                if (__ballot(z_f || z_bk) == 0ull) continue;
                for (int y2o = y_lo; y2o <= y_hi; ++y2o) {
                    const uint32_t yb = 1u << y2o;
                    const int qy = my.q0 + y2o;
                    // orientation: z1 before z2 in raster order <=> q >= 0 lexicographically
                    const bool fwd_all = z_pos || (z_zero && (my.pos & yb));
                    const bool fwd_x = z_zero && (my.zero & yb);
                    const uint32_t m_fwd = fwd_all ? ~0u : (fwd_x ? mx_nonneg : 0u);
                    const bool zy_f = z_f && (my.f & yb), zy_b = z_bk && (my.bk & yb);
                    const uint32_t m_range = (zy_f ? (mx.f & m_fwd) : 0u) | (zy_b ? (mx.bk & ~m_fwd) : 0u);
                    const uint32_t m_inter = (in_b && z_in && (my.in & yb)) ? mx.in : 0u;
                    const bool row_st = z_st && (my.st & yb);
                    uint32_t m_stored = row_st ? (mx.st & m_range) : 0u;
                    if (fwd_x) m_stored &= ~mx.zero;                 // q == 0 is never stored
                    if (__ballot((m_range | m_inter) != 0u) == 0ull) continue;
                    // foreground bits of patch B on this candidate row
                    const int b0 = (z2o * G.py + y2o) * PX, w0 = b0 >> 5, sh = b0 & 31;
                    uint64_t f64 = fbw[w0 * PA_THREADS + tid];
                    if (sh + PX > 32 && w0 + 1 < words) f64 |= (uint64_t)fbw[(w0 + 1) * PA_THREADS + tid] << 32;
                    const uint32_t fb = (uint32_t)(f64 >> sh) & ((1u << PX) - 1u);
                    // the PX consensus values of the row (lanes without a stored row read slot 0)
                    // (mx.st != 0 bounds q0, so the PX reads stay inside the padded buffer)
                    const float *rowq = cur + ((row_st && mx.st != 0u) ? Lc + (qz * G.wy + qy) * G.wx + mx.q0 : 0);
                    float v[PX];
#pragma unroll
                    for (int t = 0; t < PX; ++t) v[t] = rowq[t];
                    uint32_t valid = fb;
                    if (__ballot(m_inter != 0u) != 0ull) {
                        // thinning inside the patch intersection: the LCG advances on every
                        // foreground candidate, in candidate order
#pragma unroll
                        for (int t = 0; t < PX; ++t) {
                            const uint32_t xb = 1u << t;
                            const uint32_t nxt = rnd * 1103515245U;
                            const float rnd_t = (float)nxt / 4294967296.0f;
                            const bool hit = (m_inter & fb & xb) != 0u;
                            rnd = hit ? nxt : rnd;
                            // rnd_t > 0.2 (double)  <=>  rnd_t > largest float <= 0.2
                            if (hit && rnd_t > 0.19999998807907104f) valid &= ~xb;
                        }
                    }
                    const uint32_t add = m_stored & valid;
#pragma unroll
                    for (int t = 0; t < PX; ++t) acc += ((add >> t) & 1u) ? v[t] : 0.0f;
                    fg_cnt += __popc(m_range & valid);
                }
            }
        }
        // ---- publish the next row
        if (more) {
            float *dst = rowbuf + ((k + 1) & 1) * WB + PA_PAD;
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int e = tid + i * PA_THREADS;
                if (e < W) dst[e] = st[i];
            }
        }
        __syncthreads();
    }
    if (live) aff[row_id] = G.norm_aff ? acc / (float)(fg_cnt > 1u ? fg_cnt : 1u) : acc;
}

int patch_graph_pa_chunk(const Geo &G) {
    switch (G.px) {
    case 3: return PaCfg<3>::THREADS;
    case 5: return PaCfg<5>::THREADS;
    case 7: return PaCfg<7>::THREADS;
    case 9: return PaCfg<9>::THREADS;
    }
    return 0;
}

hipError_t launch_patch_graph_pa(const void *pred, int dtype, const float *S, const uint32_t *rows,
                                 const uint32_t *order, const long long *group_start,
                                 const long long *chunk_offsets, int n_groups, long long n_blocks,
                                 float *aff, const Geo &G, hipStream_t s) {
    if (n_groups <= 0 || n_blocks <= 0) return hipSuccess;
    const int threads = patch_graph_pa_chunk(G);
    if (threads == 0) return hipErrorNotSupported;
    const int words = (G.C + 31) / 32;
    const int W = (2 * G.pz - 1) * G.wy * G.wx;
    const int WB = (W + 2 * PA_PAD + 3) & ~3;
    // the per-thread staging registers are sized for a (2px-1)^3 row
    const int cube = (2 * G.px - 1) * (2 * G.px - 1) * (2 * G.px - 1);
    if (W > (cube + threads - 1) / threads * threads) return hipErrorNotSupported;
    const size_t lds = (size_t)(2 * WB + ((words + 3) & ~3) + (((G.C + 1) / 2 + 3) & ~3) + words * threads) * 4;
    if (lds > 64 * 1024 || n_blocks >= (1ll << 31)) return hipErrorNotSupported;
#define PPP_PA_CASE(P)                                                                             \
    case P:                                                                                        \
        if (dtype == PPP_F16)                                                                      \
            patch_graph_pa_kernel<__half, P><<<dim3((unsigned)n_blocks), dim3(threads), lds, s>>>(    \
                (const __half *)pred, S, rows, order, group_start, chunk_offsets, n_groups, aff, G); \
        else                                                                                       \
            patch_graph_pa_kernel<float, P><<<dim3((unsigned)n_blocks), dim3(threads), lds, s>>>(     \
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
