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
// predicated.
//
// Thinning (computePatchGraph.cu:75-86: inside the intersection of the two windows a candidate is
// kept with probability 0.2, drawn from the pair's own LCG).  For the 3-d patch widths the
// decisions are made BEFOREHAND by patch_graph_lcg_kernel (second half of this file: a lane per
// intersecting pair) and arrive as one 64-bit mask of dropped candidates per chunk of candidate
// rows; pair rows without masks (2-d 25-wide patches, rows beyond the caller's mask budget) run
// the generator here, in a branch only entered by waves in which some such lane is inside the
// intersection on that row.  Same stream, same decisions.
//
// The row of the next pixel: where masks are read it is loaded AFTER the pixel step (a global load
// inside the step waits for everything in flight -- every s_waitcnt here is vmcnt(0) -- so a
// register prefetch of the row would only move its latency to the first mask load); otherwise it
// is fetched into registers while the current one is consumed.  One LDS row buffer and two
// barriers per pixel from 7^3 on, two buffers and one barrier below.
//
// The per-pair float sum, the candidate order (r1 raster, then z2o, y2o, x2o ascending) and the
// LCG stream are exactly those of the reference: results are bit-identical to the other kernels.
#include "ppp_kernels.hpp"

namespace ppp {

#ifdef PA_STATS
__device__ unsigned long long pa_stats[8];   // steps, planes, rows, useful slots, lcg rows, live lanes*steps
#define PA_STAT(i, v) st_##i += (v)
#else
#define PA_STAT(i, v)
#endif

// threads (= pair rows) per workgroup: the [words][threads] foreground bits of the B patches and
// two row buffers must fit 64 KB of LDS
template <int PX> struct PaCfg {
#ifndef PPP_PA_MINWAVES7
#define PPP_PA_MINWAVES7 8
#endif
#ifndef PPP_PA_THREADS9
#define PPP_PA_THREADS9 256
#endif
    static constexpr int THREADS = PX >= 9 ? PPP_PA_THREADS9 : 256;
#ifndef PPP_PA_SINGLE_BUF_FROM
#define PPP_PA_SINGLE_BUF_FROM 7
#endif
    // Occupancy is what this kernel lacks (measured, 140^3 / 7^3 and 128^3 / 9^3):
    //   7^3: two row buffers + 5 waves/SIMD (91 VGPRs) 621 ms; ONE buffer (two barriers per
    //        pixel, 21 KB LDS -> 7 workgroups per CU) + 7 waves/SIMD (72 VGPRs, 20 spilled)
    //        588 ms; one buffer at 5 waves: no change -- the second barrier is free
    //   9^3: two buffers (64 KB, 2 workgroups per CU) 7.6 s; one buffer (45 KB, 3 per CU) 6.0 s;
    //        512-thread workgroups at 4 waves/SIMD: 6.0 s
    //   (dropping the register prefetch of the next row instead: 632 ms / 6.5 s -- worse)
    //   7^3 with the pixel list replaced by a scalar bit iterator and 8 floats of row padding:
    //        20.2 KB of LDS -> 8 workgroups per CU, 8 waves/SIMD (64 VGPRs, 22 spilled): 542 ms
    // so PX >= 7 takes one row buffer.
    static constexpr int ROW_BUFS = PX >= PPP_PA_SINGLE_BUF_FROM ? 1 : 2;
    // waves per SIMD the register budget must allow
    static constexpr int MIN_WAVES = PX <= 7 ? PPP_PA_MINWAVES7 : (PX >= 25 ? 1 : (THREADS == 512 ? 4 : (ROW_BUFS == 1 ? 3 : 2)));
    // Small workgroups for lists with few rows per patch (after set-cover thinning a patch has
    // ~60 dispatched rows at 7^3): with 256 threads three of four waves only stage and wait
    // (29 % of the issued instructions on the thinned 140^3 volume), and the 11 KB of B-patch
    // bits they reserve cap the CU at 8 workgroups.  One or two waves: 11.6 KB -> 13 per CU, and
    // a register budget without spills.
    static constexpr int THREADS_SMALL = PX >= 25 ? 64 : (PX >= 9 ? 128 : 64);
#ifndef PPP_PA_MINWAVES_SMALL9
#define PPP_PA_MINWAVES_SMALL9 2
#endif
    static constexpr int MIN_WAVES_SMALL = PX >= 25 ? 1 : (PX >= 9 ? PPP_PA_MINWAVES_SMALL9 : 4);
};
// Patch widths whose per-patch kernel reads thinning masks made beforehand (patch_graph_lcg_kernel)
// instead of running the generator: all 3-d widths.  How the masks reach the lanes decides whether
// that pays (256^3 / 9^3, S5 per volume; profiles/r04_z[h-p]_*): the generator 1219 ms; one 8-byte
// global load per chunk inside the pixel step 1286-1485 ms -- every s_waitcnt in this kernel is
// vmcnt(0), so the first mask load of a pixel also waits for the next row's 39 prefetch loads;
// staged through LDS beside the row (+2.5 KB: one workgroup less per CU) 1770-1900 ms; WITHOUT
// the register prefetch of the next row (it is loaded after the pixel step: +1.6 % by itself,
// 39 registers less) and the mask of a chunk requested while the chunk before it is worked on:
// 1046 + 33 ms for the masks.  7^3 (140^3): 72 -> 48 + 5 ms.  The 25-wide 2-d kernel is not
// generator-bound (235 -> 230 + 9 ms) and keeps it.
#ifndef PPP_PA_MASKS_MAX_PX
#define PPP_PA_MASKS_MAX_PX 9
#endif
static constexpr int PA_MASKS_MAX_PX = PPP_PA_MASKS_MAX_PX;
// experiments: 1 / 0 = the next row loaded after the pixel step / into registers during it for
// every width (default -1: after the step where masks are read); the mask of a chunk fetched at
// its use (0) or one chunk ahead (1)
#ifndef PPP_PA_NO_ROW_PREFETCH
#define PPP_PA_NO_ROW_PREFETCH -1
#endif
#ifndef PPP_PA_MASK_AHEAD
#define PPP_PA_MASK_AHEAD 1
#endif
static constexpr int PA_PAD = 8;       // floats of slack either side of the staged row (a masked row read overshoots by < PX)

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

template <typename T, int PX, int PA_THREADS>
__global__ void __launch_bounds__(PA_THREADS, PA_THREADS == PaCfg<PX>::THREADS ? PaCfg<PX>::MIN_WAVES
                                                                               : PaCfg<PX>::MIN_WAVES_SMALL)
    patch_graph_pa_kernel(const T *__restrict__ pred, const float *__restrict__ S,
                          const uint32_t *__restrict__ rows, const uint32_t *__restrict__ order,
                          const long long *__restrict__ group_start,
                          const long long *__restrict__ chunk_offsets, const int n_groups,
                          float *__restrict__ aff, const long long *__restrict__ drop_off,
                          const unsigned long long *__restrict__ drops, const Geo G) {
    extern __shared__ uint32_t lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int words = (G.C + 31) / 32;
    const int W = (2 * G.pz - 1) * G.wy * G.wx, Lc = (W - 1) / 2;
    const int WB = (W + 2 * PA_PAD + 3) & ~3;                     // floats per row buffer
    // staged floats per thread (the launcher checks W <= NST * PA_THREADS): a (2 PX - 1)^3 row --
    // for the wide kernel (25 x 25: 2-d patches only) a (2 PX - 1)^2 one.  (Sized for a cube it was
    // 1 839 floats per thread at PX = 25 with 64 threads: the staging array lived in scratch and
    // every pixel step walked 1 839 predicated iterations to move 38 floats -- what the "mask
    // algebra" of profiles/r04_zy_s5_p25_ablations.txt really was.)
    constexpr int NST = ((PX > 16 ? 1 : 2 * PX - 1) * (2 * PX - 1) * (2 * PX - 1) + PA_THREADS - 1) / PA_THREADS;
    constexpr int ROW_BUFS = PaCfg<PX>::ROW_BUFS;
    float *rowbuf = reinterpret_cast<float *>(lds_raw);           // [ROW_BUFS][WB]
    uint32_t *faw = lds_raw + ROW_BUFS * WB;                      // [words]
    uint32_t *fbw = faw + ((words + 3) & ~3);                     // [words][PA_THREADS]
    typedef unsigned long long u64;
    constexpr int RPC = 64 / PX;                         // candidate rows per 64-bit chunk
    constexpr int NCH = (PX + RPC - 1) / RPC;            // chunks (the launcher checks py <= NCH * RPC)

    // ---- The range / intersection / stored conditions of a whole (y2o, x2o) candidate plane
    // are evaluated at once as bit masks, bit (y2o - y_first) * PX + x2o, for chunks of RPC rows
    // (one chunk unless PX = 9).  Per-axis masks are expanded once per pixel row / pixel: EY*
    // repeat a y bit over the PX bits of its row, RX repeats the x mask in every row.
    constexpr uint32_t RM = (1u << PX) - 1u;
    auto expand_y = [&](uint32_t m, int c) -> u64 {
        u64 e = 0;
        for (int j = 0; j < RPC; ++j) e |= ((m >> (c * RPC + j)) & 1u) ? ((u64)RM << (PX * j)) : 0ull;
        return e;
    };
    // x mask repeated in every row, by doubling (rows beyond the chunk are harmless: every use
    // is ANDed with an EY mask, which is clear there)
    auto repeat_x = [&](uint32_t m) -> u64 {
#ifdef PPP_PA_REPEAT_BY_DOUBLING
        u64 e = m;
#pragma unroll
        for (int n = 1; n < RPC; n *= 2) e |= e << (PX * n);
        return e;
#else
        // one multiply by the constant with a 1 at the start of every row (the PX-bit fields
        // do not overlap, so there are no carries); rows >= RPC are not needed
        u64 rep = 0;
#pragma unroll
        for (int j = 0; j < RPC; ++j) rep |= 1ull << (PX * j);
        return (u64)m * rep;
#endif
    };

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
    // thinning decisions made beforehand by patch_graph_lcg_kernel (below): position of this
    // pair's masks in `drops`, < 0 = none (this kernel runs the LCG itself)
    constexpr bool MASKS = PX <= PA_MASKS_MAX_PX;
    constexpr bool NOPF = PPP_PA_NO_ROW_PREFETCH < 0 ? MASKS : PPP_PA_NO_ROW_PREFETCH != 0;
    long long soff = -1;
    if (live) {
        if (MASKS && drops != nullptr) soff = drop_off[pos];
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
    // the pixels of A that are in F_A are walked in raster order straight off the bit words
    // (scalar bit iterator: the word in an SGPR, s_ff1 / clear-lowest per pixel; no list in LDS)
    int n_u = 0;
    for (int w = 0; w < words; ++w) n_u += __popc(faw[w]);
    n_u = __builtin_amdgcn_readfirstlane(n_u);
    int it_w = -1;
    uint32_t it_m = 0;
    auto next_pixel = [&]() -> int {
        while (it_m == 0u) {
            ++it_w;
            it_m = it_w < words ? (uint32_t)__builtin_amdgcn_readfirstlane((int)faw[it_w]) : 1u;
        }
        const int r = it_w * 32 + __builtin_ctz(it_m);
        it_m &= it_m - 1u;
        return r;
    };

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
    const long long baseA = ((long long)(ay - G.by0)) * G.bX + (ax - G.bx0);        // (the slice is added per pixel)
    // (pixel indices are workgroup-uniform scalars: the row address arithmetic stays on the
    // scalar unit and the staging loads take the saddr + voffset form)
    auto row_of = [&](int r1) -> const float * {
        const int z1o = r1 / (G.py * PX), y1o = (r1 / PX) % G.py, x1o = r1 % PX;
        return S + (baseA + (long long)row_slice(G, az + z1o - G.rz) * sZ + (long long)(y1o - G.ry) * sY + (x1o - PX / 2)) * W;
    };
    float acc = 0.0f;
    unsigned fg_cnt = 0;
#ifdef PA_STATS
    unsigned long long st_0 = 0, st_1 = 0, st_2 = 0, st_3 = 0, st_4 = 0, st_5 = 0;
#endif

    // ---- stage the first row
    float st[NST];
    int r_next = n_u > 0 ? next_pixel() : 0;
    if (n_u > 0) {
        const float *src = row_of(r_next);
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int e = tid + i * PA_THREADS;
            if (e < W) rowbuf[PA_PAD + e] = src[e];
        }
    }
    __syncthreads();

    int prev_z1o = -1, prev_y1o = -1;
    AxisMasks mz = axis_masks(dz, 0, G.pz), my = axis_masks(dy, 0, G.py);
    // expanded y masks per chunk: cached across the pixels of a row of A while a plane has few
    // chunks (3-d patches: 1-2); a 25-wide 2-d patch has 13 chunks of two rows -- 156 registers
    // of masks -- and expands them inside the chunk instead (a dozen instructions per chunk)
    constexpr bool EY_CACHED = NCH <= 2;
    constexpr int NEY = EY_CACHED ? NCH : 1;
    u64 EYf[NEY], EYbk[NEY], EYpos[NEY], EYzero[NEY], EYst[NEY], EYin[NEY];
#pragma unroll
    for (int c = 0; c < NEY; ++c) EYf[c] = EYbk[c] = EYpos[c] = EYzero[c] = EYst[c] = EYin[c] = 0ull;
    for (int k = 0; k < n_u; ++k) {
        const int r1 = r_next;
        const int z1o = r1 / (G.py * PX), y1o = (r1 / PX) % G.py, x1o = r1 % PX;
        const float *cur = rowbuf + (ROW_BUFS == 2 ? (k & 1) * WB : 0) + PA_PAD;
        // ---- fetch the next row into registers while this one is consumed
        const bool more = k + 1 < n_u;
        if (more) {
            r_next = next_pixel();
            if constexpr (!NOPF) {
                const float *src = row_of(r_next);
#pragma unroll
                for (int i = 0; i < NST; ++i) {
                    const int e = tid + i * PA_THREADS;
                    st[i] = e < W ? src[e] : 0.0f;
                }
            }
        }
        // union of the lanes' candidate ranges: -p <= d + b - a <= p (wave-uniform)
        const int z_lo = max(0, z1o - dz_hi - G.pz), z_hi = min(G.pz - 1, z1o - dz_lo + G.pz);
        const int y_lo = max(0, y1o - dy_hi - G.py), y_hi = min(G.py - 1, y1o - dy_lo + G.py);
        if (wave_live && z_lo <= z_hi && y_lo <= y_hi) {
            PA_STAT(0, lane == 0 ? 1 : 0);
            PA_STAT(5, live ? 1 : 0);
            if (z1o != prev_z1o) { mz = axis_masks(dz, z1o, G.pz); prev_z1o = z1o; }
            if (y1o != prev_y1o) {
                my = axis_masks(dy, y1o, G.py);   // (bits >= py are clear: the masks stop at py)
                prev_y1o = y1o;
                if constexpr (EY_CACHED) {
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        EYf[c] = expand_y(my.f, c); EYbk[c] = expand_y(my.bk, c); EYpos[c] = expand_y(my.pos, c);
                        EYzero[c] = expand_y(my.zero, c); EYst[c] = expand_y(my.st, c); EYin[c] = expand_y(my.in, c);
                    }
                }
            }
            const int q0x = dx - x1o;
            const bool in_b = abs(x1o - PX / 2 - dx) <= PX / 2 && abs(y1o - G.ry - dy) <= G.ry &&
                              abs(z1o - G.rz - dz) <= G.rz;
            // (a workgroup table of these masks in LDS, indexed by q0x, was measured: fewer
            // VALU instructions but 4 % slower -- five dependent LDS reads at the head of
            // every pixel step)
            // this pixel's block of precomputed masks: [i1][plane of the intersection][chunk]
            long long blk = 0;
            const bool has_blk = MASKS && soff >= 0 && in_b;
            if (has_blk) {
                const int nzl = G.pz - abs(dz), nyl = G.py - abs(dy), nxl = PX - abs(dx);
                const int i1 = ((z1o - max(dz, 0)) * nyl + (y1o - max(dy, 0))) * nxl + (x1o - max(dx, 0));
                blk = soff + ((long long)i1 * nzl - max(-dz, 0)) * NCH;   // + z2o * NCH + chunk
            }
            const AxisMasks mx = axis_masks(dx, x1o, PX);
            const u64 RXf = repeat_x(mx.f), RXbk = repeat_x(mx.bk), RXst = repeat_x(mx.st),
                      RXin = in_b ? repeat_x(mx.in) : 0ull, RXnn = repeat_x(mx.pos | mx.zero),
                      RXzero = repeat_x(mx.zero);
#if PPP_PA_MASK_AHEAD
            // (planes outside the lane's intersection are not fetched)
            auto fetch = [&](int z, int c) -> u64 {
                return has_blk && z < G.pz && ((mz.in >> z) & 1u) ? drops[blk + z * NCH + c] : 0ull;
            };
            u64 pend = fetch(z_lo, 0);
#endif
            for (int z2o = z_lo; z2o <= z_hi; ++z2o) {
                const uint32_t zb = 1u << z2o;
                const bool z_f = mz.f & zb, z_bk = mz.bk & zb, z_pos = mz.pos & zb, z_zero = mz.zero & zb,
                           z_st = mz.st & zb, z_in = mz.in & zb;
                if (__ballot(z_f || z_bk || z_in) == 0ull) {
#if PPP_PA_MASK_AHEAD
                    pend = fetch(z2o + 1, 0);
#endif
                    continue;
                }
                const int qz = mz.q0 + z2o;
                if constexpr (PX > 16) {
                    // ---- wide windows (the 25 x 25 2-d patches): ONE candidate row per step.  A row
                    // of PX <= 32 candidates is one 32-bit word; the y conditions are single bits of
                    // the per-axis masks, so a row costs a dozen 32-bit operations -- the chunk
                    // form below expands six y masks into 64-bit words for every two rows (13 chunks
                    // per plane and pixel at 25 x 25: what this kernel spent its time on,
                    // profiles/r04_zy_s5_p25_ablations.txt).  Same candidates, same order (rows
                    // ascending, x ascending inside a row), same LCG stream.
                    const int idx0 = Lc + (qz * G.wy + my.q0) * G.wx + q0x;
                    const uint32_t xin = in_b ? mx.in : 0u;
#pragma unroll 1
                    for (int y2o = y_lo; y2o <= y_hi; ++y2o) {
                        const uint32_t yb1 = 1u << y2o;
                        const bool y_pos = my.pos & yb1, y_zero = my.zero & yb1;
                        const uint32_t fwd = z_pos ? RM : (z_zero ? (y_pos ? RM : (y_zero ? (mx.pos | mx.zero) : 0u)) : 0u);
                        const uint32_t range = ((z_f && (my.f & yb1)) ? (mx.f & fwd) : 0u) |
                                               ((z_bk && (my.bk & yb1)) ? (mx.bk & ~fwd & RM) : 0u);
                        const uint32_t inter = (z_in && (my.in & yb1)) ? xin : 0u;
                        if (__ballot((range | inter) != 0u) == 0ull) continue;
                        PA_STAT(1, lane == 0 ? 1 : 0);
                        uint32_t stored = (z_st && (my.st & yb1)) ? (mx.st & range) : 0u;
                        if (z_zero && y_zero) stored &= ~mx.zero;
                        // foreground bits of patch B on this candidate row
                        uint32_t valid;
                        {
                            const int o = (z2o * G.py + y2o) * PX, w0 = o >> 5, sh = o & 31;
                            const uint32_t lo = fbw[w0 * PA_THREADS + tid];
                            const uint32_t hi = w0 + 1 < words ? fbw[(w0 + 1) * PA_THREADS + tid] : 0u;
                            valid = (uint32_t)(((((u64)hi) << 32) | lo) >> sh) & RM;
                        }
                        const uint32_t hit_row = inter & valid;
                        if (__ballot(hit_row != 0u) != 0ull) {
                            PA_STAT(4, lane == 0 ? 1 : 0);
                            uint32_t drop = 0;
#pragma unroll
                            for (int t = 0; t < PX; ++t) {
                                const uint32_t xb = 1u << t;
                                const uint32_t nxt = rnd * 1103515245U;
                                const bool hit = (hit_row & xb) != 0u;
                                rnd = hit ? nxt : rnd;
                                if (hit && nxt >= 858993441u) drop |= xb;      // (see the chunk form)
                            }
                            valid &= ~drop;
                        }
                        fg_cnt += __popc(range & valid);
                        const uint32_t rb = stored & valid;
                        if (__ballot(rb != 0u) == 0ull) continue;
                        PA_STAT(2, lane == 0 ? 1 : 0);
                        PA_STAT(3, __popc(rb));
                        const float *rowq = cur + (rb != 0u ? idx0 + y2o * G.wx : 0);
#pragma unroll
                        for (int t = 0; t < PX; ++t) {
                            const int sel = ((int)(rb << (31 - t))) >> 31;
                            acc += __int_as_float(__float_as_int(rowq[t]) & sel);
                        }
                    }
                    continue;
                }
                // one chunk of candidate rows; M = mask word: 64 bits, or 32 bits for a last
                // chunk of few rows (9^3: rows 7-8, 18 bits) -- half the mask arithmetic
                auto chunk = [&](auto mtag, const int c) {
                    using M = decltype(mtag);
#if PPP_PA_MASK_AHEAD == 1
                    const u64 dropped = pend;
                    pend = c + 1 < NCH ? fetch(z2o, c + 1) : fetch(z2o + 1, 0);
#endif
                    const int c_first = c * RPC;                       // first row of the chunk
                    const int c_rows = min(RPC, G.py - c_first);
                    if (c_rows <= 0) return;
                    const int ya = max(y_lo, c_first), yb = min(y_hi, c_first + c_rows - 1);
                    if (ya > yb) return;
                    // forward orientation (pixel z1 before z2 in raster order <=> q >= 0
                    // lexicographically) and the position of q == 0 (never stored)
                    const u64 eyf = EY_CACHED ? EYf[EY_CACHED ? c : 0] : expand_y(my.f, c);
                    const u64 eybk = EY_CACHED ? EYbk[EY_CACHED ? c : 0] : expand_y(my.bk, c);
                    const u64 eypos = EY_CACHED ? EYpos[EY_CACHED ? c : 0] : expand_y(my.pos, c);
                    const u64 eyzero = EY_CACHED ? EYzero[EY_CACHED ? c : 0] : expand_y(my.zero, c);
                    const u64 eyst = EY_CACHED ? EYst[EY_CACHED ? c : 0] : expand_y(my.st, c);
                    const u64 eyin = EY_CACHED ? EYin[EY_CACHED ? c : 0] : expand_y(my.in, c);
                    const M fwd = z_pos ? (M)~(M)0 : (z_zero ? (M)((M)eypos | ((M)eyzero & (M)RXnn)) : (M)0);
                    const M range = (z_f ? (M)((M)eyf & (M)RXf & fwd) : (M)0) |
                                    (z_bk ? (M)((M)eybk & (M)RXbk & (M)~fwd) : (M)0);
                    const M inter = z_in ? (M)((M)eyin & (M)RXin) : (M)0;
                    M stored = z_st ? (M)((M)eyst & (M)RXst & range) : (M)0;
                    if (z_zero) stored &= (M) ~((M)eyzero & (M)RXzero);
                    if (__ballot((M)(range | inter) != (M)0) == 0ull) return;
                    PA_STAT(1, lane == 0 ? 1 : 0);
                    // foreground bits of patch B on these candidate rows
                    M valid;
                    {
                        const int nb = c_rows * PX;
                        const int o = (z2o * G.py + c_first) * PX, w0 = o >> 5, sh = o & 31;
                        const uint32_t lo = fbw[w0 * PA_THREADS + tid];
                        const uint32_t mi = w0 + 1 < words ? fbw[(w0 + 1) * PA_THREADS + tid] : 0u;
                        const uint32_t hi = w0 + 2 < words ? fbw[(w0 + 2) * PA_THREADS + tid] : 0u;
                        valid = (M)(((((u64)mi << 32) | lo) >> sh) | (sh ? ((u64)hi << (64 - sh)) : 0ull));
                        valid &= nb >= (int)(8 * sizeof(M)) ? (M)~(M)0 : (M)(((M)1 << nb) - (M)1);
                    }
#ifdef PPP_PA_ABL_NOLCG   // timing experiment only: results are wrong
                    if (false) {
#else
                    if (__ballot(inter != (M)0) != 0ull) {
#endif
                        // thinning inside the patch intersection: the LCG advances on every
                        // foreground candidate of the intersection, in candidate order --
                        // either done beforehand (one mask of dropped candidates per chunk) ...
                        M lcg_hits = inter & valid;
                        if (has_blk) {
#if PPP_PA_MASK_AHEAD == 1
                            valid &= (M) ~((M)dropped & inter);
#else
                            if (inter != (M)0) valid &= (M) ~((M)drops[blk + z2o * NCH + c] & inter);
#endif
                            lcg_hits = (M)0;
                        }
                        // ... or here, for the lanes without precomputed masks
                        if (__ballot(lcg_hits != (M)0) != 0ull)
                        for (int y2o = ya; y2o <= yb; ++y2o) {
                            const int bp = PX * (y2o - c_first);
                            const uint32_t hit_row = (uint32_t)(lcg_hits >> bp) & RM;
                            if (__ballot(hit_row != 0u) == 0ull) continue;
                            PA_STAT(4, lane == 0 ? 1 : 0);
                            uint32_t drop = 0;
#pragma unroll
                            for (int t = 0; t < PX; ++t) {
                                const uint32_t xb = 1u << t;
                                const uint32_t nxt = rnd * 1103515245U;
                                const bool hit = (hit_row & xb) != 0u;
                                rnd = hit ? nxt : rnd;
                                // (float)nxt / 2^32 > 0.2 (compared in double)  <=>  nxt >= 858993441:
                                // the conversion is monotone and 858993440 is the tie that rounds
                                // down to 0.2f's predecessor * 2^32 (checked exhaustively around it)
                                if (hit && nxt >= 858993441u) drop |= xb;
                            }
                            valid &= (M) ~((M)drop << bp);
                        }
                    }
                    fg_cnt += sizeof(M) == 8 ? __popcll((u64)(range & valid)) : __popc((uint32_t)(range & valid));
                    const M add = stored & valid;
                    if (__ballot(add != (M)0) == 0ull) return;
#ifdef PPP_PA_ABL_NOADD   // timing experiment only: results are wrong
                    if (add != (M)0) acc += 1.0f;
                    return;
#endif
                    // this lane's offset into the staged row for candidate (y2o = 0, x2o = 0)
                    const int idx0 = Lc + (qz * G.wy + my.q0) * G.wx + q0x;
// (measured and rejected: 73 -> 79 ms on the thinned list, 540 -> 640 ms on the dense one --
// the extra PX registers cost spills at the 64 / 128-VGPR budgets and every row is read)
#ifndef PPP_PA_ROW_PIPELINE
#define PPP_PA_ROW_PIPELINE 0
#endif
                    if constexpr (PPP_PA_ROW_PIPELINE && PX <= 9) {
                        // The PX values of a row feed a chain of PX dependent adds; with the LDS
                        // reads issued right in front of them a wave waits out the LDS latency
                        // once per row (45 % of the wave-cycles parked at 3-4 waves per SIMD).
                        // So the reads run ONE ROW AHEAD of the adds: row y2o + 1 is in flight
                        // while row y2o is summed.  A row nobody needs is still read (lanes
                        // without work point at slot 0): LDS bandwidth is not what is short.
                        float nx[PX];
                        auto fetch = [&](int y2o, float (&dst)[PX]) {
                            const uint32_t rbn = (uint32_t)(add >> (PX * (y2o - c_first))) & RM;
                            const float *rowq = cur + (rbn != 0u ? idx0 + y2o * G.wx : 0);
#pragma unroll
                            for (int t = 0; t < PX; ++t) dst[t] = rowq[t];
                        };
                        fetch(ya, nx);
                        for (int y2o = ya; y2o <= yb; ++y2o) {
                            float cv[PX];
#pragma unroll
                            for (int t = 0; t < PX; ++t) cv[t] = nx[t];
                            if (y2o < yb) fetch(y2o + 1, nx);
                            const uint32_t rb = (uint32_t)(add >> (PX * (y2o - c_first))) & RM;
                            if (__ballot(rb != 0u) == 0ull) continue;
                            PA_STAT(2, lane == 0 ? 1 : 0);
                            PA_STAT(3, __popc(rb));
#pragma unroll
                            for (int t = 0; t < PX; ++t) {
                                const int sel = ((int)(rb << (31 - t))) >> 31;
                                acc += __int_as_float(__float_as_int(cv[t]) & sel);
                            }
                        }
                    } else {
                    for (int y2o = ya; y2o <= yb; ++y2o) {
                        const uint32_t rb = (uint32_t)(add >> (PX * (y2o - c_first))) & RM;
                        if (__ballot(rb != 0u) == 0ull) continue;
                        PA_STAT(2, lane == 0 ? 1 : 0);
                        PA_STAT(3, __popc(rb));
                        // lanes with nothing to add on this row may point anywhere: they read
                        // slot 0 (one broadcast address; what they read is masked to zero)
                        const float *rowq = cur + (rb != 0u ? idx0 + y2o * G.wx : 0);
#pragma unroll
                        for (int t = 0; t < PX; ++t) {
                            // acc += bit t of rb ? rowq[t] : 0.0f
                            const int sel = ((int)(rb << (31 - t))) >> 31;
                            acc += __int_as_float(__float_as_int(rowq[t]) & sel);
                        }
                    }
                    }
                };
                // the last chunk of the 9-row planes holds (PX - (NCH-1) RPC) rows
                constexpr bool LAST32 = NCH > 1 && (PX - (NCH - 1) * RPC) * PX <= 32;
                if constexpr (NCH <= 2) {
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        if (LAST32 && c == NCH - 1) chunk(uint32_t{}, c);
                        else chunk(u64{}, c);
                    }
                } else {
                    // many chunks per plane (wide 2-d patches): a real loop, one chunk body
#pragma unroll 1
                    for (int c = 0; c < NCH; ++c) chunk(u64{}, c);
                }
            }
        }
        // ---- publish the next row
        if (ROW_BUFS == 1) __syncthreads();   // everybody is done reading the only buffer
        if (more) {
            float *dst = rowbuf + (ROW_BUFS == 2 ? ((k + 1) & 1) * WB : 0) + PA_PAD;
            if constexpr (NOPF) {
                const float *src = row_of(r_next);
#pragma unroll
                for (int i = 0; i < NST; ++i) {
                    const int e = tid + i * PA_THREADS;
                    st[i] = e < W ? src[e] : 0.0f;
                }
            }
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int e = tid + i * PA_THREADS;
                if (e < W) dst[e] = st[i];
            }
        }
        __syncthreads();
    }
#ifdef PA_STATS
    atomicAdd(&pa_stats[0], st_0); atomicAdd(&pa_stats[1], st_1); atomicAdd(&pa_stats[2], st_2);
    atomicAdd(&pa_stats[3], st_3); atomicAdd(&pa_stats[4], st_4); atomicAdd(&pa_stats[5], st_5);
#endif
    if (live) aff[row_id] = G.norm_aff ? acc / (float)(fg_cnt > 1u ? fg_cnt : 1u) : acc;
}

#ifdef PA_STATS
extern "C" void ppp_pa_stats(unsigned long long *out) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(pa_stats), sizeof(pa_stats));
}
#endif


// ---- The thinning decisions of the patch intersection, beforehand.
//
// computePatchGraph.cu:75-86 advances the pair's LCG (rnd *= 1103515245) on every combination
// (pixel z1 of A, pixel z2 of B) of foreground pixels that both lie in the intersection of the two
// windows, in loop order, and drops the combination when rnd / 2^32 > 0.2.  The decisions need
// the two foreground bit sets and the pair's seed -- not the consensus.  Inside the per-patch
// kernel above they cost half of its time: ~8 of a wave's 64 pair rows overlap their patch A at
// all, and the wave walks the union of their candidate rows with nine dependent 32-bit multiplies
// per row.  Here a lane owns one overlapping pair, the lanes of a wave have (nearly) the same
// patch offset (the caller sorts them by offset), every lane walks its OWN foreground pixels of A
// inside the intersection, and the rows of B inside the intersection are the same for the whole
// wave.  Output per pair: for every intersection pixel i1 of A (raster index inside the
// intersection box), every plane z2 of the intersection in B's window and every 64-bit chunk of
// candidate rows of that plane (the chunk layout of the kernel above: bit (y2o - first row) * PX
// + x2o), the mask of DROPPED candidates.  The kernel above reads one mask per chunk instead of
// running the generator.
template <typename T, int PX>
__global__ void __launch_bounds__(64)
    patch_graph_lcg_kernel(const T *__restrict__ pred, const uint32_t *__restrict__ rows,
                           const uint32_t *__restrict__ order, const long long *__restrict__ lcg_pos,
                           const long long n, const long long *__restrict__ drop_off,
                           unsigned long long *__restrict__ drops, const Geo G) {
    typedef unsigned long long u64;
    extern __shared__ uint32_t lds_raw[];
    const int lane = threadIdx.x;
    const int words = (G.C + 31) / 32;
    uint32_t *fa = lds_raw;                 // [words][64]  F_A inside the intersection (A's window raster)
    uint32_t *fb = lds_raw + words * 64;    // [words + 2][64]  F_B inside the intersection (B's window raster)
    constexpr uint32_t RM = (1u << PX) - 1u;
    constexpr int RPC = 64 / PX;
    constexpr int NCH = (PX + RPC - 1) / RPC;
    const long long i = (long long)blockIdx.x * 64 + lane;
    bool live = i < n;
    long long soff = -1;
    long long pos = 0;
    if (live) { pos = lcg_pos[i]; soff = drop_off[pos]; live = soff >= 0; }
    int az = 0, ay = 0, ax = 0, bz = 0, by = 0, bx = 0;
    if (live) {
        const uint32_t *rw = rows + (size_t)order[pos] * 6;
        az = (int)rw[0]; ay = (int)rw[1]; ax = (int)rw[2];
        bz = (int)rw[3]; by = (int)rw[4]; bx = (int)rw[5];
    }
    const int dz = bz - az, dy = by - ay, dx = bx - ax;
    // intersection box: n per axis, first window coordinate in A (lo1) and in B (lo2)
    const int nz = live ? G.pz - abs(dz) : 0, ny = live ? G.py - abs(dy) : 0, nx = live ? PX - abs(dx) : 0;
    const int z1lo = max(dz, 0), y1lo = max(dy, 0), x1lo = max(dx, 0);
    const int z2lo = max(-dz, 0), y2lo = max(-dy, 0), x2lo = max(-dx, 0);
    uint32_t rnd = (uint32_t)(az + G.oz) * (uint32_t)(bz + G.oz) * (uint32_t)(ay + G.oy) *
                   (uint32_t)(by + G.oy) * (uint32_t)(ax + G.ox) * (uint32_t)(bx + G.ox);
    for (int w = 0; w < words; ++w) fa[w * 64 + lane] = 0u;
    for (int w = 0; w < words + 2; ++w) fb[w * 64 + lane] = 0u;
    {
        const T *mid = pred + (long long)G.mid * G.V;
        const long long la = vox(G, az, ay, ax), lb = vox(G, bz, by, bx);
        const int n_i = nz > 0 && ny > 0 && nx > 0 ? nz * ny * nx : 0;
        for (int t = 0; t < n_i; ++t) {
            const int iz = t / (ny * nx), iy = (t / nx) % ny, ix = t % nx;
            {
                const int zo = z1lo + iz, yo = y1lo + iy, xo = x1lo + ix;
                const int r = (zo * G.py + yo) * PX + xo;
                const bool on = ldf(mid, vox(G, az + zo - G.rz, ay + yo - G.ry, ax + xo - PX / 2)) > G.th_gt &&
                                ldf(pred, (long long)r * G.V + la) > G.th_gt;
                if (on) fa[(r >> 5) * 64 + lane] |= 1u << (r & 31);
            }
            {
                const int zo = z2lo + iz, yo = y2lo + iy, xo = x2lo + ix;
                const int r = (zo * G.py + yo) * PX + xo;
                const bool on = ldf(mid, vox(G, bz + zo - G.rz, by + yo - G.ry, bx + xo - PX / 2)) > G.th_gt &&
                                ldf(pred, (long long)r * G.V + lb) > G.th_gt;
                if (on) fb[(r >> 5) * 64 + lane] |= 1u << (r & 31);
            }
        }
    }
    // (a lane only touches its own LDS column: no barrier)
    for (int w = 0; w < words; ++w) {
        uint32_t m = fa[w * 64 + lane];
        while (__ballot(m != 0u) != 0ull) {
            const bool act = m != 0u;
            const int r1 = w * 32 + (act ? __builtin_ctz(m) : 0);
            m &= m - 1u;
            const int z1o = r1 / (G.py * PX), y1o = (r1 / PX) % G.py, x1o = r1 % PX;
            const int i1 = ((z1o - z1lo) * ny + (y1o - y1lo)) * nx + (x1o - x1lo);
            const long long blk = soff + (long long)i1 * nz * NCH;
            for (int z2o = 0; z2o < G.pz; ++z2o) {
                const bool inz = act && z2o >= z2lo && z2o < z2lo + nz;
                if (__ballot(inz) == 0ull) continue;
#pragma unroll 1
                for (int c = 0; c < NCH; ++c) {
                    const int c_first = c * RPC;
                    const int c_rows = min(RPC, G.py - c_first);
                    if (c_rows <= 0) break;
                    u64 hits;
                    {
                        const int nb = c_rows * PX;
                        const int o = (z2o * G.py + c_first) * PX, w0 = o >> 5, sh = o & 31;
                        const uint32_t lo = fb[w0 * 64 + lane], mi = fb[(w0 + 1) * 64 + lane],
                                       hi = fb[(w0 + 2) * 64 + lane];
                        hits = ((((u64)mi << 32) | lo) >> sh) | (sh ? ((u64)hi << (64 - sh)) : 0ull);
                        hits &= nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);
                        if (!inz) hits = 0ull;
                    }
                    u64 drop = 0ull;
                    if (__ballot(hits != 0ull) != 0ull) {
                        for (int y = 0; y < c_rows; ++y) {
                            const uint32_t hit_row = (uint32_t)(hits >> (PX * y)) & RM;
                            if (__ballot(hit_row != 0u) == 0ull) continue;
                            uint32_t drow = 0;
#pragma unroll
                            for (int t = 0; t < PX; ++t) {
                                const uint32_t xb = 1u << t;
                                const uint32_t nxt = rnd * 1103515245U;
                                const bool hit = (hit_row & xb) != 0u;
                                rnd = hit ? nxt : rnd;
                                // (float)nxt / 2^32 > 0.2 in double  <=>  nxt >= 858993441 (see above)
                                if (hit && nxt >= 858993441u) drow |= xb;
                            }
                            drop |= (u64)drow << (PX * y);
                        }
                    }
                    if (inz) drops[blk + (z2o - z2lo) * NCH + c] = drop;
                }
            }
        }
    }
}

// ---- the same masks, a WAVE per pair (round 5).
//
// With a lane per pair a launch is a few hundred waves whose time is their heaviest pair's (the
// work of a pair grows with the square of its window intersection: 848 waves per launch at 512^3,
// the longest four times the average, 16 ms with the chip nearly empty), and every lane walks all
// 81 x 9 candidate positions of a pixel of A with a dependent multiply each.  The generator's state
// before A's j-th foreground pixel is seed * g^j with g = a^nB (nB = B's foreground pixels inside
// the intersection: every one of them advances it once), so A's pixels are INDEPENDENT: here the
// 64 lanes of a wave take 64 foreground pixels of A at a time, the two bit sets are gathered by
// the whole wave, B's candidates -- the same for every lane -- are walked set bit by set bit
// (one multiply per foreground pixel of B, none for the others), and a launch is one wave per
// pair: tens of thousands of waves of similar length.  Same masks, bit for bit.
__device__ __forceinline__ uint32_t lcg_pow(uint32_t base, uint32_t e) {      // base^e mod 2^32
    uint32_t r = 1u;
    while (e) { if (e & 1u) r *= base; base *= base; e >>= 1; }
    return r;
}
static constexpr int LCGW_WAVES = 4;
template <typename T, int PX>
__global__ void __launch_bounds__(64 * LCGW_WAVES)
    patch_graph_lcg_wave_kernel(const T *__restrict__ pred, const uint32_t *__restrict__ rows,
                                const uint32_t *__restrict__ order, const long long *__restrict__ lcg_pos,
                                const long long n, const long long *__restrict__ drop_off,
                                unsigned long long *__restrict__ drops, const Geo G) {
    typedef unsigned long long u64;
    extern __shared__ uint32_t lds_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int words = (G.C + 31) / 32;
    uint32_t *fa = lds_raw + wave * (2 * words + 2);      // F_A inside the intersection (A's window raster)
    uint32_t *fb = fa + words;                            // [words + 2]  F_B (B's window raster)
    constexpr int RPC = 64 / PX;
    constexpr int NCH = (PX + RPC - 1) / RPC;
    const long long i = (long long)blockIdx.x * LCGW_WAVES + wave;
    if (i >= n) return;                                   // (whole waves: no workgroup barrier below)
    const long long pos = lcg_pos[i];
    const long long soff = drop_off[pos];
    if (soff < 0) return;
    const uint32_t *rw = rows + (size_t)order[pos] * 6;
    const int az = (int)rw[0], ay = (int)rw[1], ax = (int)rw[2], bz = (int)rw[3], by = (int)rw[4], bx = (int)rw[5];
    const int dz = bz - az, dy = by - ay, dx = bx - ax;
    const int nz = G.pz - abs(dz), ny = G.py - abs(dy), nx = PX - abs(dx);
    if (nz <= 0 || ny <= 0 || nx <= 0) return;
    const int z1lo = max(dz, 0), y1lo = max(dy, 0), x1lo = max(dx, 0);
    const int z2lo = max(-dz, 0), y2lo = max(-dy, 0), x2lo = max(-dx, 0);
    const uint32_t seed = (uint32_t)(az + G.oz) * (uint32_t)(bz + G.oz) * (uint32_t)(ay + G.oy) *
                          (uint32_t)(by + G.oy) * (uint32_t)(ax + G.ox) * (uint32_t)(bx + G.ox);
    for (int w = lane; w < 2 * words + 2; w += 64) fa[w] = 0u;
    __builtin_amdgcn_wave_barrier();
    {
        const T *mid = pred + (long long)G.mid * G.V;
        const long long la = vox(G, az, ay, ax), lb = vox(G, bz, by, bx);
        const int n_i = nz * ny * nx;
        for (int t = lane; t < n_i; t += 64) {
            const int iz = t / (ny * nx), iy = (t / nx) % ny, ix = t % nx;
            const int ra = ((z1lo + iz) * G.py + (y1lo + iy)) * PX + (x1lo + ix);
            const int rb = ((z2lo + iz) * G.py + (y2lo + iy)) * PX + (x2lo + ix);
            const float ma = ldf(mid, vox(G, az + z1lo + iz - G.rz, ay + y1lo + iy - G.ry, ax + x1lo + ix - PX / 2));
            const float pa = ldf(pred, (long long)ra * G.V + la);
            const float mb = ldf(mid, vox(G, bz + z2lo + iz - G.rz, by + y2lo + iy - G.ry, bx + x2lo + ix - PX / 2));
            const float pb = ldf(pred, (long long)rb * G.V + lb);
            if (ma > G.th_gt && pa > G.th_gt) atomicOr(&fa[ra >> 5], 1u << (ra & 31));
            if (mb > G.th_gt && pb > G.th_gt) atomicOr(&fb[rb >> 5], 1u << (rb & 31));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int nA = 0, nB = 0;
    for (int w = 0; w < words; ++w) { nA += __popc(fa[w]); nB += __popc(fb[w]); }
    const uint32_t g = lcg_pow(1103515245U, (uint32_t)nB);
    for (int j0 = 0; j0 < nA; j0 += 64) {
        const int j = j0 + lane;
        const bool act = j < nA;
        // A's j-th foreground pixel (window raster order)
        int r1 = 0;
        {
            int left = act ? j : 0;
            for (int w = 0; w < words; ++w) {
                const uint32_t m = fa[w];
                const int c = __popc(m);
                if (left < c) {
                    uint32_t mm = m;
                    for (int k = 0; k < left; ++k) mm &= mm - 1u;
                    r1 = w * 32 + __builtin_ctz(mm);
                    break;
                }
                left -= c;
            }
        }
        const int z1o = r1 / (G.py * PX), y1o = (r1 / PX) % G.py, x1o = r1 % PX;
        const int i1 = ((z1o - z1lo) * ny + (y1o - y1lo)) * nx + (x1o - x1lo);
        const long long blk = soff + (long long)i1 * nz * NCH;
        uint32_t rnd = seed * lcg_pow(g, (uint32_t)(act ? j : 0));
        for (int z2o = z2lo; z2o < z2lo + nz; ++z2o) {
#pragma unroll 1
            for (int c = 0; c < NCH; ++c) {
                const int c_first = c * RPC;
                const int c_rows = min(RPC, G.py - c_first);
                if (c_rows <= 0) break;
                const int nb = c_rows * PX;
                const int o = (z2o * G.py + c_first) * PX, w0 = o >> 5, sh = o & 31;
                const uint32_t lo = fb[w0], mi = fb[w0 + 1], hi = fb[w0 + 2];
                u64 hits = ((((u64)mi << 32) | lo) >> sh) | (sh ? ((u64)hi << (64 - sh)) : 0ull);
                hits &= nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);
                hits = (u64)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)hits) |
                       ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(hits >> 32)) << 32);
                u64 drop = 0ull;
                while (hits) {                              // (the same bits for every lane)
                    const int b = __builtin_ctzll(hits);
                    hits &= hits - 1ull;
                    rnd *= 1103515245U;
                    // (float)rnd / 2^32 > 0.2 in double  <=>  rnd >= 858993441 (see above)
                    if (rnd >= 858993441u) drop |= 1ull << b;
                }
                if (act) drops[blk + (z2o - z2lo) * NCH + c] = drop;
            }
        }
    }
}

// u64 words of precomputed masks for a pair with patch offset (dz, dy, dx); 0 = the windows do
// not intersect
long long patch_graph_lcg_words(const Geo &G, int dz, int dy, int dx) {
    const int nz = G.pz - abs(dz), ny = G.py - abs(dy), nx = G.px - abs(dx);
    if (nz <= 0 || ny <= 0 || nx <= 0 || G.px > PA_MASKS_MAX_PX) return 0;
    const int rpc = 64 / G.px, nch = (G.px + rpc - 1) / rpc;
    return (long long)nz * ny * nx * nz * nch;
}

template <typename T, int PX>
static hipError_t launch_lcg(const T *pred, const uint32_t *rows, const uint32_t *order, const long long *lcg_pos,
                             long long n, const long long *drop_off, unsigned long long *drops, const Geo &G,
                             hipStream_t s) {
    const int words = (G.C + 31) / 32;
    const size_t lds = (size_t)(2 * words + 2) * 64 * 4;
    // a wave per pair (PPP_PA_LCG_WAVE=0: a lane per pair)
    static EnvSwitch wave_sw("PPP_PA_LCG_WAVE");
    if (!(wave_sw.get() && wave_sw.get()[0] == '0') &&
        !grid_too_big((unsigned long long)((n + LCGW_WAVES - 1) / LCGW_WAVES), 64 * LCGW_WAVES)) {
        const size_t ldsw = (size_t)(2 * words + 2) * LCGW_WAVES * 4;
        patch_graph_lcg_wave_kernel<T, PX><<<dim3((unsigned)((n + LCGW_WAVES - 1) / LCGW_WAVES)), dim3(64 * LCGW_WAVES), ldsw, s>>>(
            pred, rows, order, lcg_pos, n, drop_off, drops, G);
        return hipGetLastError();
    }
    patch_graph_lcg_kernel<T, PX><<<dim3((unsigned)((n + 63) / 64)), dim3(64), lds, s>>>(
        pred, rows, order, lcg_pos, n, drop_off, drops, G);
    return hipGetLastError();
}

hipError_t launch_patch_graph_lcg(const void *pred, int dtype, const uint32_t *rows, const uint32_t *order,
                                  const long long *lcg_pos, long long n, const long long *drop_off,
                                  unsigned long long *drops, const Geo &G, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    // (the shapes the per-patch kernel takes)
    if (patch_graph_pa_chunk(G, false) == 0 ||
        G.py > (G.px + 64 / G.px - 1) / (64 / G.px) * (64 / G.px) || G.pz > 32)
        return hipErrorNotSupported;
    const int words = (G.C + 31) / 32;
    if ((size_t)(2 * words + 2) * 64 * 4 > 64 * 1024 || grid_too_big((unsigned long long)((n + 63) / 64), 64))
        return hipErrorNotSupported;
#define PPP_LCG_CASE(P)                                                                                  \
    case P:                                                                                              \
        return dtype == PPP_F16 ? launch_lcg<__half, P>((const __half *)pred, rows, order, lcg_pos, n, drop_off, drops, G, s) \
                                : launch_lcg<float, P>((const float *)pred, rows, order, lcg_pos, n, drop_off, drops, G, s);
    switch (G.px) {
        PPP_LCG_CASE(3)
        PPP_LCG_CASE(5)
        PPP_LCG_CASE(7)
        PPP_LCG_CASE(9)
        PPP_LCG_CASE(25)
    default:
        return hipErrorNotSupported;
    }
#undef PPP_LCG_CASE
}

int patch_graph_pa_chunk(const Geo &G, bool small) {
    switch (G.px) {
    case 3: return small ? PaCfg<3>::THREADS_SMALL : PaCfg<3>::THREADS;
    case 5: return small ? PaCfg<5>::THREADS_SMALL : PaCfg<5>::THREADS;
    case 7: return small ? PaCfg<7>::THREADS_SMALL : PaCfg<7>::THREADS;
    case 9: return small ? PaCfg<9>::THREADS_SMALL : PaCfg<9>::THREADS;
    case 25: return G.pz == 1 ? (small ? PaCfg<25>::THREADS_SMALL : PaCfg<25>::THREADS) : 0;   // 2-d only
    }
    return 0;
}

template <typename T, int PX, int THREADS>
static hipError_t launch_pa(const T *pred, const float *S, const uint32_t *rows, const uint32_t *order,
                            const long long *group_start, const long long *chunk_offsets, int n_groups,
                            long long n_blocks, float *aff, const long long *drop_off,
                            const unsigned long long *drops, const Geo &G, size_t lds, hipStream_t s) {
    // (dynamic LDS above 64 KB -- the 9^3 rows -- is an opt-in per kernel)
    if (lds > 64 * 1024) {
        hipError_t ea = hipFuncSetAttribute((const void *)patch_graph_pa_kernel<T, PX, THREADS>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) return ea;
    }
    patch_graph_pa_kernel<T, PX, THREADS><<<dim3((unsigned)n_blocks), dim3(THREADS), lds, s>>>(
        pred, S, rows, order, group_start, chunk_offsets, n_groups, aff, drop_off, drops, G);
    return hipGetLastError();
}

hipError_t launch_patch_graph_pa(const void *pred, int dtype, const float *S, const uint32_t *rows,
                                 const uint32_t *order, const long long *group_start,
                                 const long long *chunk_offsets, int n_groups, long long n_blocks,
                                 int chunk, float *aff, const long long *drop_off,
                                 const unsigned long long *drops, const Geo &G, hipStream_t s) {
    if (n_groups <= 0 || n_blocks <= 0) return hipSuccess;
    const int threads = chunk;
    if (threads == 0 || (threads != patch_graph_pa_chunk(G, false) && threads != patch_graph_pa_chunk(G, true)))
        return hipErrorNotSupported;
    const bool small = threads != patch_graph_pa_chunk(G, false);
    const int words = (G.C + 31) / 32;
    const int W = (2 * G.pz - 1) * G.wy * G.wx;
    const int WB = (W + 2 * PA_PAD + 3) & ~3;
    // the per-thread staging registers are sized for a (2px-1)^3 row ((2px-1)^2 for the wide 2-d kernel)
    const int cube = (G.px > 16 ? 1 : 2 * G.px - 1) * (2 * G.px - 1) * (2 * G.px - 1);
    if (W > (cube + threads - 1) / threads * threads) return hipErrorNotSupported;
    // candidate planes are handled as ceil(px / (64 / px)) 64-bit chunks of 64 / px rows
    if (G.py > (G.px + 64 / G.px - 1) / (64 / G.px) * (64 / G.px) || G.pz > 32) return hipErrorNotSupported;
    const int row_bufs = G.px >= PPP_PA_SINGLE_BUF_FROM ? 1 : 2;
    const size_t lds = (size_t)(row_bufs * WB + ((words + 3) & ~3) + ((words * threads + 1) & ~1)) * 4;
    if (lds > 80 * 1024 || n_blocks >= (1ll << 31) || grid_too_big((unsigned long long)n_blocks, threads))
        return hipErrorNotSupported;
#define PPP_PA_CASE(P)                                                                                  \
    case P:                                                                                             \
        if (dtype == PPP_F16)                                                                           \
            return small ? launch_pa<__half, P, PaCfg<P>::THREADS_SMALL>((const __half *)pred, S, rows, order, group_start, chunk_offsets, n_groups, n_blocks, aff, drop_off, drops, G, lds, s) \
                         : launch_pa<__half, P, PaCfg<P>::THREADS>((const __half *)pred, S, rows, order, group_start, chunk_offsets, n_groups, n_blocks, aff, drop_off, drops, G, lds, s); \
        return small ? launch_pa<float, P, PaCfg<P>::THREADS_SMALL>((const float *)pred, S, rows, order, group_start, chunk_offsets, n_groups, n_blocks, aff, drop_off, drops, G, lds, s) \
                     : launch_pa<float, P, PaCfg<P>::THREADS>((const float *)pred, S, rows, order, group_start, chunk_offsets, n_groups, n_blocks, aff, drop_off, drops, G, lds, s);
    switch (G.px) {
        PPP_PA_CASE(3)
        PPP_PA_CASE(5)
        PPP_PA_CASE(7)
        PPP_PA_CASE(9)
        PPP_PA_CASE(25)
    default:
        return hipErrorNotSupported;
    }
#undef PPP_PA_CASE
}

}  // namespace ppp
