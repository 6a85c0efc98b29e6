// ppp_patch_graph.hip -- S5: affinity of a pair of selected patches from the consensus.
//
// Reference: cuda/computePatchGraph.cu:3-136 -- one thread per patch pair (A, B) walks all
// p^3 x p^3 pixel pairs (z1 in A, z2 in B); pairs of foreground pixels that are close enough
// gather consensus[z2 - z1][earlier pixel]; pixel pairs inside the patch intersection are
// thinned to ~20 % by a per-pair LCG.  The float sum and the LCG are ORDER dependent, so each
// patch pair is evaluated by ONE lane in exactly the reference's loop order (bit-identical).
//
// MI355X mapping.  With one lane per pair, everything that depends only on the patch offset
// d = cB - cA (which (r1, r2) combinations are in range, which consensus plane and base
// offset they address, whether they lie in the intersection) is the same for all pairs with
// the same d.  The caller passes a processing ORDER that groups pairs by d; a wave then runs
// the loops with scalar (SGPR) control flow and scalar address arithmetic, and the per-lane
// work shrinks to: two bit tests, the predicated LCG step, one gather at
// `lane_base + scalar_offset`, one add.  The per-pixel foreground tests of the reference
// (pred[mid][z] > TH and pred[r][c] > TH, :44-52,60-66) are evaluated once per patch into bit
// masks held in LDS, instead of p^3 times inside the inner loop.
#include <cstdlib>

#include "ppp_kernels.hpp"

namespace ppp {

static constexpr int PG_WAVES = 4;     // waves per workgroup (independent of each other)

// bit r = (pred[mid][c + r - rad] > TH) && (pred[r][c] > TH)
template <typename T>
__device__ __forceinline__ void patch_fg_words(const T *__restrict__ pred, const Geo &G, int cz,
                                               int cy, int cx, uint32_t *lds_col /* [word*64] */,
                                               int words) {
    const T *mid = pred + (long long)G.mid * G.V;
    const long long lc = vox(G, cz, cy, cx);
    int r = 0;
    for (int w = 0; w < words; ++w) {
        uint32_t bits = 0;
        for (int b = 0; b < 32 && r < G.C; ++b, ++r) {
            const int z = cz + r / (G.py * G.px) - G.rz;
            const int y = cy + (r / G.px) % G.py - G.ry;
            const int x = cx + r % G.px - G.rx;
            const bool on = ldf(mid, vox(G, z, y, x)) > G.th_gt &&
                            ldf(pred, (long long)r * G.V + lc) > G.th_gt;
            bits |= (on ? 1u : 0u) << b;
        }
        lds_col[w * 64] = bits;
    }
}

template <typename T>
__global__ void __launch_bounds__(64 * PG_WAVES)
    patch_graph_kernel(const T *__restrict__ pred, const float *__restrict__ cons,
                       const uint32_t *__restrict__ pairs, const uint32_t *__restrict__ order,
                       const uint64_t n, float *__restrict__ aff, const Geo G) {
    extern __shared__ uint32_t lds_raw[];  // [PG_WAVES][2][words][64]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t slot = ((uint64_t)blockIdx.x * PG_WAVES + wave) * 64 + lane;
    const bool live = slot < n;
    const uint64_t id = live ? (order ? (uint64_t)order[slot] : slot) : 0;
    const int words = (G.C + 31) / 32;
    uint32_t *lds_a = lds_raw + (size_t)(wave * 2 + 0) * words * 64;
    uint32_t *lds_b = lds_raw + (size_t)(wave * 2 + 1) * words * 64;

    int az = 0, ay = 0, ax = 0, bz = 0, by = 0, bx = 0;
    if (live) {
        az = (int)pairs[id * 6 + 0]; ay = (int)pairs[id * 6 + 1]; ax = (int)pairs[id * 6 + 2];
        bz = (int)pairs[id * 6 + 3]; by = (int)pairs[id * 6 + 4]; bx = (int)pairs[id * 6 + 5];
        patch_fg_words(pred, G, az, ay, ax, lds_a + lane, words);
        patch_fg_words(pred, G, bz, by, bx, lds_b + lane, words);
    }
    // the LCG seed is the product of the GLOBAL coordinates (computePatchGraph.cu:24-27)
    uint32_t rnd = (uint32_t)(az + G.oz) * (uint32_t)(bz + G.oz) * (uint32_t)(ay + G.oy) *
                   (uint32_t)(by + G.oy) * (uint32_t)(ax + G.ox) * (uint32_t)(bx + G.ox);
    // consensus strides of the base voxel (tile or volume) and this lane's base index of cA
    long long sY, sZ, laneA;
    if (G.layout == PPP_CONS_REFERENCE) {
        sY = G.X; sZ = (long long)G.X * G.Y;
        laneA = vox(G, az, ay, ax);
    } else {
        sY = G.bX; sZ = (long long)G.bX * G.bY;
        laneA = ((long long)(az - G.bz0) * G.bY + (ay - G.by0)) * G.bX + (ax - G.bx0);
    }
    const long long plane_stride = G.layout == PPP_CONS_REFERENCE ? G.V : G.BV;
    const int dzl = bz - az, dyl = by - ay, dxl = bx - ax;
    float acc = 0.0f;
    unsigned fg_cnt = 0;

    // process the distinct patch offsets d present in this wave one after the other
    unsigned long long todo = __ballot(live);
    while (todo) {
        const int first = __ffsll((long long)todo) - 1;
        // readlane keeps d in SGPRs: all loop bounds / address math below stay scalar
        const int dz = __builtin_amdgcn_readlane(dzl, first), dy = __builtin_amdgcn_readlane(dyl, first),
                  dx = __builtin_amdgcn_readlane(dxl, first);
        const bool mine = live && dzl == dz && dyl == dy && dxl == dx;
        todo &= ~__ballot(mine);

        int r1 = 0;
        for (int z1o = 0; z1o < G.pz; ++z1o)
            for (int y1o = 0; y1o < G.py; ++y1o)
                for (int x1o = 0; x1o < G.px; ++x1o, ++r1) {
                    const bool bit_a = mine && ((lds_a[(r1 >> 5) * 64 + lane] >> (r1 & 31)) & 1u);
                    if (__ballot(bit_a) == 0) continue;
                    // z1 relative to cA / cB
                    const int e1z = z1o - G.rz, e1y = y1o - G.ry, e1x = x1o - G.rx;
                    const bool in_b = abs(e1x - dx) <= G.rx && abs(e1y - dy) <= G.ry &&
                                      abs(e1z - dz) <= G.rz;
                    // r2 candidates: |d + r2 - r1| <= p per axis
                    const int z_lo = max(0, z1o - dz - G.pz), z_hi = min(G.pz - 1, z1o - dz + G.pz);
                    const int y_lo = max(0, y1o - dy - G.py), y_hi = min(G.py - 1, y1o - dy + G.py);
                    const int x_lo = max(0, x1o - dx - G.px), x_hi = min(G.px - 1, x1o - dx + G.px);
                    for (int z2o = z_lo; z2o <= z_hi; ++z2o)
                        for (int y2o = y_lo; y2o <= y_hi; ++y2o) {
                            const int r2row = (z2o * G.py + y2o) * G.px;
                            for (int x2o = x_lo; x2o <= x_hi; ++x2o) {
                                const int r2 = r2row + x2o;
                                // z2 - z1
                                int qz = dz + z2o - z1o, qy = dy + y2o - y1o, qx = dx + x2o - x1o;
                                const bool fwd = qz > 0 || (qz == 0 && (qy > 0 || (qy == 0 && qx >= 0)));
                                long long off;  // base voxel relative to cA
                                if (fwd) {
                                    off = (long long)e1z * sZ + (long long)e1y * sY + e1x;
                                } else {
                                    qz = -qz; qy = -qy; qx = -qx;
                                    off = (long long)(dz + z2o - G.rz) * sZ +
                                          (long long)(dy + y2o - G.ry) * sY + (dx + x2o - G.rx);
                                }
                                const bool inter = in_b && abs(dx + x2o - G.rx) <= G.rx &&
                                                   abs(dy + y2o - G.ry) <= G.ry &&
                                                   abs(dz + z2o - G.rz) <= G.rz;
                                // reference bound 0 <= q + p - 1 < 2p (checked AFTER the LCG step)
                                const bool in_range = qz >= -(G.pz - 1) && qz <= G.pz &&
                                                      qy >= -(G.py - 1) && qy <= G.py &&
                                                      qx >= -(G.px - 1) && qx <= G.px;
                                if (!inter && !in_range) continue;
                                bool valid = bit_a && ((lds_b[(r2 >> 5) * 64 + lane] >> (r2 & 31)) & 1u);
                                if (inter) {
                                    const uint32_t nxt = rnd * 1103515245U;
                                    const float rnd_t = (float)nxt / 4294967296.0f;
                                    rnd = valid ? nxt : rnd;
                                    valid = valid && !((double)rnd_t > 0.2);
                                }
                                if (!in_range) continue;
                                // planes with a component == +p and the zero offset are never
                                // written by S1: they read as 0 but still count
                                const bool stored = qz < G.pz && qy < G.py && qx < G.px &&
                                                    (qz | qy | qx) != 0;
                                if (stored) {
                                    long long plane;
                                    if (G.layout == PPP_CONS_REFERENCE)
                                        plane = ((long long)(qz + G.pz - 1) * G.nsy + (qy + G.py - 1)) * G.nsx +
                                                (qx + G.px - 1);
                                    else
                                        plane = ((long long)qz * G.wy + qy) * G.wx + qx - 1;
                                    const float *src = cons + plane * plane_stride + off;
                                    float v = 0.0f;
                                    if (valid) v = src[laneA];
                                    acc += v;
                                }
                                fg_cnt += valid ? 1u : 0u;
                            }
                        }
                }
    }
    if (live) aff[id] = G.norm_aff ? acc / (float)(fg_cnt > 1u ? fg_cnt : 1u) : acc;
}

// ---- fast path: symmetric voxel-major consensus (see cons_voxel_major_kernel) -------------
// S[v][L(q)] = consensus between voxel v and v + q for every signed offset q.  For a fixed
// pixel z1 = cA + e1 the inner loops over r2 walk q = d + r2 - r1 in raster order, i.e. L
// ascending: every lane sweeps one contiguous row segment of S, so each cache line it pulls
// is used completely (the plane-major layout costs one 64-byte line per 4-byte gather).
template <typename T>
__global__ void __launch_bounds__(64 * PG_WAVES)
    patch_graph_vm_kernel(const T *__restrict__ pred, const float *__restrict__ S,
                          const uint32_t *__restrict__ pairs, const uint32_t *__restrict__ order,
                          const uint64_t n, float *__restrict__ aff, const Geo G) {
    extern __shared__ uint32_t lds_raw[];  // [PG_WAVES][2][words][64]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t slot = ((uint64_t)blockIdx.x * PG_WAVES + wave) * 64 + lane;
    const bool live = slot < n;
    const uint64_t id = live ? (order ? (uint64_t)order[slot] : slot) : 0;
    const int words = (G.C + 31) / 32;
    uint32_t *lds_a = lds_raw + (size_t)(wave * 2 + 0) * words * 64;
    uint32_t *lds_b = lds_raw + (size_t)(wave * 2 + 1) * words * 64;

    int az = 0, ay = 0, ax = 0, bz = 0, by = 0, bx = 0;
    if (live) {
        az = (int)pairs[id * 6 + 0]; ay = (int)pairs[id * 6 + 1]; ax = (int)pairs[id * 6 + 2];
        bz = (int)pairs[id * 6 + 3]; by = (int)pairs[id * 6 + 4]; bx = (int)pairs[id * 6 + 5];
        patch_fg_words(pred, G, az, ay, ax, lds_a + lane, words);
        patch_fg_words(pred, G, bz, by, bx, lds_b + lane, words);
    }
    // the LCG seed is the product of the GLOBAL coordinates (computePatchGraph.cu:24-27)
    uint32_t rnd = (uint32_t)(az + G.oz) * (uint32_t)(bz + G.oz) * (uint32_t)(ay + G.oy) *
                   (uint32_t)(by + G.oy) * (uint32_t)(ax + G.ox) * (uint32_t)(bx + G.ox);
    const long long sY = G.bX, sZ = (long long)G.bX * G.bY;
    const long long laneA = ((long long)(az - G.bz0) * G.bY + (ay - G.by0)) * G.bX + (ax - G.bx0);
    const int W = (2 * G.pz - 1) * G.wy * G.wx, Lc = (W - 1) / 2;
    const int dzl = bz - az, dyl = by - ay, dxl = bx - ax;
    float acc = 0.0f;
    unsigned fg_cnt = 0;

    unsigned long long todo = __ballot(live);
    while (todo) {
        const int first = __ffsll((long long)todo) - 1;
        const int dz = __builtin_amdgcn_readlane(dzl, first), dy = __builtin_amdgcn_readlane(dyl, first),
                  dx = __builtin_amdgcn_readlane(dxl, first);
        const bool mine = live && dzl == dz && dyl == dy && dxl == dx;
        todo &= ~__ballot(mine);

        int r1 = 0;
        for (int z1o = 0; z1o < G.pz; ++z1o)
            for (int y1o = 0; y1o < G.py; ++y1o)
                for (int x1o = 0; x1o < G.px; ++x1o, ++r1) {
                    const bool bit_a = mine && ((lds_a[(r1 >> 5) * 64 + lane] >> (r1 & 31)) & 1u);
                    if (__ballot(bit_a) == 0) continue;
                    const int e1z = z1o - G.rz, e1y = y1o - G.ry, e1x = x1o - G.rx;
                    const bool in_b = abs(e1x - dx) <= G.rx && abs(e1y - dy) <= G.ry &&
                                      abs(e1z - dz) <= G.rz;
                    // this lane's row of S for pixel z1
                    const float *row = S + (laneA + (long long)e1z * sZ + (long long)e1y * sY + e1x) * W + Lc;
                    const int z_lo = max(0, z1o - dz - G.pz), z_hi = min(G.pz - 1, z1o - dz + G.pz);
                    const int y_lo = max(0, y1o - dy - G.py), y_hi = min(G.py - 1, y1o - dy + G.py);
                    const int x_lo = max(0, x1o - dx - G.px), x_hi = min(G.px - 1, x1o - dx + G.px);
                    for (int z2o = z_lo; z2o <= z_hi; ++z2o) {
                        const int qz = dz + z2o - z1o;
                        const bool iz = in_b && abs(dz + z2o - G.rz) <= G.rz;
                        for (int y2o = y_lo; y2o <= y_hi; ++y2o) {
                            const int qy = dy + y2o - y1o;
                            const bool izy = iz && abs(dy + y2o - G.ry) <= G.ry;
                            const int r2row = (z2o * G.py + y2o) * G.px;
                            const int Lrow = (qz * G.wy + qy) * G.wx;
                            for (int x2o = x_lo; x2o <= x_hi; ++x2o) {
                                const int r2 = r2row + x2o;
                                const int qx = dx + x2o - x1o;
                                // z1 <= z2 (raster)  <=>  q >= 0 lexicographically
                                const bool fwd = qz > 0 || (qz == 0 && (qy > 0 || (qy == 0 && qx >= 0)));
                                // reference bound on the ORIENTED offset o = +-q: -(p-1) <= o <= p
                                const int lo_z = fwd ? -(G.pz - 1) : -G.pz, hi_z = fwd ? G.pz : G.pz - 1;
                                const int lo_y = fwd ? -(G.py - 1) : -G.py, hi_y = fwd ? G.py : G.py - 1;
                                const int lo_x = fwd ? -(G.px - 1) : -G.px, hi_x = fwd ? G.px : G.px - 1;
                                const bool in_range = qz >= lo_z && qz <= hi_z && qy >= lo_y &&
                                                      qy <= hi_y && qx >= lo_x && qx <= hi_x;
                                const bool inter = izy && abs(dx + x2o - G.rx) <= G.rx;
                                if (!inter && !in_range) continue;
                                bool valid = bit_a && ((lds_b[(r2 >> 5) * 64 + lane] >> (r2 & 31)) & 1u);
                                if (inter) {
                                    const uint32_t nxt = rnd * 1103515245U;
                                    const float rnd_t = (float)nxt / 4294967296.0f;
                                    rnd = valid ? nxt : rnd;
                                    valid = valid && !((double)rnd_t > 0.2);
                                }
                                if (!in_range) continue;
                                // |q_i| == p and q == 0 are never written by S1: 0, but counted
                                const bool stored = abs(qz) < G.pz && abs(qy) < G.py && abs(qx) < G.px &&
                                                    (qz | qy | qx) != 0;
                                if (stored) {
                                    float v = 0.0f;
                                    if (valid) v = row[Lrow + qx];
                                    acc += v;
                                }
                                fg_cnt += valid ? 1u : 0u;
                            }
                        }
                    }
                }
    }
    if (live) aff[id] = G.norm_aff ? acc / (float)(fg_cnt > 1u ? fg_cnt : 1u) : acc;
}

// ---- voxel-major fast path, px-specialised: the x-run of one (r1, z2o, y2o) row is fetched
// with (up to) two 16-byte loads per lane and consumed from registers -- 2 gather
// instructions instead of up to PX (each gather instruction costs the texture-address unit 64
// distinct cache lines, which is what bounds this kernel).
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

template <typename T, int PX>
__global__ void __launch_bounds__(64 * PG_WAVES)
    patch_graph_vm2_kernel(const T *__restrict__ pred, const float *__restrict__ S,
                           const uint32_t *__restrict__ pairs, const uint32_t *__restrict__ order,
                           const uint64_t n, float *__restrict__ aff, const Geo G) {
    extern __shared__ uint32_t lds_raw[];  // [PG_WAVES][2][words][64]
    constexpr int RX = PX / 2;
    constexpr int NV = ((PX + 3) / 4) * 4;   // floats of the per-row register window
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t slot = ((uint64_t)blockIdx.x * PG_WAVES + wave) * 64 + lane;
    const bool live = slot < n;
    const uint64_t id = live ? (order ? (uint64_t)order[slot] : slot) : 0;
    const int words = (G.C + 31) / 32;
    uint32_t *lds_a = lds_raw + (size_t)(wave * 2 + 0) * words * 64;
    uint32_t *lds_b = lds_raw + (size_t)(wave * 2 + 1) * words * 64;

    int az = 0, ay = 0, ax = 0, bz = 0, by = 0, bx = 0;
    if (live) {
        az = (int)pairs[id * 6 + 0]; ay = (int)pairs[id * 6 + 1]; ax = (int)pairs[id * 6 + 2];
        bz = (int)pairs[id * 6 + 3]; by = (int)pairs[id * 6 + 4]; bx = (int)pairs[id * 6 + 5];
        patch_fg_words(pred, G, az, ay, ax, lds_a + lane, words);
        patch_fg_words(pred, G, bz, by, bx, lds_b + lane, words);
    }
    // the LCG seed is the product of the GLOBAL coordinates (computePatchGraph.cu:24-27)
    uint32_t rnd = (uint32_t)(az + G.oz) * (uint32_t)(bz + G.oz) * (uint32_t)(ay + G.oy) *
                   (uint32_t)(by + G.oy) * (uint32_t)(ax + G.ox) * (uint32_t)(bx + G.ox);
    const long long sY = G.bX, sZ = (long long)G.bX * G.bY;
    const long long laneA = live ? ((long long)(az - G.bz0) * G.bY + (ay - G.by0)) * G.bX + (ax - G.bx0) : 0;
    const int W = (2 * G.pz - 1) * G.wy * G.wx, Lc = (W - 1) / 2;
    const long long n_elems = G.BV * (long long)W;
    const int dzl = bz - az, dyl = by - ay, dxl = bx - ax;
    float acc = 0.0f;
    unsigned fg_cnt = 0;

    unsigned long long todo = __ballot(live);
    while (todo) {
        const int first = __ffsll((long long)todo) - 1;
        const int dz = __builtin_amdgcn_readlane(dzl, first), dy = __builtin_amdgcn_readlane(dyl, first),
                  dx = __builtin_amdgcn_readlane(dxl, first);
        const bool mine = live && dzl == dz && dyl == dy && dxl == dx;
        todo &= ~__ballot(mine);

        int r1 = 0;
        for (int z1o = 0; z1o < G.pz; ++z1o)
            for (int y1o = 0; y1o < G.py; ++y1o)
                for (int x1o = 0; x1o < PX; ++x1o, ++r1) {
                    const bool bit_a = mine && ((lds_a[(r1 >> 5) * 64 + lane] >> (r1 & 31)) & 1u);
                    if (__ballot(bit_a) == 0) continue;
                    const int e1z = z1o - G.rz, e1y = y1o - G.ry, e1x = x1o - RX;
                    const bool in_b = abs(e1x - dx) <= RX && abs(e1y - dy) <= G.ry &&
                                      abs(e1z - dz) <= G.rz;
                    // index of this lane's S row for pixel z1 (clamped for idle lanes)
                    const long long rowi = (laneA + (long long)e1z * sZ + (long long)e1y * sY + e1x) * W + Lc;
                    const int z_lo = max(0, z1o - dz - G.pz), z_hi = min(G.pz - 1, z1o - dz + G.pz);
                    const int y_lo = max(0, y1o - dy - G.py), y_hi = min(G.py - 1, y1o - dy + G.py);
                    const int x_lo = max(0, x1o - dx - PX), x_hi = min(PX - 1, x1o - dx + PX);
                    if (x_lo > x_hi) continue;
                    const int n_x = x_hi - x_lo + 1;          // candidates on a row (<= PX)
                    const int qx0 = dx + x_lo - x1o;          // offset of candidate t = 0
                    // bit t set <=> lo <= qx0 + t <= hi  (t in [0, n_x))
                    auto tmask = [&](int lo_q, int hi_q) -> uint32_t {
                        const int a = max(lo_q - qx0, 0), b = min(hi_q - qx0, n_x - 1);
                        return a > b ? 0u : (((2u << b) - 1u) & ~((1u << a) - 1u));
                    };
                    // x parts (the same for every row of this r1)
                    const uint32_t mx_fwd = tmask(-(PX - 1), PX), mx_back = tmask(-PX, PX - 1);
                    const uint32_t mx_nonneg = tmask(0, 2 * PX), mx_stored = tmask(-(PX - 1), PX - 1);
                    // |dx + x2o - RX| <= RX with x2o = x_lo + t
                    uint32_t mx_inter = 0;
                    if (in_b) {
                        const int a = max(-dx - x_lo, 0), b = min(2 * RX - dx - x_lo, n_x - 1);
                        mx_inter = a > b ? 0u : (((2u << b) - 1u) & ~((1u << a) - 1u));
                    }
                    for (int z2o = z_lo; z2o <= z_hi; ++z2o) {
                        const int qz = dz + z2o - z1o;
                        const bool iz = abs(dz + z2o - G.rz) <= G.rz;
                        for (int y2o = y_lo; y2o <= y_hi; ++y2o) {
                            const int qy = dy + y2o - y1o;
                            const bool izy = iz && abs(dy + y2o - G.ry) <= G.ry;
                            const int r2row = (z2o * G.py + y2o) * PX;
                            // orientation of the pixel pair: z1 before z2 (raster) <=> q >= 0
                            const int sgn = qz != 0 ? qz : qy;      // sign of (qz, qy)
                            const bool zy_f = qz >= -(G.pz - 1) && qz <= G.pz && qy >= -(G.py - 1) && qy <= G.py;
                            const bool zy_b = qz >= -G.pz && qz <= G.pz - 1 && qy >= -G.py && qy <= G.py - 1;
                            const bool row_stored = abs(qz) < G.pz && abs(qy) < G.py;
                            // per-candidate masks of this row (all scalar)
                            const uint32_t m_fwd = sgn > 0 ? ~0u : (sgn < 0 ? 0u : mx_nonneg);
                            const uint32_t m_range = (zy_f ? (mx_fwd & m_fwd) : 0u) | (zy_b ? (mx_back & ~m_fwd) : 0u);
                            const uint32_t m_inter = izy ? mx_inter : 0u;
                            uint32_t m_stored = row_stored ? (mx_stored & m_range) : 0u;
                            if (sgn == 0 && qx0 <= 0 && -qx0 < n_x) m_stored &= ~(1u << (-qx0));   // q == 0
                            if ((m_range | m_inter) == 0u) continue;
                            // foreground bits of patch B for the candidates of this row
                            const int b0 = r2row + x_lo, w0 = b0 >> 5, sh = b0 & 31;
                            uint64_t f64 = lds_b[w0 * 64 + lane];
                            if (sh + n_x > 32) f64 |= (uint64_t)lds_b[(w0 + 1) * 64 + lane] << 32;
                            const uint32_t fb = bit_a ? (uint32_t)(f64 >> sh) : 0u;
                            // consensus values: window [qx0, qx0 + NV) of this lane's row, fetched
                            // with as many 16-byte loads as the run needs (uniform count)
                            float v[NV];
#pragma unroll
                            for (int k = 0; k < NV; ++k) v[k] = 0.0f;
                            if (m_stored != 0u) {
                                const int n_ld = (n_x + 3) >> 2;          // 1 .. NV/4
                                long long i0 = rowi + (long long)(qz * G.wy + qy) * G.wx + qx0;
                                const long long want = i0;
                                i0 = max(0ll, min(i0, n_elems - 4 * n_ld));
                                if (fb != 0u) {
#pragma unroll
                                    for (int q4 = 0; q4 < NV / 4; ++q4)
                                        if (q4 < n_ld) {
                                            const f4u t4 = *reinterpret_cast<const f4u *>(S + i0 + 4 * q4);
                                            v[4 * q4 + 0] = t4.x; v[4 * q4 + 1] = t4.y;
                                            v[4 * q4 + 2] = t4.z; v[4 * q4 + 3] = t4.w;
                                        }
                                }
                                // the clamp only moves the window at the two ends of the buffer
                                const int shift = (int)(want - i0);
                                if (__builtin_expect(__ballot(shift != 0 && fb != 0u) != 0ull, 0)) {
                                    float tt[NV];
#pragma unroll
                                    for (int k = 0; k < NV; ++k) {
                                        float val = 0.f;
#pragma unroll
                                        for (int m = 0; m < NV; ++m) val = (m == k + shift) ? v[m] : val;
                                        tt[k] = val;
                                    }
#pragma unroll
                                    for (int k = 0; k < NV; ++k) v[k] = shift != 0 ? tt[k] : v[k];
                                }
                            }
#pragma unroll
                            for (int t = 0; t < PX; ++t) {
                                if (t >= n_x) break;
                                const bool inter = (m_inter >> t) & 1u, in_range = (m_range >> t) & 1u;
                                if (!inter && !in_range) continue;
                                bool valid = (fb >> t) & 1u;
                                if (inter) {
                                    const uint32_t nxt = rnd * 1103515245U;
                                    const float rnd_t = (float)nxt / 4294967296.0f;
                                    rnd = valid ? nxt : rnd;
                                    // rnd_t > 0.2 (double)  <=>  rnd_t > largest float <= 0.2
                                    valid = valid && !(rnd_t > 0.19999998807907104f);
                                }
                                if (!in_range) continue;
                                if ((m_stored >> t) & 1u) acc += valid ? v[t] : 0.0f;
                                fg_cnt += valid ? 1u : 0u;
                            }
                        }
                    }
                }
    }
    if (live) aff[id] = G.norm_aff ? acc / (float)(fg_cnt > 1u ? fg_cnt : 1u) : acc;
}

hipError_t launch_patch_graph(const void *pred, int dtype, const float *cons,
                              const uint32_t *pairs, const uint32_t *order, uint64_t n,
                              float *aff, const Geo &G, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const size_t lds_bytes = (size_t)PG_WAVES * 2 * ((G.C + 31) / 32) * 64 * sizeof(uint32_t);
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    const uint64_t per_block = 64ull * PG_WAVES;
    PPP_GRID_CHECK((n + per_block - 1) / per_block, 64 * PG_WAVES);
    const dim3 grid((unsigned)((n + per_block - 1) / per_block));
    static EnvSwitch generic_sw("PPP_PATCH_GRAPH_GENERIC");
    const bool vm_generic = generic_sw.get() != nullptr;
    if (G.layout == PPP_CONS_VOXEL_MAJOR && !vm_generic &&
        (G.px == 3 || G.px == 5 || G.px == 7 || G.px == 9)) {
#define PPP_PG_CASE(P)                                                                                          \
    case P:                                                                                                     \
        if (dtype == PPP_F16)                                                                                   \
            patch_graph_vm2_kernel<__half, P><<<grid, dim3(64 * PG_WAVES), lds_bytes, s>>>(                     \
                (const __half *)pred, cons, pairs, order, n, aff, G);                                           \
        else                                                                                                    \
            patch_graph_vm2_kernel<float, P><<<grid, dim3(64 * PG_WAVES), lds_bytes, s>>>(                      \
                (const float *)pred, cons, pairs, order, n, aff, G);                                            \
        return hipGetLastError();
        switch (G.px) {
            PPP_PG_CASE(3)
            PPP_PG_CASE(5)
            PPP_PG_CASE(7)
            PPP_PG_CASE(9)
        }
#undef PPP_PG_CASE
    }
    if (G.layout == PPP_CONS_VOXEL_MAJOR) {
        if (dtype == PPP_F16)
            patch_graph_vm_kernel<__half><<<grid, dim3(64 * PG_WAVES), lds_bytes, s>>>((const __half *)pred, cons, pairs, order, n, aff, G);
        else
            patch_graph_vm_kernel<float><<<grid, dim3(64 * PG_WAVES), lds_bytes, s>>>((const float *)pred, cons, pairs, order, n, aff, G);
        return hipGetLastError();
    }
    if (dtype == PPP_F16)
        patch_graph_kernel<__half><<<grid, dim3(64 * PG_WAVES), lds_bytes, s>>>((const __half *)pred, cons, pairs, order, n, aff, G);
    else
        patch_graph_kernel<float><<<grid, dim3(64 * PG_WAVES), lds_bytes, s>>>((const float *)pred, cons, pairs, order, n, aff, G);
    return hipGetLastError();
}

}  // namespace ppp
