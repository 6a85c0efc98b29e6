// ppp_rank_v3.hip -- S2 with KZ patch centres stacked along z per lane (px in {3,5,7,9}).
//
// Same arithmetic and the same per-centre summation order as ppp_rank_v2.hip / the reference
// (cuda/rankPatches.cu:28-147): bit-identical scores.  What changes is how often a consensus
// entry is fetched.  The kernel is bound by the rate of its 256-byte gathers (3.6 TB/s of
// fabric traffic at 140^3 / 7^3), and two centres c and c + e_z need the SAME entry
//     cons[b - a][c + a - rad]   at their steps (a, b)  and  (a - e_z, b - e_z):
// same plane b - a, same base voxel.  A lane therefore carries KZ centres (cz .. cz+KZ-1, same
// y, x) through ONE loop nest over the unified indices AZ = az_k + k, BZ = bz_k + k:
// centre k is active while 0 <= AZ-k, BZ-k < pz; every gathered value is added to the
// accumulators of all centres whose own masks ask for it.  For each centre the loop still
// visits its (a, b) combinations in raster order, and the position of the partner row relative
// to a's row (before / same / after: which key orientation is read) is the same for all k.
// Rows: (pz+KZ-1)^2 / (KZ pz^2) of the per-centre gathers of the one-centre kernel (0.65 at
// KZ = 2, 0.55 at KZ = 3 for pz = 7).
//
// STATUS: bit-identical (parity-tested with PPP_RANK_KZ=2), but SLOWER than ppp_rank_v2.hip on
// MI355X -- 216 ms (KZ = 2) / 299 ms (KZ = 3) against 195 ms at 140^3 / 7^3: the per-centre bit
// masks double the LDS per wave and halve the occupancy, and this kernel lives on waves in
// flight.  Kept selectable (PPP_RANK_KZ) as the measured counter-example; not the default.
#include <type_traits>

#include "ppp_kernels.hpp"

namespace ppp {

static constexpr int R3_WAVES = 4;

template <typename T, int PX, bool COUNT_POS_NEG, int KZ>
__global__ void __launch_bounds__(64 * R3_WAVES)
    rank_v3_kernel(const T *__restrict__ pred, const float *__restrict__ cons,
                   const uint8_t *__restrict__ ov, float *__restrict__ score, const ppp_box sb,
                   const Geo G, const int runs_per_line, const long long n_waves) {
    extern __shared__ uint32_t lds_raw[];  // [waves][KZ][2][words][64]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long wid = (long long)blockIdx.x * (blockDim.x >> 6) + wave;
    if (wid >= n_waves) return;
    const int words = (G.C + 31) / 32;
    uint32_t *Pw[KZ], *Nw[KZ];
#pragma unroll
    for (int k = 0; k < KZ; ++k) {
        Pw[k] = lds_raw + (size_t)((wave * KZ + k) * 2 + 0) * words * 64 + lane;
        Nw[k] = lds_raw + (size_t)((wave * KZ + k) * 2 + 1) * words * 64 + lane;
    }
    const int sX = sb.x1 - sb.x0, sY = sb.y1 - sb.y0;
    const int xr = (int)(wid % runs_per_line);
    const long long t0 = wid / runs_per_line;
    const int cy = sb.y0 + (int)(t0 % sY);
    const int cz0 = sb.z0 + (int)(t0 / sY) * KZ;
    const int cx = sb.x0 + xr * 64 + lane;
    const bool in_box = cx < sb.x0 + sX;
    const T *mid = pred + (long long)G.mid * G.V;
    bool fg[KZ];
    long long lc[KZ];
    bool any_fg = false;
#pragma unroll
    for (int k = 0; k < KZ; ++k) {
        const int cz = cz0 + k;
        const bool exists = in_box && cz < sb.z1;
        lc[k] = vox(G, min(cz, G.Z - 1), cy, min(cx, G.X - 1));
        const bool inter = exists && interior(G, cz, cy, cx);
        fg[k] = inter && ldf(mid, lc[k]) > G.th_gt;
        if (exists && !inter) score[lc[k]] = G.norm_rank ? -1.0f : -9999999.0f;
        if (exists && inter && !fg[k]) score[lc[k]] = 0.0f;   // the reference leaves the allocation's zero
        any_fg = any_fg || fg[k];
    }
    if (__ballot(any_fg) == 0) return;

    // ---- per-centre bit masks (see ppp_rank_v2.hip)
    unsigned nP[KZ], nV[KZ];
#pragma unroll
    for (int k = 0; k < KZ; ++k) {
        const int cz = cz0 + k;
        nP[k] = nV[k] = 0;
        int r = 0;
        for (int w = 0; w < words; ++w) {
            uint32_t p = 0, n = 0, v = 0;
            for (int b = 0; b < 32 && r < G.C; ++b, ++r) {
                if (fg[k]) {
                    const int z = cz + r / (G.py * PX) - G.rz;
                    const int y = cy + (r / PX) % G.py - G.ry;
                    const int x = cx + r % PX - PX / 2;
                    const long long lz = vox(G, z, y, x);
                    const bool valid = ldf(mid, lz) > G.th_gt && (!G.use_overlap || ov[lz] == 0);
                    const float val = ldf(pred, (long long)r * G.V + lc[k]);
                    const uint32_t bit = 1u << b;
                    if (valid) v |= bit;
                    if (valid && val > G.th_gt) p |= bit;
                    if (valid && val < G.bg_lt) n |= bit;
                }
            }
            Pw[k][w * 64] = p; Nw[k][w * 64] = n;
            nP[k] += __popc(p); nV[k] += __popc(v);
        }
    }
    // base index of centre 0 inside the consensus buffer, and the strides of a base voxel
    long long sYc, sZc, laneC, plane_stride;
    if (G.layout == PPP_CONS_REFERENCE) {
        sYc = G.X; sZc = (long long)G.X * G.Y; plane_stride = G.V;
        laneC = vox(G, cz0, cy, min(cx, G.X - 1));
    } else {
        sYc = G.bX; sZc = (long long)G.bX * G.bY; plane_stride = G.BV;
        laneC = ((long long)(cz0 - G.bz0) * G.bY + (cy - G.by0)) * G.bX + (min(cx, G.X - 1) - G.bx0);
    }
    const long long step_y = (long long)(G.layout == PPP_CONS_REFERENCE ? G.nsx : G.wx) * plane_stride;
    const long long step_z = (long long)(G.layout == PPP_CONS_REFERENCE ? G.nsy * G.nsx : G.wy * G.wx) * plane_stride;
    constexpr uint32_t RM = (1u << PX) - 1u;
    // PX-bit field of a mask starting at bit b0 (may straddle two words)
    auto field = [&](const uint32_t *m, int b0) -> uint32_t {
        const int w0 = b0 >> 5, sh = b0 & 31;
        uint64_t v = m[w0 * 64];
        if (sh + PX > 32) v |= (uint64_t)m[(w0 + 1) * 64] << 32;
        return (uint32_t)(v >> sh) & RM;
    };

    float acc[KZ];
#pragma unroll
    for (int k = 0; k < KZ; ++k) acc[k] = 0.0f;
    const int NZ = G.pz + KZ - 1;                      // unified first-pixel / partner planes
    for (int AZ = 0; AZ < NZ; ++AZ)
        for (int ay = 0; ay < G.py; ++ay)
            for (int ax = 0; ax < PX; ++ax) {
                bool pa[KZ], pa_any = false;
#pragma unroll
                for (int k = 0; k < KZ; ++k) {
                    const int az = AZ - k;
                    const int a = (az * G.py + ay) * PX + ax;
                    pa[k] = fg[k] && az >= 0 && az < G.pz && ((Pw[k][(a >> 5) * 64] >> (a & 31)) & 1u);
                    pa_any = pa_any || pa[k];
                }
                if (__ballot(pa_any) == 0) continue;
                const long long off_a = (long long)(AZ - G.rz) * sZc + (long long)(ay - G.ry) * sYc + (ax - PX / 2);
                // running base pointers of the partner row (BZ, by), j = 0 (ppp_rank_v2.hip)
                const float *pf_z, *pb_z;
                if (G.layout == PPP_CONS_REFERENCE) {
                    pf_z = cons + (((long long)(-AZ + G.pz - 1) * G.nsy + (-ay + G.py - 1)) * G.nsx + (-ax + PX - 1)) * plane_stride + off_a;
                    pb_z = cons + (((long long)(AZ + G.pz - 1) * G.nsy + (ay + G.py - 1)) * G.nsx + (ax + PX - 1)) * plane_stride +
                           ((long long)(-G.rz) * sZc + (long long)(-G.ry) * sYc - PX / 2);
                } else {
                    pf_z = cons + (((long long)(-AZ) * G.wy - ay) * G.wx - ax - 1) * plane_stride + off_a;
                    pb_z = cons + (((long long)AZ * G.wy + ay) * G.wx + ax - 1) * plane_stride +
                           ((long long)(-G.rz) * sZc + (long long)(-G.ry) * sYc - PX / 2);
                }
                const int R = NZ * G.py, ra = AZ * G.py + ay;
                // up / un: bits j whose gathered value is ADDED to / SUBTRACTED from acc[k]
                struct RowBuf { float v[PX]; uint32_t up[KZ], un[KZ]; };
                int ibz = 0, iby = 0;
                const float *pf_row_z = pf_z, *pb_row_z = pb_z, *pf_row = pf_z, *pb_row = pb_z;
                auto gather = [&](auto mode_tag, RowBuf &B, uint32_t need, const float *pf_y, const float *pb_y) {
                    constexpr int MODE = decltype(mode_tag)::value;
                    const float *pf_ptr = pf_y, *pb_ptr = pb_y;
#pragma unroll
                    for (int j = 0; j < PX; ++j) {
                        const bool back = MODE < 0 || (MODE == 0 && j < ax);
                        const float *src = MODE < 0 ? pb_ptr : (MODE > 0 ? pf_ptr : (back ? pb_ptr : pf_ptr));
                        // a lane only touches what one of its own centres asked for (in bounds)
                        B.v[j] = ((need >> j) & 1u) ? src[laneC] : 0.0f;
                        if (MODE >= 0) pf_ptr += plane_stride;
                        if (MODE <= 0) pb_ptr += 1 - plane_stride;
                    }
                };
                auto issue = [&](RowBuf &B) -> bool {
                    const int r = ibz * G.py + iby;
                    uint32_t need = 0;
#pragma unroll
                    for (int k = 0; k < KZ; ++k) {
                        const int bz = ibz - k;
                        uint32_t pf = 0, nf = 0;
                        if (bz >= 0 && bz < G.pz) {      // (wave-uniform)
                            const int b0 = (bz * G.py + iby) * PX;
                            pf = field(Pw[k], b0);
                            nf = field(Nw[k], b0);
                        }
                        // pos votes only for b > a; b == a never votes
                        if (r < ra) pf = 0;
                        else if (r == ra) { pf &= ~((2u << ax) - 1u); nf &= ~(1u << ax); }
                        B.up[k] = pa[k] ? pf : 0u;
                        B.un[k] = pa[k] ? nf : 0u;
                        need |= B.up[k] | B.un[k];
                    }
                    const bool any = __ballot(need != 0) != 0;
                    if (any) {
                        if (r < ra) gather(std::integral_constant<int, -1>{}, B, need, pf_row, pb_row);
                        else if (r > ra) gather(std::integral_constant<int, 1>{}, B, need, pf_row, pb_row);
                        else gather(std::integral_constant<int, 0>{}, B, need, pf_row, pb_row);
                    }
                    if (++iby == G.py) {
                        iby = 0; ++ibz;
                        pf_row_z += step_z; pb_row_z += sZc - step_z;
                        pf_row = pf_row_z; pb_row = pb_row_z;
                    } else {
                        pf_row += step_y; pb_row += sYc - step_y;
                    }
                    return any;
                };
                auto consume = [&](const RowBuf &B) {
#pragma unroll
                    for (int j = 0; j < PX; ++j) {
#pragma unroll
                        for (int k = 0; k < KZ; ++k) {
                            const bool is_p = (B.up[k] >> j) & 1u, is_n = (B.un[k] >> j) & 1u;
                            float c = B.v[j];
                            if (COUNT_POS_NEG) c = (c != 0.0f) ? copysignf(1.0f, c) : (is_p ? -1.0f : 1.0f);
                            // acc += c for a foreground partner, acc -= c for a background one
                            const float term = is_p ? c : -c;
                            acc[k] = acc[k] + ((is_p || is_n) ? term : 0.0f);
                        }
                    }
                };
                RowBuf bufA, bufB;
                bool anyA = issue(bufA), anyB = false;
                for (int r = 0; r < R; r += 2) {
                    if (r + 1 < R) anyB = issue(bufB);
                    if (anyA) consume(bufA);
                    if (r + 2 < R) anyA = issue(bufA);
                    if (r + 1 < R && anyB) consume(bufB);
                }
            }
#pragma unroll
    for (int k = 0; k < KZ; ++k)
        if (fg[k]) {
            const unsigned fg_cnt = nP[k] * (nV[k] - 1u) - nP[k] * (nP[k] - 1u) / 2u;
            score[lc[k]] = G.norm_rank ? acc[k] / (float)(fg_cnt > 1u ? fg_cnt : 1u) : acc[k];
        }
}

template <typename T, int PX, int KZ>
static hipError_t launch_r3(const T *pred, const float *cons, const uint8_t *ov, float *score,
                            const ppp_box &sb, const Geo &G, hipStream_t s) {
    const int sX = sb.x1 - sb.x0, sY = sb.y1 - sb.y0, sZ = sb.z1 - sb.z0;
    const int runs_per_line = (sX + 63) / 64;
    const long long n_waves = (long long)runs_per_line * sY * ((sZ + KZ - 1) / KZ);
    const size_t per_wave = (size_t)KZ * 2 * ((G.C + 31) / 32) * 64 * sizeof(uint32_t);
    int waves = R3_WAVES;
    while (waves > 1 && waves * per_wave > 48 * 1024) waves >>= 1;
    if (waves * per_wave > 64 * 1024) return hipErrorNotSupported;
    const size_t lds = waves * per_wave;
    PPP_GRID_CHECK((n_waves + waves - 1) / waves, 64 * waves);
    const dim3 grid((unsigned)((n_waves + waves - 1) / waves)), block(64 * waves);
    if (G.count_pos_neg)
        rank_v3_kernel<T, PX, true, KZ><<<grid, block, lds, s>>>(pred, cons, ov, score, sb, G, runs_per_line, n_waves);
    else
        rank_v3_kernel<T, PX, false, KZ><<<grid, block, lds, s>>>(pred, cons, ov, score, sb, G, runs_per_line, n_waves);
    return hipGetLastError();
}

// kz centres per lane (2 or 3); hipErrorNotSupported when there is no such specialisation
hipError_t launch_rank_v3(const void *pred, int dtype, const float *cons, const uint8_t *ov,
                          float *score, const ppp_box &sb, const Geo &G, int kz, hipStream_t s) {
#define PPP_R3_CASE(P, K)                                                                              \
    if (G.px == P && kz == K)                                                                          \
        return dtype == PPP_F16 ? launch_r3<__half, P, K>((const __half *)pred, cons, ov, score, sb, G, s) \
                                : launch_r3<float, P, K>((const float *)pred, cons, ov, score, sb, G, s);
    PPP_R3_CASE(3, 2) PPP_R3_CASE(5, 2) PPP_R3_CASE(7, 2) PPP_R3_CASE(9, 2)
    PPP_R3_CASE(7, 3)
#undef PPP_R3_CASE
    return hipErrorNotSupported;
}

}  // namespace ppp
