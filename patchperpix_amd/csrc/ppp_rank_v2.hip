// ppp_rank_v2.hip -- S2, second generation (px in {3,5,7,9}; other shapes use ppp_rank.hip).
// Same sequential float accumulation order as the reference (cuda/rankPatches.cu:28-147), hence
// bit-identical scores given the same consensus.
//
// One lane = one patch centre c, a wave = 64 consecutive centres along x.  What the reference
// re-derives inside its p^3 x p^3 loop from global memory is hoisted into three per-centre bit
// masks over the patch pixels r (kept in LDS, word-major so a wave reads them conflict-free):
//     V[r]  pixel z = c + r - rad is valid        (pred[mid][z] > TH  and, if OVERLAP, not overlap)
//     P[r]  V[r] and pred[r][c] > TH              ("foreground" pixel of the patch)
//     N[r]  V[r] and pred[r][c] < BG              ("background" pixel of the patch)
// For a in P (first pixel) the reference walks every b != a with V[b]:
//     b in P, b > a : acc += cons[b-a][z_a]                         (:88-101)
//     b in N        : acc -= cons[key], key = (b-a, z_a) if b > a else (a-b, z_b)   (:102-138)
//     fgCnt += 1 unless (b in P and b < a)                           (:139)
// so fgCnt = |P|*(|V|-1) - |P|*(|P|-1)/2 in closed form, and only the (a, b) combinations above
// touch memory.  Loops over a and (bz, by) are wave-uniform (scalar); the innermost PX pixels
// are unrolled: PX independent coalesced gathers (lane = consecutive base voxels) are issued
// together, then added in order.  Groups no lane needs are skipped with one ballot.
#include <type_traits>

#include "ppp_kernels.hpp"

namespace ppp {

static constexpr int R2_WAVES = 4;

template <typename T, int PX, bool COUNT_POS_NEG>
__global__ void __launch_bounds__(64 * R2_WAVES)
    rank_v2_kernel(const T *__restrict__ pred, const float *__restrict__ cons,
                   const uint8_t *__restrict__ ov, float *__restrict__ score, const ppp_box sb,
                   const Geo G, const int runs_per_line, const long long n_waves) {
    extern __shared__ uint32_t lds_raw[];  // [R2_WAVES][2][words][64]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long wid = (long long)blockIdx.x * (blockDim.x >> 6) + wave;
    if (wid >= n_waves) return;
    const int words = (G.C + 31) / 32;
    uint32_t *Pw = lds_raw + (size_t)(wave * 2 + 0) * words * 64 + lane;
    uint32_t *Nw = lds_raw + (size_t)(wave * 2 + 1) * words * 64 + lane;

    const int sX = sb.x1 - sb.x0, sY = sb.y1 - sb.y0;
    const int xr = (int)(wid % runs_per_line);
    const long long t0 = wid / runs_per_line;
    const int cy = sb.y0 + (int)(t0 % sY);
    const int cz = sb.z0 + (int)(t0 / sY);
    const int cx = sb.x0 + xr * 64 + lane;
    const bool in_box = cx < sb.x0 + sX;
    const long long lc = vox(G, cz, cy, min(cx, G.X - 1));
    const T *mid = pred + (long long)G.mid * G.V;
    const bool inter = in_box && interior(G, cz, cy, cx);
    const bool fg = inter && ldf(mid, lc) > G.th_gt;
    if (in_box && !inter) score[lc] = G.norm_rank ? -1.0f : -9999999.0f;
    if (in_box && inter && !fg) score[lc] = 0.0f;   // the reference leaves the allocation's zero
    if (__ballot(fg) == 0) return;

    // ---- per-centre bit masks
    unsigned nP = 0, nV = 0;
    {
        int r = 0;
        for (int w = 0; w < words; ++w) {
            uint32_t p = 0, n = 0, v = 0;
            for (int b = 0; b < 32 && r < G.C; ++b, ++r) {
                if (fg) {
                    const int z = cz + r / (G.py * PX) - G.rz;
                    const int y = cy + (r / PX) % G.py - G.ry;
                    const int x = cx + r % PX - PX / 2;
                    const long long lz = vox(G, z, y, x);
                    const bool valid = ldf(mid, lz) > G.th_gt && (!G.use_overlap || ov[lz] == 0);
                    const float val = ldf(pred, (long long)r * G.V + lc);
                    const uint32_t bit = 1u << b;
                    if (valid) v |= bit;
                    if (valid && val > G.th_gt) p |= bit;
                    if (valid && val < G.bg_lt) n |= bit;
                }
            }
            Pw[w * 64] = p; Nw[w * 64] = n;   // (V is only needed for |V|)
            nP += __popc(p); nV += __popc(v);
        }
    }
    // this lane's base index of c inside the consensus buffer, and the strides of a base voxel
    long long sYc, sZc, laneC, plane_stride;
    if (G.layout == PPP_CONS_REFERENCE) {
        sYc = G.X; sZc = (long long)G.X * G.Y; plane_stride = G.V;
        laneC = lc;
    } else {
        sYc = G.bX; sZc = (long long)G.bX * G.bY; plane_stride = G.BV;
        laneC = ((long long)(cz - G.bz0) * G.bY + (cy - G.by0)) * G.bX + (min(cx, G.X - 1) - G.bx0);
    }
    if (!fg) laneC = 0;   // inactive lanes read a harmless in-bounds element
    const long long step_y = (long long)(G.layout == PPP_CONS_REFERENCE ? G.nsx : G.wx) * plane_stride;
    const long long step_z = (long long)(G.layout == PPP_CONS_REFERENCE ? G.nsy * G.nsx : G.wy * G.wx) * plane_stride;

    float acc = 0.0f;
    for (int az = 0, a = 0; az < G.pz; ++az)
        for (int ay = 0; ay < G.py; ++ay)
            for (int ax = 0; ax < PX; ++ax, ++a) {
                const bool pa = fg && ((Pw[(a >> 5) * 64] >> (a & 31)) & 1u);
                if (__ballot(pa) == 0) continue;
                const long long off_a = (long long)(az - G.rz) * sZc + (long long)(ay - G.ry) * sYc + (ax - PX / 2);
                // Running base pointers of the partner row (bz, by), j = 0: forward keys
                // (b - a, z_a) and backward keys (a - b, z_b).  One scalar 64-bit add per row
                // step instead of re-deriving plane * stride (the loop is scalar-ALU bound).
                const float *pf_z, *pb_z;
                if (G.layout == PPP_CONS_REFERENCE) {
                    pf_z = cons + (((long long)(-az + G.pz - 1) * G.nsy + (-ay + G.py - 1)) * G.nsx + (-ax + PX - 1)) * plane_stride + off_a;
                    pb_z = cons + (((long long)(az + G.pz - 1) * G.nsy + (ay + G.py - 1)) * G.nsx + (ax + PX - 1)) * plane_stride +
                           ((long long)(-G.rz) * sZc + (long long)(-G.ry) * sYc - PX / 2);
                } else {
                    pf_z = cons + (((long long)(-az) * G.wy - ay) * G.wx - ax - 1) * plane_stride + off_a;
                    pb_z = cons + (((long long)az * G.wy + ay) * G.wx + ax - 1) * plane_stride +
                           ((long long)(-G.rz) * sZc + (long long)(-G.ry) * sYc - PX / 2);
                }
                // The partner rows (bz, by) are walked as one flattened sequence r = 0..R-1,
                // software-pipelined by one row: the gathers of row r+1 are in flight while
                // row r is accumulated (two row buffers, loop unrolled by two).  The kernel is
                // bound by memory-level parallelism (every gather misses L2), so doubling the
                // loads in flight per wave is what matters.  MODE (position of the partner
                // row relative to a's row: -1 before, 0 same, +1 after) stays a compile-time
                // constant of the gather / accumulate bodies (scalar-ALU pressure).
                const int R = G.pz * G.py, ra = az * G.py + ay;
                struct RowBuf { float v[PX]; uint32_t pf, nf; };
                // row cursor of the issue stream
                int ibz = 0, iby = 0;
                const float *pf_row_z = pf_z, *pb_row_z = pb_z, *pf_row = pf_z, *pb_row = pb_z;
                auto gather = [&](auto mode_tag, RowBuf &B, const float *pf_y, const float *pb_y) {
                    constexpr int MODE = decltype(mode_tag)::value;
                    const float *pf_ptr = pf_y, *pb_ptr = pb_y;
#pragma unroll
                    for (int j = 0; j < PX; ++j) {
                        const bool back = MODE < 0 || (MODE == 0 && j < ax);
                        const bool skip = MODE == 0 && j == ax;
                        const float *src = MODE < 0 ? pb_ptr : (MODE > 0 ? pf_ptr : (back ? pb_ptr : pf_ptr));
                        B.v[j] = (fg && !skip) ? src[laneC] : 0.0f;
                        if (MODE >= 0) pf_ptr += plane_stride;
                        if (MODE <= 0) pb_ptr += 1 - plane_stride;
                    }
                };
                // masks of the cursor row + its gathers; advances the cursor; false = no lane
                // needs the row (nothing issued)
                auto issue = [&](RowBuf &B) -> bool {
                    const int r = ibz * G.py + iby;
                    const int b0 = r * PX;
                    // PX-bit fields of P and N starting at bit b0 (may straddle two words)
                    const int w0 = b0 >> 5, sh = b0 & 31;
                    uint64_t p64 = Pw[w0 * 64], n64 = Nw[w0 * 64];
                    if (sh + PX > 32) {
                        p64 |= (uint64_t)Pw[(w0 + 1) * 64] << 32;
                        n64 |= (uint64_t)Nw[(w0 + 1) * 64] << 32;
                    }
                    uint32_t pf = (uint32_t)(p64 >> sh) & ((1u << PX) - 1u);
                    const uint32_t nf = (uint32_t)(n64 >> sh) & ((1u << PX) - 1u);
                    // pos votes only for b > a
                    if (r < ra) pf = 0;
                    else if (r == ra) pf &= ~((2u << ax) - 1u);   // keep bits j > ax
                    uint32_t need = pa ? (pf | nf) : 0u;
                    if (r == ra) need &= ~(1u << ax);             // b == a never votes
                    const bool any = __ballot(need != 0) != 0;
                    if (any) {
                        B.pf = pf; B.nf = nf;
                        if (r < ra) gather(std::integral_constant<int, -1>{}, B, pf_row, pb_row);
                        else if (r > ra) gather(std::integral_constant<int, 1>{}, B, pf_row, pb_row);
                        else gather(std::integral_constant<int, 0>{}, B, pf_row, pb_row);
                    }
                    // advance the cursor (running base pointers: one scalar 64-bit add per step)
                    if (++iby == G.py) {
                        iby = 0; ++ibz;
                        pf_row_z += step_z; pb_row_z += sZc - step_z;
                        pf_row = pf_row_z; pb_row = pb_row_z;
                    } else {
                        pf_row += step_y; pb_row += sYc - step_y;
                    }
                    return any;
                };
                // ordered accumulation of one buffered row r
                auto consume = [&](const RowBuf &B, const int r) {
                    auto body = [&](auto mode_tag) {
                        constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
                        for (int j = 0; j < PX; ++j) {
                            const bool is_p = (B.pf >> j) & 1u, is_n = (B.nf >> j) & 1u;
                            const bool use = pa && (is_p || is_n) && !(MODE == 0 && j == ax);
                            float c = B.v[j];
                            if (COUNT_POS_NEG) c = (c != 0.0f) ? copysignf(1.0f, c) : (is_p ? -1.0f : 1.0f);
                            // acc += c for a foreground partner, acc -= c for a background one
                            const float term = is_p ? c : -c;
                            acc = acc + (use ? term : 0.0f);
                        }
                    };
                    if (r == ra) body(std::integral_constant<int, 0>{});
                    else body(std::integral_constant<int, 1>{});   // (MODE only matters on a's row)
                };
                RowBuf bufA, bufB;
                bool anyA = issue(bufA), anyB = false;
                for (int r = 0; r < R; r += 2) {
                    if (r + 1 < R) anyB = issue(bufB);
                    if (anyA) consume(bufA, r);
                    if (r + 2 < R) anyA = issue(bufA);
                    if (r + 1 < R && anyB) consume(bufB, r + 1);
                }
            }
    if (fg) {
        const unsigned fg_cnt = nP * (nV - 1u) - nP * (nP - 1u) / 2u;
        score[lc] = G.norm_rank ? acc / (float)(fg_cnt > 1u ? fg_cnt : 1u) : acc;
    }
}

template <typename T, int PX>
static hipError_t launch_r2(const T *pred, const float *cons, const uint8_t *ov, float *score,
                            const ppp_box &sb, const Geo &G, hipStream_t s) {
    const int sX = sb.x1 - sb.x0, sY = sb.y1 - sb.y0, sZ = sb.z1 - sb.z0;
    const int runs_per_line = (sX + 63) / 64;
    const long long n_waves = (long long)runs_per_line * sY * sZ;
    const size_t per_wave = (size_t)2 * ((G.C + 31) / 32) * 64 * sizeof(uint32_t);
    int waves = R2_WAVES;
    while (waves > 1 && waves * per_wave > 40 * 1024) waves >>= 1;   // keep >= 4 blocks per CU
    if (waves * per_wave > 64 * 1024) return hipErrorNotSupported;
    const size_t lds = waves * per_wave;
    PPP_GRID_CHECK((n_waves + waves - 1) / waves, 64 * waves);
    const dim3 grid((unsigned)((n_waves + waves - 1) / waves)), block(64 * waves);
    if (G.count_pos_neg)
        rank_v2_kernel<T, PX, true><<<grid, block, lds, s>>>(pred, cons, ov, score, sb, G, runs_per_line, n_waves);
    else
        rank_v2_kernel<T, PX, false><<<grid, block, lds, s>>>(pred, cons, ov, score, sb, G, runs_per_line, n_waves);
    return hipGetLastError();
}

hipError_t launch_rank_v2(const void *pred, int dtype, const float *cons, const uint8_t *ov,
                          float *score, const ppp_box &sb, const Geo &G, hipStream_t s) {
#define PPP_R2_CASE(P)                                                                          \
    case P:                                                                                     \
        return dtype == PPP_F16 ? launch_r2<__half, P>((const __half *)pred, cons, ov, score, sb, G, s) \
                                : launch_r2<float, P>((const float *)pred, cons, ov, score, sb, G, s);
    switch (G.px) {
        PPP_R2_CASE(3)
        PPP_R2_CASE(5)
        PPP_R2_CASE(7)
        PPP_R2_CASE(9)
    default:
        return hipErrorNotSupported;
    }
#undef PPP_R2_CASE
}

}  // namespace ppp
