// ppp_consensus_v2.hip -- S1, second generation (px in {3,5,7,9}; other shapes use the generic
// kernel in ppp_consensus.hip).  Same arithmetic, same summation order, bit-identical output.
//
//   cons[d][u] = sum over centres c (raster order) of vote(pred[k][c], pred[k+d][c]),
//   c = u - k + rad, k = patch offset of u, k + d = patch offset of w = u + d.
//
// Work decomposition
//   wave  = 64 consecutive base voxels u along x  x  one offset row (dz, dy);
//   lane  = one u, holding the 2*PX-1 accumulators (+ integer counts) of all dx in registers;
//   loops = kz, ky (descending, scalar), kx and the partner column j = kx + dx fully unrolled.
//   Summing k in descending order == raster order of the centre, per key.
//
// Operands.  For one (kz, ky) the wave needs two rows of PX channels, (kz, ky, *) "about u" and
// (kz+dz, ky+dy, *) "about w", at the 64+PX-1 centres of its x-run.  They are CLASSIFIED while
// being staged into a wave-private LDS image:
//       t = +v        if v > TH        (and the pixel it talks about is valid foreground)
//       t = -(1 - v)  if v < BG        (same condition)
//       t = 0         otherwise
// ("about u" entries also fold in the centre's own foreground / interior test).  A pair then
// votes iff x = ta*tb != 0 and not both negative; |x| is exactly the float product the
// reference forms (v1*v2 or v1*(1-v2)), its sign is the sign of the vote.
//
// Normalised votes.  The reference evaluates (float)(((double)x - TH*TH) / (1.0 - TH*TH)).  The
// kernel computes (|x| - TH^2) in double exactly like that, multiplies by the rounded reciprocal
// instead of dividing (<= 2.5 double ulp away from the correctly rounded quotient) and rounds to
// float.  The float result can differ from the reference only if a float rounding boundary lies
// within those few double ulps, which is visible in the low 29 mantissa bits of the product; in
// that (~2^-26 probability) case the lane redoes the vote with the true double division.  The
// result is therefore bit-identical, at one double multiply instead of a double division.
#include <stdlib.h>

#include "ppp_kernels.hpp"

// (the 25-wide kernel's vote loops exceed the full-unroll size limit in its rarely used rule
// variants; partial unrolling is fine there -- and measurably better than forcing it: 88 vs 104 ms)
#pragma clang diagnostic ignored "-Wpass-failed"

namespace ppp {

static constexpr int V2_WAVES = 4;
#ifndef PPP_S1_PREFETCH
#define PPP_S1_PREFETCH(PX) ((PX) >= 9)
#endif
#ifndef PPP_S1_MINWAVES
#define PPP_S1_MINWAVES(PX) ((PX) <= 7 ? 3 : 2)   // waves per SIMD the register budget must allow
#endif

// FLAT: the 64 base voxels of a wave are consecutive in the FLATTENED (y, x) order of a z-slice of
// the consensus box, so a run may continue on the next line (two segments A / B, each with its
// own PX-1 halo of centres).  Without it the last run of every line carries idle lanes: 140
// columns = 64 + 64 + 12, 27 % of all lanes at the 140^3 benchmark volume.
template <int PX, bool FLAT>
struct V2 {
    static constexpr int RX = PX / 2;
    static constexpr int NC = 64 + (FLAT ? 2 : 1) * (PX - 1);   // centres per run (both segments)
    static constexpr int NT = 64 + (FLAT ? 4 : 2) * (PX - 1);   // target pixels per run
    static constexpr int NACC = 2 * PX - 1;
    static constexpr int IMG = 2 * PX * NC;      // floats of LDS image per wave
};

// one vote; VAL, EXACT compile-time so that the unrolled tile is straight-line code.
// amb_min tracks (VALU only, no lane masks) how close any fast-path double result came to a
// float rounding midpoint.
template <int VAL, bool EXACT, bool TH05>
__device__ __forceinline__ void vote(const double th2, const double den, const double inv_den,
                                     float ta, int lim, float tb, float &acc, unsigned &cnt,
                                     unsigned &amb_min) {
    // a vote needs two classified operands that are not both "background" (negative): the
    // caller passes lim = 0 when ta is negative (a negative tb then becomes +0 in the integer
    // max) and INT_MIN otherwise, so that validity is just "product != 0"
    const float x = ta * __int_as_float(max(__float_as_int(tb), lim));
    const bool valid = x != 0.0f;
    float y;
    if constexpr (VAL == PPP_VAL_NORM_PROB_PRODUCT && TH05 && !EXACT) {
        // TH = 0.5: (float)(((double)|x| - 0.25) / 0.75) in float arithmetic only.  |x| >= 0.25
        // for every classified pair, so d = |x| - 0.25 is exact in float; q0 = d * fl(4/3) is
        // within an ulp of the quotient, r = d - 0.75 * q0 is exact (fma) and fma(r, fl(4/3), q0)
        // is the correctly rounded d / 0.75.  The double division of the reference never lands
        // within half a double ulp of a float midpoint (4d/3 has the bit pattern 0101.. / 1010..
        // past the quotient), so its double rounding is harmless.  Checked EXHAUSTIVELY for all
        // 2.0e8 floats x in [0.25, 2^22] (tests/test_abi_and_host.py::test_th05_quotient).  All
        // operations are odd-symmetric, so the sign of x rides along.
        const float d = x - copysignf(0.25f, x);
        const float c43 = 0x1.555556p+0f;
        const float q0 = d * c43;
        const float r = __fmaf_rn(-0.75f, q0, d);
        y = __fmaf_rn(r, c43, q0);
    } else if constexpr (VAL == PPP_VAL_NORM_PROB_PRODUCT) {
        // sign(x) * fl32(fl64(fl64(|x| - TH^2) / den)); rounding is sign-symmetric, so the
        // sign is carried through the double arithmetic
        const double xd = (double)x;
        const double r = xd - __builtin_copysign(th2, xd);
        if constexpr (EXACT) {
            y = (float)(r / den);
        } else {
            const double yd = r * inv_den;
            y = (float)yd;
            // distance of the 29 discarded mantissa bits from the midpoint pattern (window
            // [mid-4, mid+4] maps to [0, 8]); invalid lanes may raise a harmless false alarm
            const unsigned lo = (unsigned)__double2loint(yd) & 0x1FFFFFFFu;
            amb_min = min(amb_min, lo - (0x10000000u - 4u));
        }
    } else if constexpr (VAL == PPP_VAL_PROB_PRODUCT) {
        y = x;
    } else {
        y = copysignf(1.0f, x);
    }
    acc = acc + (valid ? y : 0.0f);
    cnt += valid ? 1u : 0u;
}

// all votes of one (kz, ky): kx descending (raster order of the centre), every partner column.
// ROW0: offset row (dz, dy) == (0, 0), where only dx > 0 exists.  Returns "some fast-path
// result was ambiguous" (never when EXACT).
template <int PX, int NC, int VAL, bool ROW0, bool EXACT, bool TH05>
__device__ __forceinline__ bool tile_votes(const float *ia, const float *ib, const bool u_ok,
                                           const double th2, const double den,
                                           const double inv_den, float (&acc)[2 * PX - 1],
                                           unsigned (&cnt)[2 * PX - 1]) {
    unsigned amb_min = 0xFFFFFFFFu;
#pragma unroll
    for (int kx = PX - 1; kx >= 0; --kx) {
        const float ta = u_ok ? ia[kx * NC - kx] : 0.0f;
        int lim = ta < 0.0f ? 0 : (int)0x80000000;
        asm volatile("" : "+v"(lim));   // keep it ONE v_max_i32 per vote (no select per vote)
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            if (ROW0 && j <= kx) continue;   // offsets must be lexicographically positive
            vote<VAL, EXACT, TH05>(th2, den, inv_den, ta, lim, ib[j * NC - kx], acc[j - kx + PX - 1],
                             cnt[j - kx + PX - 1], amb_min);
        }
#ifndef PPP_V2_NOSCHED
        // keep the scheduler from interleaving all PX*PX votes (lane-mask register pressure)
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
    return amb_min <= 8u;
}

// element at a 32-bit byte offset from a (wave-uniform) base
template <typename T>
__device__ __forceinline__ float ldf_at(const T *base, unsigned byte_off) {
    return ldf(reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off), 0);
}

template <typename T, int PX, int VAL, bool TH05, bool FLAT>
__global__ void __launch_bounds__(64 * V2_WAVES, PPP_S1_MINWAVES(PX))
    consensus_v2_kernel(const T *__restrict__ pred, const uint8_t *__restrict__ ov,
                        float *__restrict__ cons, float *__restrict__ cnt_out, const Geo G,
                        const int n_rows, const int runs_per_line, const long long n_waves) {
    using K = V2<PX, FLAT>;
    constexpr int NE = (K::IMG + 63) / 64;           // staged elements per lane and tile
    constexpr int NEA = (PX * K::NC + 63) / 64;      // ... of which may belong to the "about u" rows
    __shared__ float lds[V2_WAVES][K::IMG];
    __shared__ uint8_t lds_valid[V2_WAVES][2][K::NT + 2];
    // (wave-uniform values are made scalar explicitly: the compiler cannot see that
    // threadIdx.x >> 6 is uniform and would keep the whole tile address arithmetic in VGPRs)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // XCD-aware order: blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous
    // range of (x-run, row) work so that the rows of one x-run meet in one L2
    long long bid = blockIdx.x;
    {
        const long long nb = gridDim.x, per = nb / 8, main = per * 8;
        if (bid < main) bid = (bid % 8) * per + bid / 8;
    }
    const long long wid = bid * V2_WAVES + wave;
    if (wid >= n_waves) return;
    const int row = (int)(wid % n_rows);
    long long run = wid / n_rows;
    // offset row (dz, dy): row 0..py-1 -> dz = 0, dy = 0..py-1; then dz >= 1, dy = -(py-1)..py-1
    int dz, dy;
    if (row < G.py) { dz = 0; dy = row; }
    else { const int t = row - G.py; dz = 1 + t / G.wy; dy = t % G.wy - (G.py - 1); }
    // run -> base voxels (box coordinates -> global).  `runs_per_line` = runs per line, or per
    // z-slice when FLAT.  Segment A: nA lanes from (uy, ux0) on; segment B (FLAT): the other
    // lanes from the start of the next line.
    const int xr = (int)(run % runs_per_line);
    run /= runs_per_line;
    int uy, uz, ux0, nA;
    if (FLAT) {
        const int flat0 = xr * 64;
        uy = G.by0 + flat0 / G.bX;
        uz = G.bz0 + (int)run;
        ux0 = G.bx0 + flat0 % G.bX;
        nA = min(64, G.bX - flat0 % G.bX);
    } else {
        uy = G.by0 + (int)(run % G.bY);
        uz = G.bz0 + (int)(run / G.bY);
        ux0 = G.bx0 + xr * 64;
        nA = 64;
    }
    const bool in_b = FLAT && lane >= nA;                          // this lane sits on line B
    const bool have_b = FLAT && nA < 64 && uy + 1 < G.by0 + G.bY;  // (wave-uniform)
    const int ux = in_b ? G.bx0 + (lane - nA) : ux0 + lane;
    const int uy_l = in_b ? uy + 1 : uy;
    const bool lane_ok = in_b ? have_b : ux < G.bx0 + G.bX;
    // image column of this lane's own centre (kx = PX-1); segment B starts after A's halo
    const int pos_l = lane + (PX - 1) + (in_b ? PX - 1 : 0);
    const int wz = uz + dz, wy = uy + dy;                          // (line A; line B: wy + 1)
    const bool w_ok_a = wz < G.Z && wy >= 0 && wy < G.Y;
    const bool w_ok_b = have_b && wz < G.Z && wy + 1 >= 0 && wy + 1 < G.Y;
    const bool w_row_ok = w_ok_a || w_ok_b;
    const bool row0 = dz == 0 && dy == 0;

    float acc[K::NACC];
    unsigned cnt[K::NACC];
#pragma unroll
    for (int i = 0; i < K::NACC; ++i) { acc[i] = 0.0f; cnt[i] = 0u; }

    const T *mid = pred + (long long)G.mid * G.V;
    float *img = lds[wave];
    uint8_t *uval = lds_valid[wave][0], *wval = lds_valid[wave][1];
    // validity (foreground && !overlap) of the target pixels on the u row and on the w row:
    // per segment x in [first - (PX-1), first + n - 1 + (PX-1)]; segment B follows A in the arrays
    const int ntA = nA + 2 * (PX - 1);
    for (int i = lane; i < K::NT; i += 64) {
        const bool sb = FLAT && i >= ntA;
        const int x = sb ? G.bx0 - (PX - 1) + (i - ntA) : ux0 - (PX - 1) + i;
        const int yy = sb ? uy + 1 : uy, wyy = sb ? wy + 1 : wy;
        bool vu = false, vw = false;
        if (x >= 0 && x < G.X && (!sb || have_b)) {
            const long long lu = vox(G, uz, yy, x);
            vu = ldf(mid, lu) > G.th_gt && (!G.use_overlap || ov[lu] == 0);
            if (sb ? w_ok_b : w_ok_a) {
                const long long lw = vox(G, wz, wyy, x);
                vw = ldf(mid, lw) > G.th_gt && (!G.use_overlap || ov[lw] == 0);
            }
        }
        uval[i] = vu; wval[i] = vw;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool u_ok = lane_ok && uval[(in_b ? ntA + (lane - nA) : lane) + PX - 1];
    const double inv_den = 1.0 / G.den;

    // Per-lane description of the NE image elements this lane stages for every tile
    // (element e = it*64 + lane  ->  row half, column j, centre i): constant over the tiles.
    //   ok_bits  bit it : the pixel the value talks about is valid and the centre is inside
    //                     the x-interior (centre row / foreground are tile dependent, tested later)
    //   seg_bits bit it : the centre belongs to segment B (next line)
    //   el_off[it]      : j * V + (line) * X + clamped centre x  (added to the tile's row base)
    // (32-bit BYTE offsets from a scalar row base: one VGPR per element and the
    // saddr + voffset addressing form; the launcher checks that they fit)
    unsigned ok_bits = 0, seg_bits = 0;
    unsigned el_off[NE];
    unsigned el_cx[NE];
    const int ncA = nA + (PX - 1);                    // centres of segment A
#pragma unroll
    for (int it = 0; it < NE; ++it) {
        const int e = it * 64 + lane;
        const int i = e % K::NC;
        const int j = (e / K::NC) % PX;
        const bool is_b = e >= PX * K::NC;
        const bool sb = FLAT && i >= ncA;
        const int iseg = sb ? i - ncA : i;
        const int cx = (sb ? G.bx0 : ux0) - (PX - 1) + K::RX + iseg;
        bool ok = e < K::IMG && cx >= K::RX && cx < G.X - K::RX && (!sb || have_b);
        if (ok) {
            const int ti = (sb ? ntA : 0) + iseg + j;        // target pixel = centre + j - RX
            ok = is_b ? wval[ti] : uval[ti];
        }
        ok_bits |= (ok ? 1u : 0u) << it;
        seg_bits |= (sb ? 1u : 0u) << it;
        const int cxc = min(max(cx, 0), G.X - 1) + (sb ? G.X : 0);
        el_cx[it] = (unsigned)cxc * (unsigned)sizeof(T);
        el_off[it] = (unsigned)(((long long)j * G.V + cxc) * (long long)sizeof(T));
    }

    if (w_row_ok) {
        const int kz_hi = min(G.pz - 1, G.pz - 1 - dz), kz_lo = max(0, -dz);
        const int ky_hi = min(G.py - 1, G.py - 1 - dy), ky_lo = max(0, -dy);
        // tiles (kz, ky) in descending order, skipping centre rows outside the interior
        int kz = kz_hi, ky = ky_hi + 1;
        bool row_a_ok = true, row_b_ok = false;   // centre row of the tile inside the interior, per segment
        auto next_tile = [&](int &z, int &y) -> bool {
            while (true) {
                if (--y < ky_lo) { y = ky_hi; --z; }
                if (z < kz_lo) return false;
                const int cz = uz - z + G.rz, cy = uy - y + G.ry;
                if (!(cz >= G.rz && cz < G.Z - G.rz)) continue;
                row_a_ok = cy >= G.ry && cy < G.Y - G.ry;
                row_b_ok = have_b && cy + 1 >= G.ry && cy + 1 < G.Y - G.ry;
                if (row_a_ok || row_b_ok) return true;
            }
        };
        float raw[NE], cmid[NEA];
        // issue the (independent, unconditional) loads of one tile into registers
        auto load_tile = [&](int z, int y) {
            // (FLAT: a tile is kept when the centre row of EITHER line is interior; the other
            // line's elements are masked but still loaded -- harmless: with ry >= 1, "row cy + 1
            // interior" implies cy >= 0 and "row cy interior" implies cy + 1 <= Y - 1, so both rows
            // are inside the slice)
            const long long crow = vox(G, uz - z + G.rz, uy - y + G.ry, 0);
            const long long ra0 = (long long)((z * G.py + y) * PX) * G.V + crow;
            const long long rb0 = (long long)(((z + dz) * G.py + (y + dy)) * PX) * G.V + crow;
#pragma unroll
            for (int it = 0; it < NE; ++it) {
                const bool is_b = it * 64 + lane >= PX * K::NC;
                raw[it] = ldf_at(pred + (is_b ? rb0 : ra0), el_off[it]);
                if (it < NEA) cmid[it] = ldf_at(mid + crow, el_cx[it]);
            }
        };
        // Register prefetch of the next tile (24-32 VGPRs) only where the kernel runs at two
        // waves per SIMD anyway (9^3).  At 7^3 dropping it frees the registers for a third
        // wave WITHOUT spilling: 124 -> 102 ms (the spilled variant moved 22 GB of scratch
        // per launch); the load latency is covered by the other waves.
        constexpr bool PREFETCH = PPP_S1_PREFETCH(PX);
        bool have = next_tile(kz, ky);
        if (PREFETCH && have) load_tile(kz, ky);
        while (have) {
            if (!PREFETCH) load_tile(kz, ky);
            // ---- classify the loaded tile into the wave-private LDS image
            bool big = false;   // TH05: an operand outside the verified range of the float quotient
#pragma unroll
            for (int it = 0; it < NE; ++it) {
                const int e = it * 64 + lane;
                bool ok = (ok_bits >> it) & 1u;
                if (FLAT) ok = ok && (((seg_bits >> it) & 1u) ? row_b_ok : row_a_ok);   // centre row interior
                if (it < NEA) ok = ok && (e >= PX * K::NC || cmid[it] > G.th_gt);  // centre fg
                const float v = raw[it];
                const float t = v > G.th_gt ? v : (v < G.bg_lt ? -(1.0f - v) : 0.0f);
                if (TH05) big = big || !(fabsf(t) <= 1024.0f);
                if (e < K::IMG) img[e] = ok ? t : 0.0f;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- prefetch the next tile; its latency hides behind this tile's votes
            have = next_tile(kz, ky);
            if (PREFETCH && have) load_tile(kz, ky);
            // ---- votes (fast path; redo the tile with true divisions if any lane saw an
            //      ambiguous rounding -- probability ~2^-26 per vote)
            const float *ia = img + pos_l;
            const float *ib = img + PX * K::NC + pos_l;
            if constexpr (TH05 && VAL == PPP_VAL_NORM_PROB_PRODUCT) {
                // float-only quotient; a tile with an out-of-range operand (never for
                // probabilities) takes the double divisions
                if (__ballot(big) == 0ull) {
                    if (row0) tile_votes<PX, K::NC, VAL, true, false, true>(ia, ib, u_ok, G.th2, G.den, inv_den, acc, cnt);
                    else tile_votes<PX, K::NC, VAL, false, false, true>(ia, ib, u_ok, G.th2, G.den, inv_den, acc, cnt);
                } else {
                    if (row0) tile_votes<PX, K::NC, VAL, true, true, true>(ia, ib, u_ok, G.th2, G.den, inv_den, acc, cnt);
                    else tile_votes<PX, K::NC, VAL, false, true, true>(ia, ib, u_ok, G.th2, G.den, inv_den, acc, cnt);
                }
            } else {
                float acc0[K::NACC];
                unsigned cnt0[K::NACC];
#pragma unroll
                for (int i = 0; i < K::NACC; ++i) { acc0[i] = acc[i]; cnt0[i] = cnt[i]; }
                const bool amb = row0 ? tile_votes<PX, K::NC, VAL, true, false, false>(ia, ib, u_ok, G.th2, G.den, inv_den, acc, cnt)
                                      : tile_votes<PX, K::NC, VAL, false, false, false>(ia, ib, u_ok, G.th2, G.den, inv_den, acc, cnt);
                if (VAL == PPP_VAL_NORM_PROB_PRODUCT && __ballot(amb) != 0ull) {
#pragma unroll
                    for (int i = 0; i < K::NACC; ++i) { acc[i] = acc0[i]; cnt[i] = cnt0[i]; }
                    if (row0) tile_votes<PX, K::NC, VAL, true, true, false>(ia, ib, u_ok, G.th2, G.den, inv_den, acc, cnt);
                    else tile_votes<PX, K::NC, VAL, false, true, false>(ia, ib, u_ok, G.th2, G.den, inv_den, acc, cnt);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (!lane_ok) return;
#pragma unroll
    for (int i = 0; i < K::NACC; ++i) {
        const int dx = i - (PX - 1);
        if (dz == 0 && dy == 0 && dx <= 0) continue;
        const long long o = cons_at(G, dz, dy, dx, uz, uy_l, ux);
        const float c = (float)cnt[i];
        if (cons) cons[o] = (G.normalise && cnt[i] != 0u) ? acc[i] / c : acc[i];
        if (cnt_out) cnt_out[o] = c;
    }
}

template <typename T, int PX, bool FLAT>
static hipError_t launch_v2f(const T *pred, const uint8_t *ov, float *cons, float *cnt,
                             const Geo &G, hipStream_t s) {
    const int n_rows = (G.pz - 1) * G.wy + G.py;
    // FLAT: runs of 64 over the flattened (y, x) slice of the box; else runs per line
    const int runs_per_line = FLAT ? (int)(((long long)G.bX * G.bY + 63) / 64) : (G.bX + 63) / 64;
    const long long n_waves = (long long)runs_per_line * (FLAT ? 1 : G.bY) * G.bZ * n_rows;
    const long long n_blocks = (n_waves + V2_WAVES - 1) / V2_WAVES;
    if (n_blocks >= (1ll << 31)) return hipErrorInvalidValue;
    PPP_GRID_CHECK(n_blocks, 64 * V2_WAVES);
    // per-lane element offsets are 32-bit byte offsets within PX channel volumes (+ one line)
    if (((long long)(PX - 1) * G.V + 2ll * G.X) * (long long)sizeof(T) >= (1ll << 32)) return hipErrorNotSupported;
    const dim3 grid((unsigned)n_blocks), block(64 * V2_WAVES);
    static EnvSwitch no_th05("PPP_S1_NO_TH05");
    if (G.value_rule == PPP_VAL_NORM_PROB_PRODUCT && G.th2 == 0.25 && G.den == 0.75 &&
        G.bg_lt <= 0.5f && !no_th05.get())
        consensus_v2_kernel<T, PX, PPP_VAL_NORM_PROB_PRODUCT, true, FLAT><<<grid, block, 0, s>>>(
            pred, ov, cons, cnt, G, n_rows, runs_per_line, n_waves);
    else if (G.value_rule == PPP_VAL_NORM_PROB_PRODUCT)
        consensus_v2_kernel<T, PX, PPP_VAL_NORM_PROB_PRODUCT, false, FLAT><<<grid, block, 0, s>>>(
            pred, ov, cons, cnt, G, n_rows, runs_per_line, n_waves);
    else if (G.value_rule == PPP_VAL_PROB_PRODUCT)
        consensus_v2_kernel<T, PX, PPP_VAL_PROB_PRODUCT, false, FLAT><<<grid, block, 0, s>>>(
            pred, ov, cons, cnt, G, n_rows, runs_per_line, n_waves);
    else
        consensus_v2_kernel<T, PX, PPP_VAL_COUNT, false, FLAT><<<grid, block, 0, s>>>(
            pred, ov, cons, cnt, G, n_rows, runs_per_line, n_waves);
    return hipGetLastError();
}

template <typename T, int PX>
static hipError_t launch_v2(const T *pred, const uint8_t *ov, float *cons, float *cnt,
                            const Geo &G, hipStream_t s) {
    // Flattened runs pay when the lines leave idle lanes (bX not a multiple of 64); they need
    // lines of at least 64 base voxels (a run then touches two lines at most), a patch with
    // py >= 3 (see load_tile) and more than one line.  PPP_S1_FLAT=0 / 1 overrides.
    static EnvSwitch sw("PPP_S1_FLAT");
    const char *e = sw.get();
    bool flat = G.bX >= 64 && G.bX % 64 != 0 && G.py >= 3 && G.bY > 1;
    if (e && e[0] == '0') flat = false;
    if (e && e[0] == '1' && G.bX >= 64 && G.py >= 3) flat = true;
    return flat ? launch_v2f<T, PX, true>(pred, ov, cons, cnt, G, s)
                : launch_v2f<T, PX, false>(pred, ov, cons, cnt, G, s);
}

// ---- wide patches (px = 25: the 2-d configuration) ---------------------------------------------
// Same decomposition, but a lane cannot hold 2*PX-1 = 49 accumulator pairs and the per-lane
// staging descriptors of a 2*PX*NC image: a wave takes a WINDOW of NW consecutive dx (the offset
// row is split over ceil((2 PX - 1) / NW) waves, each staging the full images again), and the
// images are staged by a rolled loop that derives every element's address on the fly.
template <int PX>
struct V2W {
    static constexpr int RX = PX / 2;
    static constexpr int NC = 64 + (PX - 1);     // centres per run
    static constexpr int NT = 64 + 2 * (PX - 1); // target pixels per run
    static constexpr int NACC = 2 * PX - 1;
#ifndef PPP_S1W_NW
#define PPP_S1W_NW 25
#endif
    static constexpr int NW = PPP_S1W_NW;        // accumulators per wave
    static constexpr int NWIN = (NACC + NW - 1) / NW;
    static constexpr int IMG = 2 * PX * NC;
};

template <int PX, int WIN, int VAL, bool ROW0, bool EXACT, bool TH05>
__device__ __forceinline__ bool tile_votes_w(const float *ia, const float *ib, const bool u_ok,
                                             const double th2, const double den, const double inv_den,
                                             float (&acc)[V2W<PX>::NW], unsigned (&cnt)[V2W<PX>::NW]) {
    using K = V2W<PX>;
    constexpr int I0 = WIN * K::NW;
    unsigned amb_min = 0xFFFFFFFFu;
#pragma unroll
    for (int kx = PX - 1; kx >= 0; --kx) {
        // partner columns of this kx inside the window: i = j - kx + PX - 1 in [I0, I0 + NW)
        const int j_lo = max(ROW0 ? kx + 1 : 0, I0 + kx - (PX - 1));
        const int j_hi = min(PX - 1, I0 + K::NW - 1 + kx - (PX - 1));
        if (j_lo > j_hi) continue;
        const float ta = u_ok ? ia[kx * K::NC - kx] : 0.0f;
        int lim = ta < 0.0f ? 0 : (int)0x80000000;
        asm volatile("" : "+v"(lim));
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            if (j < j_lo || j > j_hi) continue;
            vote<VAL, EXACT, TH05>(th2, den, inv_den, ta, lim, ib[j * K::NC - kx], acc[j - kx + PX - 1 - I0],
                                   cnt[j - kx + PX - 1 - I0], amb_min);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return amb_min <= 8u;
}

template <typename T, int PX, int WIN, int VAL, bool TH05>
__global__ void __launch_bounds__(64, 2)
    consensus_wide_kernel(const T *__restrict__ pred, const uint8_t *__restrict__ ov,
                          float *__restrict__ cons, float *__restrict__ cnt_out, const Geo G,
                          const int n_rows, const int runs_per_line, const long long n_waves) {
    using K = V2W<PX>;
    __shared__ float img[K::IMG];
    __shared__ uint8_t lds_valid[2][K::NT + 2];
    const int lane = threadIdx.x;
    const long long wid = blockIdx.x;
    if (wid >= n_waves) return;
    const int row = (int)(wid % n_rows);
    long long run = wid / n_rows;
    int dz, dy;
    if (row < G.py) { dz = 0; dy = row; }
    else { const int t = row - G.py; dz = 1 + t / G.wy; dy = t % G.wy - (G.py - 1); }
    const int xr = (int)(run % runs_per_line);
    run /= runs_per_line;
    const int uy = G.by0 + (int)(run % G.bY), uz = G.bz0 + (int)(run / G.bY), ux0 = G.bx0 + xr * 64;
    const int ux = ux0 + lane;
    const bool lane_ok = ux < G.bx0 + G.bX;
    const int wz = uz + dz, wy = uy + dy;
    const bool w_row_ok = wz < G.Z && wy >= 0 && wy < G.Y;
    const bool row0 = dz == 0 && dy == 0;
    constexpr int I0 = WIN * K::NW;

    float acc[K::NW];
    unsigned cnt[K::NW];
#pragma unroll
    for (int i = 0; i < K::NW; ++i) { acc[i] = 0.0f; cnt[i] = 0u; }
    const T *mid = pred + (long long)G.mid * G.V;
    uint8_t *uval = lds_valid[0], *wval = lds_valid[1];
    for (int i = lane; i < K::NT; i += 64) {
        const int x = ux0 - (PX - 1) + i;
        bool vu = false, vw = false;
        if (x >= 0 && x < G.X) {
            const long long lu = vox(G, uz, uy, x);
            vu = ldf(mid, lu) > G.th_gt && (!G.use_overlap || ov[lu] == 0);
            if (w_row_ok) {
                const long long lw = vox(G, wz, wy, x);
                vw = ldf(mid, lw) > G.th_gt && (!G.use_overlap || ov[lw] == 0);
            }
        }
        uval[i] = vu; wval[i] = vw;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool u_ok = lane_ok && uval[lane + PX - 1];
    const double inv_den = 1.0 / G.den;
    const int pos_l = lane + (PX - 1);

    if (w_row_ok) {
        const int kz_hi = min(G.pz - 1, G.pz - 1 - dz), kz_lo = max(0, -dz);
        const int ky_hi = min(G.py - 1, G.py - 1 - dy), ky_lo = max(0, -dy);
        for (int kz = kz_hi; kz >= kz_lo; --kz) {
            const int cz = uz - kz + G.rz;
            if (!(cz >= G.rz && cz < G.Z - G.rz)) continue;
            for (int ky = ky_hi; ky >= ky_lo; --ky) {
                const int cy = uy - ky + G.ry;
                if (!(cy >= G.ry && cy < G.Y - G.ry)) continue;
                // ---- stage + classify both images (rolled: element e -> half, column j, centre i)
                const long long crow = vox(G, cz, cy, 0);
                const long long cha = (long long)((kz * G.py + ky) * PX) * G.V + crow;
                const long long chb = (long long)(((kz + dz) * G.py + (ky + dy)) * PX) * G.V + crow;
                bool big = false;
                // U elements at a time: all their (independent) loads first, then the
                // classification -- one element per iteration waits a full memory latency each
#ifndef PPP_S1W_U
#define PPP_S1W_U 8
#endif
                constexpr int U = PPP_S1W_U;
                int i = lane, j = 0;                       // element e = j * NC + i (+ PX*NC for "about w")
                static_assert(K::NC >= 64, "one wrap per step of 64");
                for (int e0 = lane; e0 < K::IMG; e0 += 64 * U) {
                    float v[U], m[U];
                    bool okx[U];
                    int ti[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const bool in = e0 + 64 * u < K::IMG;
                        const int ju = in ? j : 0, iu = in ? i : 0;
                        const bool is_b = ju >= PX;
                        const int jj = is_b ? ju - PX : ju;
                        const int cx = ux0 - (PX - 1) + K::RX + iu;
                        okx[u] = in && cx >= K::RX && cx < G.X - K::RX;
                        const int cxc = min(max(cx, 0), G.X - 1);
                        v[u] = ldf(pred, (is_b ? chb : cha) + (long long)jj * G.V + cxc);
                        m[u] = is_b ? 1.0f : ldf(mid, crow + cxc);
                        ti[u] = (is_b ? K::NT + 2 : 0) + iu + jj;        // index into uval / wval (contiguous arrays)
                        i += 64;
                        if (i >= K::NC) { i -= K::NC; ++j; }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int e = e0 + 64 * u;
                        if (e < K::IMG) {
                            const bool ok = okx[u] && lds_valid[0][ti[u]] != 0 && (m[u] > G.th_gt || ti[u] >= K::NT + 2);
                            const float t = v[u] > G.th_gt ? v[u] : (v[u] < G.bg_lt ? -(1.0f - v[u]) : 0.0f);
                            if (TH05) big = big || !(fabsf(t) <= 1024.0f);
                            img[e] = ok ? t : 0.0f;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const float *ia = img + pos_l;
                const float *ib = img + PX * K::NC + pos_l;
                if constexpr (TH05 && VAL == PPP_VAL_NORM_PROB_PRODUCT) {
                    if (__ballot(big) == 0ull) {
                        if (row0) tile_votes_w<PX, WIN, VAL, true, false, true>(ia, ib, u_ok, G.th2, G.den, inv_den, acc, cnt);
                        else tile_votes_w<PX, WIN, VAL, false, false, true>(ia, ib, u_ok, G.th2, G.den, inv_den, acc, cnt);
                    } else {
                        if (row0) tile_votes_w<PX, WIN, VAL, true, true, true>(ia, ib, u_ok, G.th2, G.den, inv_den, acc, cnt);
                        else tile_votes_w<PX, WIN, VAL, false, true, true>(ia, ib, u_ok, G.th2, G.den, inv_den, acc, cnt);
                    }
                } else {
                    // general rules: the exact form (double division where the rule has one)
                    if (row0) tile_votes_w<PX, WIN, VAL, true, true, false>(ia, ib, u_ok, G.th2, G.den, inv_den, acc, cnt);
                    else tile_votes_w<PX, WIN, VAL, false, true, false>(ia, ib, u_ok, G.th2, G.den, inv_den, acc, cnt);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (!lane_ok) return;
#pragma unroll
    for (int k = 0; k < K::NW; ++k) {
        const int i = I0 + k;
        if (i >= K::NACC) continue;
        const int dx = i - (PX - 1);
        if (dz == 0 && dy == 0 && dx <= 0) continue;
        const long long o = cons_at(G, dz, dy, dx, uz, uy, ux);
        const float c = (float)cnt[k];
        if (cons) cons[o] = (G.normalise && cnt[k] != 0u) ? acc[k] / c : acc[k];
        if (cnt_out) cnt_out[o] = c;
    }
}

template <typename T, int PX, int VAL, bool TH05>
static hipError_t launch_wide_rule(const T *pred, const uint8_t *ov, float *cons, float *cnt,
                                   const Geo &G, hipStream_t s) {
    using K = V2W<PX>;
    static_assert(K::NWIN <= 4, "launcher written for up to four windows");
    const int n_rows = (G.pz - 1) * G.wy + G.py;
    const int runs_per_line = (G.bX + 63) / 64;
    const long long n_waves = (long long)runs_per_line * G.bY * G.bZ * n_rows;
    if (n_waves >= (1ll << 31)) return hipErrorInvalidValue;
    PPP_GRID_CHECK(n_waves, 64);
    const dim3 grid((unsigned)n_waves), block(64);
    consensus_wide_kernel<T, PX, 0, VAL, TH05><<<grid, block, 0, s>>>(pred, ov, cons, cnt, G, n_rows, runs_per_line, n_waves);
    if constexpr (K::NWIN > 1)
        consensus_wide_kernel<T, PX, 1, VAL, TH05><<<grid, block, 0, s>>>(pred, ov, cons, cnt, G, n_rows, runs_per_line, n_waves);
    if constexpr (K::NWIN > 2)
        consensus_wide_kernel<T, PX, 2, VAL, TH05><<<grid, block, 0, s>>>(pred, ov, cons, cnt, G, n_rows, runs_per_line, n_waves);
    if constexpr (K::NWIN > 3)
        consensus_wide_kernel<T, PX, 3, VAL, TH05><<<grid, block, 0, s>>>(pred, ov, cons, cnt, G, n_rows, runs_per_line, n_waves);
    return hipGetLastError();
}

template <typename T, int PX>
static hipError_t launch_wide(const T *pred, const uint8_t *ov, float *cons, float *cnt,
                              const Geo &G, hipStream_t s) {
    if (G.value_rule == PPP_VAL_NORM_PROB_PRODUCT && G.th2 == 0.25 && G.den == 0.75 && G.bg_lt <= 0.5f)
        return launch_wide_rule<T, PX, PPP_VAL_NORM_PROB_PRODUCT, true>(pred, ov, cons, cnt, G, s);
    if (G.value_rule == PPP_VAL_NORM_PROB_PRODUCT)
        return launch_wide_rule<T, PX, PPP_VAL_NORM_PROB_PRODUCT, false>(pred, ov, cons, cnt, G, s);
    if (G.value_rule == PPP_VAL_PROB_PRODUCT)
        return launch_wide_rule<T, PX, PPP_VAL_PROB_PRODUCT, false>(pred, ov, cons, cnt, G, s);
    return launch_wide_rule<T, PX, PPP_VAL_COUNT, false>(pred, ov, cons, cnt, G, s);
}

// returns hipErrorNotSupported when the shape has no specialised kernel
hipError_t launch_consensus_v2(const void *pred, int dtype, const uint8_t *ov, float *cons,
                               float *cnt, const Geo &G, hipStream_t s) {
#define PPP_V2_CASE(P)                                                                          \
    case P:                                                                                     \
        return dtype == PPP_F16                                                                 \
                   ? launch_v2<__half, P>((const __half *)pred, ov, cons, cnt, G, s)            \
                   : launch_v2<float, P>((const float *)pred, ov, cons, cnt, G, s);
    switch (G.px) {
        PPP_V2_CASE(3)
        PPP_V2_CASE(5)
        PPP_V2_CASE(7)
        PPP_V2_CASE(9)
    case 25:
        // (2-d patches only: a 25-wide 3-d patch has 15 625 channels)
        static EnvSwitch wide("PPP_S1_WIDE");
        if (G.pz != 1 || (wide.get() && wide.get()[0] == '0')) return hipErrorNotSupported;
        return dtype == PPP_F16 ? launch_wide<__half, 25>((const __half *)pred, ov, cons, cnt, G, s)
                                : launch_wide<float, 25>((const float *)pred, ov, cons, cnt, G, s);
    default:
        return hipErrorNotSupported;
    }
#undef PPP_V2_CASE
}

}  // namespace ppp
