// ppp_consensus_v4.hip -- S1, fourth generation: v3's packed two-slice vote chain with the
// accumulators of a run split over the TWO waves of a workgroup.
//
// Same sums, same summation order, bit-identical output as ppp_consensus_v3.hip (whose header
// describes the chain: wave = 64 base voxels x one offset row (dz, dy) x two z-slices, operands
// classified into LDS images, 8 packed vector instructions per pair of votes).  What limited v3 at
// 9^3 (profiles/r04_r_s1_ablations.txt, r04_y_s1_half_lds_bytes.txt): a lane holds the 2 PX - 1
// accumulators of all dx for both slices plus their counts and the register prefetch of the next
// tile -- 256 VGPRs, two waves per SIMD, and the latency of the LDS reads of the chain is exposed
// (40 % of the issue slots stay empty).  Here:
//
//   * a workgroup = TWO waves that work on the SAME run, offset row and slice pair and share ONE
//     pair of operand images in LDS.  Wave h owns the accumulators i = dx + PX - 1 with
//     owner(i) == h: wave 1 the PX - 2 accumulators dx = 0 ... PX - 3, wave 0 the others
//     (dx < 0 and the two outermost positive ones) -- 39 / 42 of the 81 (kx, j) combinations of a
//     tile at 9^3, 24 / 25 at 7^3.  An accumulator still receives its votes in raster order of the
//     voting centre (kz, ky, kx descending): the order inside ONE accumulator is what defines the
//     float sum, and an accumulator never changes hands.
//   * both waves stage: the elements of a tile's two images are dealt out alternately (wave h
//     loads / classifies staging iteration it = 2 i + h), so the register prefetch of the next
//     tile is half as large per wave, and so are the per-lane offset tables.
//   * the code of the two waves differs only in compile-time ownership: the kernel branches once
//     (wave-uniform) into body<0> / body<1>.
//   * workgroup barriers are `s_waitcnt lgkmcnt(0); s_barrier` (inline asm): a __syncthreads()
//     would also wait for the prefetch loads of the next tile (vmcnt(0)).
//
// fillConsensusArray.cu:5-218 / normConsensusArray.cu:5-43 are what is computed.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "ppp_kernels.hpp"

namespace ppp {
namespace s1v4 {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const volatile __attribute__((address_space(3))) v4f *lds_v4f_cvp;

#ifndef PPP_S1V4_GROUP
#define PPP_S1V4_GROUP 3
#endif
#ifndef PPP_S1V4_MINWAVES
#define PPP_S1V4_MINWAVES(PX) ((PX) <= 7 ? 4 : 3)
#endif

template <int PX, bool FLAT>
struct V4 {
    static constexpr int RX = PX / 2;
    static constexpr int NC = 64 + (FLAT ? 2 : 1) * (PX - 1);   // centres per run (both segments)
    static constexpr int NT = 64 + (FLAT ? 4 : 2) * (PX - 1);   // target pixels per run
    static constexpr int NACC = 2 * PX - 1;
    static constexpr int NEL = PX * NC;                         // elements of one operand image
    static constexpr int NIT = (NEL + 63) / 64;                 // staging iterations per image (both waves)
    static constexpr int NELP = NIT * 64;
    static constexpr int FOLD = 255 / PX;                       // tiles between two count folds
};

// accumulator i = dx + PX - 1 belongs to wave owner(PX, i)
__host__ __device__ constexpr int owner(int PX, int i) { return (i >= PX - 1 && i <= 2 * PX - 4) ? 1 : 0; }

// ---- packed helpers (as v3) ----------------------------------------------------------------
__device__ __forceinline__ v2f pk_mul_clamp(v2f a, v2f b) {
    v2f d;
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2f pk_fma_clamp(v2f a, v2f b, v2f c) {
    v2f d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// clamp01(-a + c)
__device__ __forceinline__ v2f pk_nadd_clamp(v2f a, v2f c) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[1,0] neg_hi:[1,0] clamp" : "=v"(d) : "v"(a), "v"(c));
    return d;
}
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ v2f splat(float x) { return (v2f){x, x}; }
__device__ __forceinline__ uint32_t pack_codes(v2f pos, v2f neg) {
    const v2f c = pk_fma(neg, splat(256.0f), pos);
    return (uint32_t)c.x | ((uint32_t)c.y << 16);
}
// LDS writes of this wave visible to the other wave of the workgroup (and its to this one); the
// global prefetch loads stay in flight
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The (kx, j) combinations of a tile that wave H executes, in processing order: kx descending
// (raster order of the centre), j ascending, cut into groups of at most GS inside one kx.
template <int PX, int H, bool ROW0, int GS>
struct Sched {
    static constexpr int MAXG = PX * ((PX + GS - 1) / GS) + PX;
    int n;
    int kx[MAXG], cnt[MAXG], j[MAXG][GS];
    bool first[MAXG];                                           // first group of its kx: the "about u" operand is read
    constexpr Sched() : n(0), kx{}, cnt{}, j{}, first{} {
        for (int k = PX - 1; k >= 0; --k) {
            int c = 0;
            bool f = true;
            for (int jj = 0; jj < PX; ++jj) {
                if (ROW0 && jj <= k) continue;
                if (owner(PX, jj - k + PX - 1) != H) continue;
                if (c == 0) { kx[n] = k; first[n] = f; f = false; }
                j[n][c++] = jj;
                if (c == GS) { cnt[n++] = c; c = 0; }
            }
            if (c) cnt[n++] = c;
        }
    }
};

// all votes of one (kz, ky) that wave H owns, both slices.  The "about w" image holds 16-byte
// elements {t slice 0, t slice 1, codes, -}; the LDS reads of the next group are issued before
// the arithmetic of the current one.
template <int PX, int NC, int H, bool ROW0, bool EXACT>
__device__ __forceinline__ void tile_votes4(const v2f *at, lds_v4f_cvp bt, const double th2, const double den,
                                            v2f (&acc)[2 * PX - 1], uint32_t (&tc)[2 * PX - 1]) {
    const v2f kq = splat(-0.25f), c43 = splat(0x1.555556p+0f), c43lo = splat(-0x1.555556p-25f);
    constexpr int GS = PPP_S1V4_GROUP;
    constexpr Sched<PX, H, ROW0, GS> S{};
    constexpr int NG = S.n;
    if constexpr (NG == 0) return;
    v4f bcur[GS], bnxt[GS];
    v2f ta, ta_nxt, ga, na;
    uint32_t ca = 0u;
    auto load_group = [&](auto n_c, v4f (&b)[GS], v2f &t) {
        constexpr int n = decltype(n_c)::value;
        constexpr int kx = S.kx[n];
        if constexpr (S.first[n]) t = at[kx * NC - kx];
#pragma unroll
        for (int g = 0; g < GS; ++g)
            if (g < S.cnt[n]) b[g] = bt[S.j[n][g] * NC - kx];
    };
    load_group(std::integral_constant<int, 0>{}, bcur, ta);
    auto step = [&](auto n_c) {
        constexpr int n = decltype(n_c)::value;
        constexpr int kx = S.kx[n];
        constexpr int ng = S.cnt[n];
        if constexpr (n + 1 < NG) load_group(std::integral_constant<int, n + 1>{}, bnxt, ta_nxt);
        if constexpr (S.first[n]) {
            if constexpr (!EXACT) {
                ga = pk_mul_clamp(ta, splat(4.0f));            // [ta > 0]  (|ta| > 0.5 when classified)
                na = pk_mul_clamp(ta, splat(-4.0f));           // [ta < 0]
            } else {
                ga = (v2f){ta.x > 0.0f ? 1.0f : 0.0f, ta.y > 0.0f ? 1.0f : 0.0f};
                na = (v2f){ta.x < 0.0f ? 1.0f : 0.0f, ta.y < 0.0f ? 1.0f : 0.0f};
            }
            ca = pack_codes(ga, na);
        }
        v2f tb[GS], x[GS], y[GS];
        uint32_t cb[GS];
#pragma unroll
        for (int g = 0; g < ng; ++g) {
            tb[g] = (v2f){bcur[g].x, bcur[g].y};
            cb[g] = __float_as_uint(bcur[g].z);
            x[g] = ta * tb[g];
        }
        if constexpr (!EXACT) {
            v2f dp[GS], dn[GS], d[GS], q0[GS];
#pragma unroll
            for (int g = 0; g < ng; ++g) dp[g] = pk_fma_clamp(x[g], ga, kq);
#pragma unroll
            for (int g = 0; g < ng; ++g) dn[g] = pk_nadd_clamp(x[g], kq);
#pragma unroll
            for (int g = 0; g < ng; ++g) d[g] = dp[g] - dn[g];
#pragma unroll
            for (int g = 0; g < ng; ++g) q0[g] = d[g] * c43lo;
#pragma unroll
            for (int g = 0; g < ng; ++g) y[g] = pk_fma(d[g], c43, q0[g]);
        } else {
#pragma unroll
            for (int g = 0; g < ng; ++g) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const float a = s ? ta.y : ta.x, b = s ? tb[g].y : tb[g].x, xs = s ? x[g].y : x[g].x;
                    const bool valid = a != 0.0f && b != 0.0f && !(a < 0.0f && b < 0.0f);
                    const double xd = (double)xs;
                    const float ys = (float)((xd - __builtin_copysign(th2, xd)) / den);
                    if (s) y[g].y = valid ? ys : 0.0f; else y[g].x = valid ? ys : 0.0f;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < ng; ++g) {
            const int i = S.j[n][g] - kx + PX - 1;
            acc[i] = acc[i] + y[g];
            tc[i] = pk_mad_u16(ca, cb[g], tc[i]);
        }
        if constexpr (n + 1 < NG) {
#pragma unroll
            for (int g = 0; g < GS; ++g) bcur[g] = bnxt[g];
            if constexpr (S.first[n + 1]) ta = ta_nxt;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // (template recursion instead of a loop: every index of the schedule is a constant expression)
    auto run = [&](auto self, auto n_c) {
        constexpr int n = decltype(n_c)::value;
        if constexpr (n < NG) {
            step(n_c);
            self(self, std::integral_constant<int, n + 1>{});
        }
    };
    run(run, std::integral_constant<int, 0>{});
}

template <typename T>
__device__ __forceinline__ float ldf_at4(const T *base, unsigned byte_off) {
    // (keeps the zero extension of the offset next to the load, as v3)
    asm volatile("" : "+v"(byte_off));
    return ldf(reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off), 0);
}

struct Smem {
    v4f *bt;
    v2f *at, *cf;
    uint8_t *valid;     // [2 slices][2 rows (u, w)][NT + 2]
    uint32_t *big;      // [2]
};

template <typename T, int PX, bool FLAT, int H>
__device__ __forceinline__ void body(const T *__restrict__ pred, const uint8_t *__restrict__ ov,
                                     float *__restrict__ cons, float *__restrict__ cnt_out, const Geo &G,
                                     const int n_rows, const int runs_per_line, const long long wid,
                                     const Smem sm, const int lane) {
    using K = V4<PX, FLAT>;
    constexpr int NIT = K::NIT;
    constexpr int NITH = (NIT - H + 1) / 2;                        // staging iterations of this wave: it = 2 i + H
    constexpr int VST = K::NT + 2;
    const int row = (int)(wid % n_rows);
    long long run = wid / n_rows;
    int dz, dy;
    if (row < G.py) { dz = 0; dy = row; }
    else { const int t = row - G.py; dz = 1 + t / G.wy; dy = t % G.wy - (G.py - 1); }
    const int xr = (int)(run % runs_per_line);
    run /= runs_per_line;
    int uy, uz, ux0, nA;
    if (FLAT) {
        const int flat0 = xr * 64;
        uy = G.by0 + flat0 / G.bX;
        uz = G.bz0 + 2 * (int)run;
        ux0 = G.bx0 + flat0 % G.bX;
        nA = min(64, G.bX - flat0 % G.bX);
    } else {
        uy = G.by0 + (int)(run % G.bY);
        uz = G.bz0 + 2 * (int)(run / G.bY);
        ux0 = G.bx0 + xr * 64;
        nA = 64;
    }
    const bool have_s1 = uz + 1 < G.bz0 + G.bZ;                    // slice 1 exists (workgroup-uniform)
    const bool in_b = FLAT && lane >= nA;                          // this lane sits on line B
    const bool have_b = FLAT && nA < 64 && uy + 1 < G.by0 + G.bY;  // (workgroup-uniform)
    const int ux = in_b ? G.bx0 + (lane - nA) : ux0 + lane;
    const int uy_l = in_b ? uy + 1 : uy;
    const bool lane_ok = in_b ? have_b : ux < G.bx0 + G.bX;
    const int pos_l = lane + (PX - 1) + (in_b ? PX - 1 : 0);       // image column of the lane's centre at kx = PX-1
    const int wy = uy + dy;                                        // (line A; line B: wy + 1)
    const bool wy_ok_a = wy >= 0 && wy < G.Y;
    const bool wy_ok_b = have_b && wy + 1 >= 0 && wy + 1 < G.Y;
    const bool wz_ok0 = uz + dz < G.Z, wz_ok1 = have_s1 && uz + 1 + dz < G.Z;
    const bool w_row_ok = (wy_ok_a || wy_ok_b) && (wz_ok0 || wz_ok1);
    const bool row0 = dz == 0 && dy == 0;

    v2f acc[K::NACC];
    uint32_t tc[K::NACC], cnt[K::NACC];
#pragma unroll
    for (int i = 0; i < K::NACC; ++i) { acc[i] = splat(0.0f); tc[i] = 0u; cnt[i] = 0u; }

    const T *mid = pred + (long long)G.mid * G.V;
    // validity (foreground && !overlap) of the target pixels on the u row and on the w row: wave H
    // fills slice H; applied when the accumulators are written
    const int ntA = nA + 2 * (PX - 1);
    {
        constexpr int s = H;
        uint8_t *uval = sm.valid + (s * 2 + 0) * VST, *wval = sm.valid + (s * 2 + 1) * VST;
        const bool s_ok = s == 0 || have_s1;
        const bool wz_ok = s ? wz_ok1 : wz_ok0;
        for (int i = lane; i < K::NT; i += 64) {
            const bool sb = FLAT && i >= ntA;
            const int x = sb ? G.bx0 - (PX - 1) + (i - ntA) : ux0 - (PX - 1) + i;
            const int yy = sb ? uy + 1 : uy, wyy = sb ? wy + 1 : wy;
            bool vu = false, vw = false;
            if (s_ok && x >= 0 && x < G.X && (!sb || have_b)) {
                const long long lu = vox(G, uz + s, yy, x);
                vu = ldf(mid, lu) > G.th_gt && (!G.use_overlap || ov[lu] == 0);
                if (wz_ok && (sb ? wy_ok_b : wy_ok_a)) {
                    const long long lw = vox(G, uz + s + dz, wyy, x);
                    vw = ldf(mid, lw) > G.th_gt && (!G.use_overlap || ov[lw] == 0);
                }
            }
            uval[i] = vu; wval[i] = vw;
        }
    }

    // per-lane description of the elements this wave stages for every tile (as v3, every second
    // iteration) and of the ONE centre whose factor this lane computes (centre lane + 64 H)
    unsigned el_off[NITH > 0 ? NITH : 1];
    unsigned cf_idx[NITH > 0 ? NITH : 1];
    const int ncA = nA + (PX - 1);                    // centres of segment A
    auto centre_x = [&](int i, bool &sb, bool &ok) -> int {
        sb = FLAT && i >= ncA;
        const int iseg = sb ? i - ncA : i;
        const int cx = (sb ? G.bx0 : ux0) - (PX - 1) + K::RX + iseg;
        ok = cx >= K::RX && cx < G.X - K::RX && (!sb || have_b);
        return min(max(cx, 0), G.X - 1) + (sb ? G.X : 0);
    };
#pragma unroll
    for (int q = 0; q < NITH; ++q) {
        const int e = (2 * q + H) * 64 + lane;
        const int i = e % K::NC;
        const int j = min(e / K::NC, PX - 1);         // (padding elements repeat the last column)
        bool sb, ok;
        const int cxc = centre_x(i, sb, ok);
        cf_idx[q] = (unsigned)i;
        el_off[q] = (unsigned)(((long long)j * G.V + cxc) * (long long)sizeof(T));
    }
    unsigned cf_off;
    bool cf_st, cf_sb;
    {
        const int i = min(lane + 64 * H, K::NC - 1);
        bool sb, ok;
        const int cxc = centre_x(i, sb, ok);
        cf_off = (unsigned)cxc * (unsigned)sizeof(T);
        cf_st = ok && lane + 64 * H < K::NC;
        cf_sb = sb;
    }

    v2f *at = sm.at, *cf = sm.cf;
    v4f *bt = sm.bt;
    wg_barrier();

    if (w_row_ok) {
        const int kz_hi = min(G.pz - 1, G.pz - 1 - dz), kz_lo = max(0, -dz);
        const int ky_hi = min(G.py - 1, G.py - 1 - dy), ky_lo = max(0, -dy);
        const long long slice = (long long)G.Y * G.X;
        // tiles (kz, ky) in descending order, skipping tiles without an interior centre row
        int kz = kz_hi, ky = ky_hi + 1;
        bool row_a_ok = true, row_b_ok = false;   // centre row inside the y-interior, per segment
        bool z_ok0 = true, z_ok1 = false;         // centre slice inside the z-interior, per slice
        auto next_tile = [&](int &z, int &y) -> bool {
            while (true) {
                if (--y < ky_lo) { y = ky_hi; --z; }
                if (z < kz_lo) return false;
                const int cz = uz - z + G.rz, cy = uy - y + G.ry;
                z_ok0 = cz >= G.rz && cz < G.Z - G.rz;
                z_ok1 = have_s1 && cz + 1 >= G.rz && cz + 1 < G.Z - G.rz;
                if (!(z_ok0 || z_ok1)) continue;
                row_a_ok = cy >= G.ry && cy < G.Y - G.ry;
                row_b_ok = have_b && cy + 1 >= G.ry && cy + 1 < G.Y - G.ry;
                if (row_a_ok || row_b_ok) return true;
            }
        };
        float ra[2][NITH > 0 ? NITH : 1], rb[2][NITH > 0 ? NITH : 1], rc[2];
        bool lz0 = true, lz1 = false, la = true, lb = false;   // flags of the LOADED tile
        auto load_tile = [&](int z, int y) {
            lz0 = z_ok0; lz1 = z_ok1; la = row_a_ok; lb = row_b_ok;
            const int cz = uz - z + G.rz;
            const long long crow0 = vox(G, z_ok0 ? cz : cz + 1, uy - y + G.ry, 0);
            const long long crow1 = z_ok1 ? crow0 + (z_ok0 ? slice : 0) : crow0;
            const long long cha = (long long)((z * G.py + y) * PX) * G.V;
            const long long chb = (long long)(((z + dz) * G.py + (y + dy)) * PX) * G.V;
#pragma unroll
            for (int q = 0; q < NITH; ++q) {
                ra[0][q] = ldf_at4(pred + cha + crow0, el_off[q]);
                ra[1][q] = ldf_at4(pred + cha + crow1, el_off[q]);
                rb[0][q] = ldf_at4(pred + chb + crow0, el_off[q]);
                rb[1][q] = ldf_at4(pred + chb + crow1, el_off[q]);
            }
            rc[0] = ldf_at4(mid + crow0, cf_off);
            rc[1] = ldf_at4(mid + crow1, cf_off);
        };
        const v2f big26 = splat(0x1p26f), nbig26 = splat(-0x1p26f), hb = splat(0x1p25f),
                  nhb = splat(-0x1p25f);
        int n_fold = 0;
        bool have = next_tile(kz, ky);
        if (have) load_tile(kz, ky);
        while (have) {
            // ---- centre factors: foreground && interior (x, y, z) && segment exists
            {
                const bool rok = cf_st && (cf_sb ? lb : la);
                v2f f;
                f.x = (rok && lz0 && rc[0] > G.th_gt) ? 1.0f : 0.0f;
                f.y = (rok && lz1 && rc[1] > G.th_gt) ? 1.0f : 0.0f;
                if (lane + 64 * H < K::NC) cf[lane + 64 * H] = f;
            }
            wg_barrier();
            // ---- classify this wave's share of both images of both slices into LDS
            //      t = v (v > 0.5), v - 1 (v < 0.5), 0 (v == 0.5):  g = [v > 0.5], h = [v < 0.5]
            unsigned bigmax = 0u;
#pragma unroll
            for (int q = 0; q < NITH; ++q) {
                const int e = (2 * q + H) * 64 + lane;
                const v2f cfe = cf[cf_idx[q]];
                {
                    const v2f v = {ra[0][q], ra[1][q]};
                    bigmax = max(bigmax, max(__float_as_uint(v.x), __float_as_uint(v.y)));
                    const v2f g = pk_fma_clamp(v, big26, nhb), h = pk_fma_clamp(v, nbig26, hb);
                    v2f t = pk_fma(v, g + h, -h);
                    t = t * cfe;
                    at[e] = t;
                }
                {
                    const v2f v = {rb[0][q], rb[1][q]};
                    bigmax = max(bigmax, max(__float_as_uint(v.x), __float_as_uint(v.y)));
                    const v2f g = pk_fma_clamp(v, big26, nhb), h = pk_fma_clamp(v, nbig26, hb);
                    const v2f t = pk_fma(v, g + h, -h);
                    bt[e] = (v4f){t.x, t.y, __uint_as_float(pack_codes(g, h)), 0.0f};
                }
            }
            // a value outside [0, 1] anywhere in the tile (either wave's share) sends the tile down
            // the exact path
            const bool my_big = __ballot(bigmax > 0x3F800000u) != 0ull;
            if (lane == 0) sm.big[H] = my_big ? 1u : 0u;
            wg_barrier();
            const bool any_big = (sm.big[0] | sm.big[1]) != 0u;
            if (any_big) {
#pragma unroll
                for (int q = 0; q < NITH; ++q) {
                    const int e = (2 * q + H) * 64 + lane;
                    const v2f f = cf[cf_idx[q]];
                    v2f ta, tb;
                    unsigned code = 0u;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const float va = ra[s][q], vb = rb[s][q];
                        float xa = va > G.th_gt ? va : (va < G.bg_lt ? va - 1.0f : 0.0f);
                        const float xb = vb > G.th_gt ? vb : (vb < G.bg_lt ? vb - 1.0f : 0.0f);
                        if ((s ? f.y : f.x) == 0.0f) xa = 0.0f;
                        if (s) { ta.y = xa; tb.y = xb; } else { ta.x = xa; tb.x = xb; }
                        code |= (xb > 0.0f ? 1u : (xb < 0.0f ? 256u : 0u)) << (16 * s);
                    }
                    at[e] = ta;
                    bt[e] = (v4f){tb.x, tb.y, __uint_as_float(code), 0.0f};
                }
                wg_barrier();
            }
            // ---- prefetch the next tile; its latency hides behind this tile's votes
            have = next_tile(kz, ky);
            if (have) load_tile(kz, ky);
            // ---- votes
            const v2f *ia = at + pos_l;
            lds_v4f_cvp ib = (lds_v4f_cvp)(bt + pos_l);
            if (!any_big) {
                if (row0) tile_votes4<PX, K::NC, H, true, false>(ia, ib, G.th2, G.den, acc, tc);
                else tile_votes4<PX, K::NC, H, false, false>(ia, ib, G.th2, G.den, acc, tc);
            } else {
                if (row0) tile_votes4<PX, K::NC, H, true, true>(ia, ib, G.th2, G.den, acc, tc);
                else tile_votes4<PX, K::NC, H, false, true>(ia, ib, G.th2, G.den, acc, tc);
            }
            if (++n_fold == K::FOLD) {
                n_fold = 0;
#pragma unroll
                for (int i = 0; i < K::NACC; ++i) {
                    if (owner(PX, i) != H) continue;
                    cnt[i] += (tc[i] & 0x00FF00FFu) + ((tc[i] >> 8) & 0x00FF00FFu);
                    tc[i] = 0u;
                }
            }
            wg_barrier();
        }
    }
    if (!lane_ok) return;
    const int ti_u = (in_b ? ntA + (lane - nA) : lane) + PX - 1;   // index of u in the validity rows
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (s == 1 && !have_s1) break;
        const uint8_t *uval = sm.valid + (s * 2 + 0) * VST, *wval = sm.valid + (s * 2 + 1) * VST;
        const bool u_ok = uval[ti_u] != 0;
        float val[K::NACC];
#pragma unroll
        for (int i = 0; i < K::NACC; ++i) {
            const int dx = i - (PX - 1);
            val[i] = 0.0f;
            if (owner(PX, i) != H) continue;
            if (dz == 0 && dy == 0 && dx <= 0) continue;
            const unsigned total = cnt[i] + (tc[i] & 0x00FF00FFu) + ((tc[i] >> 8) & 0x00FF00FFu);
            const bool ok = u_ok && wval[ti_u + dx] != 0;
            const unsigned n = ok ? ((s ? total >> 16 : total) & 0xFFFFu) : 0u;
            const float a = ok ? (s ? acc[i].y : acc[i].x) : 0.0f;
            const float c = (float)n;
            val[i] = (G.normalise && n != 0u) ? a / c : a;
            if (G.layout != PPP_CONS_VOXEL_MAJOR) {
                const long long o = cons_at(G, dz, dy, dx, uz + s, uy_l, ux);
                if (cons) cons[o] = val[i];
                if (cnt_out) cnt_out[o] = c;
            }
        }
        if (G.layout == PPP_CONS_VOXEL_MAJOR) {
            // symmetric voxel-major rows written directly (as v3): S[u][Lc + L(d)] -- this wave's
            // entries of the offset row, four at a time where four neighbours are its own -- and,
            // when w = u + d lies in the box, the mirror S[w][Lc - L(d)].
            const int W = (2 * G.pz - 1) * G.wy * G.wx, Lc = (W - 1) / 2;
            const int L0 = (dz * G.wy + dy) * G.wx;
            const long long vu = ((long long)(uz + s - G.bz0) * G.bY + (uy_l - G.by0)) * G.bX + (ux - G.bx0);
            float *pos = cons + vu * W + Lc + L0 - (PX - 1);     // entry of dx = -(PX-1)
            if (row0) {
                if (owner(PX, PX - 1) == H) pos[PX - 1] = 0.0f;  // offset 0
#pragma unroll
                for (int i = PX; i < K::NACC; ++i)
                    if (owner(PX, i) == H) pos[i] = val[i];
            } else {
                typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                // first index of this wave's two ranges: wave 0 owns [0, PX-2] and [2PX-3, 2PX-2],
                // wave 1 owns [PX-1, 2PX-4]
                constexpr int r0 = H ? PX - 1 : 0, r1 = H ? 2 * PX - 4 : PX - 2;
                constexpr int nvec = (r1 - r0 + 1) / 4;
#pragma unroll
                for (int g = 0; g < nvec; ++g) {
                    const int i = r0 + 4 * g;
                    *reinterpret_cast<f4u *>(pos + i) = (f4u){val[i], val[i + 1], val[i + 2], val[i + 3]};
                }
#pragma unroll
                for (int i = r0 + 4 * nvec; i <= r1; ++i) pos[i] = val[i];
                if (H == 0) { pos[2 * PX - 3] = val[2 * PX - 3]; pos[2 * PX - 2] = val[2 * PX - 2]; }
            }
            const int wz = uz + s + dz, wy2 = uy_l + dy;
            if (wz < G.bz0 + G.bZ && wy2 >= G.by0 && wy2 < G.by0 + G.bY) {
                const long long vw0 = vu + ((long long)dz * G.bY + dy) * G.bX;
#pragma unroll
                for (int i = 0; i < K::NACC; ++i) {
                    const int dx = i - (PX - 1);
                    if (owner(PX, i) != H) continue;
                    if (row0 && dx <= 0) continue;
                    const int wx2 = ux + dx;
                    if (wx2 >= G.bx0 && wx2 < G.bx0 + G.bX) cons[(vw0 + dx) * W + Lc - L0 - dx] = val[i];
                }
            }
        }
    }
}

template <typename T, int PX, bool FLAT>
__global__ void __launch_bounds__(128, PPP_S1V4_MINWAVES(PX))
    consensus_v4_kernel(const T *__restrict__ pred, const uint8_t *__restrict__ ov,
                        float *__restrict__ cons, float *__restrict__ cnt_out, const Geo G,
                        const int n_rows, const int runs_per_line, const long long n_units) {
    using K = V4<PX, FLAT>;
    __shared__ v4f lds_bt[K::NELP];
    __shared__ v2f lds_at[K::NELP];
    __shared__ v2f lds_cf[K::NC];
    __shared__ uint8_t lds_valid[2 * 2 * (K::NT + 2)];
    __shared__ uint32_t lds_big[2];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // XCD-aware order (as v2 / v3): each XCD gets a contiguous range of (x-run, row) work
    long long bid = blockIdx.x;
    {
        const long long nb = gridDim.x, per = nb / 8, main = per * 8;
        if (bid < main) bid = (bid % 8) * per + bid / 8;
    }
    if (bid >= n_units) return;
    const Smem sm = {lds_bt, lds_at, lds_cf, lds_valid, lds_big};
    if (wave == 0) body<T, PX, FLAT, 0>(pred, ov, cons, cnt_out, G, n_rows, runs_per_line, bid, sm, lane);
    else body<T, PX, FLAT, 1>(pred, ov, cons, cnt_out, G, n_rows, runs_per_line, bid, sm, lane);
}

template <typename T, int PX, bool FLAT>
static hipError_t launch_v4f(const T *pred, const uint8_t *ov, float *cons, float *cnt, const Geo &G, hipStream_t s) {
    const int n_rows = (G.pz - 1) * G.wy + G.py;
    const int runs_per_line = FLAT ? (int)(((long long)G.bX * G.bY + 63) / 64) : (G.bX + 63) / 64;
    const int bZ2 = (G.bZ + 1) / 2;
    const long long n_units = (long long)runs_per_line * (FLAT ? 1 : G.bY) * bZ2 * n_rows;
    if (n_units >= (1ll << 31)) return hipErrorInvalidValue;
    PPP_GRID_CHECK(n_units, 128);
    if (((long long)(PX - 1) * G.V + 2ll * G.X) * (long long)sizeof(T) >= (1ll << 32)) return hipErrorNotSupported;
    consensus_v4_kernel<T, PX, FLAT><<<dim3((unsigned)n_units), dim3(128), 0, s>>>(pred, ov, cons, cnt, G, n_rows,
                                                                                  runs_per_line, n_units);
    return hipGetLastError();
}

template <typename T, int PX>
static hipError_t launch_v4(const T *pred, const uint8_t *ov, float *cons, float *cnt, const Geo &G, hipStream_t s) {
    static EnvSwitch sw("PPP_S1_FLAT");        // same rule as v2 / v3
    const char *e = sw.get();
    bool flat = G.bX >= 64 && G.bX % 64 != 0 && G.py >= 3 && G.bY > 1;
    if (e && e[0] == '0') flat = false;
    if (e && e[0] == '1' && G.bX >= 64 && G.py >= 3) flat = true;
    return flat ? launch_v4f<T, PX, true>(pred, ov, cons, cnt, G, s) : launch_v4f<T, PX, false>(pred, ov, cons, cnt, G, s);
}

}  // namespace s1v4

// the shapes the two-wave kernel serves: v3's rule (TH = 0.5, normalised product) at px in {5, 7, 9}
// (at px = 3 an even split of the 5 accumulators does not exist).  Measured SLOWER than the
// one-wave kernel (profiles/r05_a_s1_v4.txt: 9^3 slab 280 vs 264 ms at three waves per SIMD, 140^3 /
// 7^3 52 vs 41 ms): the two waves meet at three barriers per tile and each repeats the scalar tile
// bookkeeping and the "about u" reads, which costs more than the third wave per SIMD hides.  It
// serves only when asked for (PPP_S1_V4=1) and stays under test as a second implementation.
bool consensus_v4_supported(const Geo &G) {
    static EnvSwitch sw("PPP_S1_V4");
    const char *e = sw.get();
    if (!e || e[0] != '1') return false;
    return consensus_v3_supported(G) && (G.px == 5 || G.px == 7 || G.px == 9) && !G.ring;
}

hipError_t launch_vm_zero_faces(float *S, const Geo &G, hipStream_t s);   // ppp_consensus_v3.hip

hipError_t launch_consensus_v4(const void *pred, int dtype, const uint8_t *ov, float *cons, float *cnt, const Geo &G,
                               hipStream_t s) {
    if (!consensus_v4_supported(G)) return hipErrorNotSupported;
    if (G.layout == PPP_CONS_VOXEL_MAJOR && (cnt || !cons)) return hipErrorInvalidValue;
    if (G.layout == PPP_CONS_VOXEL_MAJOR && !G.vm_open) {
        const hipError_t ez = launch_vm_zero_faces(cons, G, s);
        if (ez != hipSuccess) return ez;
    }
#define PPP_V4_CASE(P)                                                                          \
    case P:                                                                                     \
        return dtype == PPP_F16 ? s1v4::launch_v4<__half, P>((const __half *)pred, ov, cons, cnt, G, s) \
                                : s1v4::launch_v4<float, P>((const float *)pred, ov, cons, cnt, G, s);
    switch (G.px) {
        PPP_V4_CASE(5)
        PPP_V4_CASE(7)
        PPP_V4_CASE(9)
    default:
        return hipErrorNotSupported;
    }
#undef PPP_V4_CASE
}

}  // namespace ppp
