// ppp_consensus_v3.hip -- S1, third generation: TWO z-slices per lane, packed f32 arithmetic.
//
// Same sums, same summation order, bit-identical output as ppp_consensus_v2.hip (whose header
// describes the work decomposition: wave = 64 base voxels x one offset row (dz, dy), lane = one
// base voxel with the 2*PX-1 accumulators of all dx, loops over the patch rows (kz, ky), kx and
// the partner column unrolled).  What is new:
//
//   * a lane owns the base voxels (uz, uy, ux) AND (uz + 1, uy, ux).  Every quantity of the vote
//     chain is a float2 {slice 0, slice 1} in an aligned register pair, the LDS images hold the
//     two slices interleaved (one ds_read_b64 per operand pair, conflict free), and the chain is
//     executed with v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: 8 vector instructions per PAIR
//     of votes instead of 11 per vote.  (Pairing two votes of ONE voxel is not possible: the
//     accumulator pair (i, i+1) and the LDS pair would need opposite alignments for odd / even
//     kx.)  The x / y geometry (flattened runs, segments) is shared by the two slices.
//   * no compares / selects in the chain.  With TH = 0.5 a classified operand is t = v (> 0.5),
//     t = v - 1 (< -0.5) or 0, x = ta*tb, and the vote is valid iff x != 0 and not both
//     operands negative.  Per kx the lane forms ga = [ta > 0] as a float; then
//         dp = clamp01(x*ga - 0.25)      (positive votes; 0 for x <= 0 and for ta < 0)
//         dn = clamp01(-x - 0.25)        (negative votes; 0 for x >= 0, i.e. also for (neg, neg))
//         d  = dp - dn                   = x -/+ 0.25 exactly, or +0 for an invalid pair
//     followed by a two-operation correctly rounded quotient y = fma(d, fl(4/3), d * lo(4/3)) and
//     acc += y (y = +0 for an invalid pair).  clamp01 is the VOP3P clamp bit (inline asm).  Needs |x| <= 1.25: any value
//     of the tile outside [0, 1] (never for probabilities) sends the tile down the exact path
//     (compares + double division, as v2).
//   * counts: operand codes {0, 1 (pos), 256 (neg)} as 16-bit integers, one v_pk_mad_u16 per
//     vote pair: pos*pos adds 1 to the low byte, pos*neg / neg*pos add 1 to the high byte and
//     neg*neg = 65536 vanishes mod 2^16.  The two bytes are folded into the running counts
//     before they can overflow.
//   * masks by construction instead of per element: everything that depends on the CENTRE
//     (foreground, interior in x / y / z, segment exists) is one float factor cf[centre][slice]
//     multiplied into the "about u" operands; everything that depends on the TARGET pixels
//     (u and w = u + d valid foreground) is constant per accumulator and applied once, when the
//     accumulators are written.
#include <stdlib.h>

#include <algorithm>

#include "ppp_kernels.hpp"

namespace ppp {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
// (explicit LDS address space: a volatile access through a generic pointer stays a flat load)
typedef const volatile __attribute__((address_space(3))) v4f *lds_v4f_cvp;

// waves per workgroup: LDS (16 / 20 KB per wave at 7^3 / 9^3) decides how many waves a CU holds;
// small workgroups waste less of it
#ifndef PPP_S1V3_WAVES
#define PPP_S1V3_WAVES(PX) 1
#endif
// votes whose chains are interleaved (dependent packed operations issue back to back otherwise)
#ifndef PPP_S1V3_GROUP
#define PPP_S1V3_GROUP 3
#endif
#ifndef PPP_S1V3_PREFETCH
#define PPP_S1V3_PREFETCH(PX) ((PX) >= 9)
#endif
#ifndef PPP_S1V3_MINWAVES
#define PPP_S1V3_MINWAVES(PX) ((PX) <= 7 ? 3 : 2)
#endif

// "about w" image: PPP_S1V3_SPLIT=1 keeps {t slice 0, t slice 1} (8 bytes) and the codes (4 bytes)
// in two arrays -- a ds_read_b64 and a ds_read_b32 per vote pair, 12 bytes instead of the 16 of
// the {t0, t1, codes, pad} element read by ds_read_b128: the LDS pipe is what this kernel keeps
// busiest (~80 % next to 46 % VALU issue), and a quarter of its bytes were padding.
// Measured (tools/time_s1.py, profiles/r04_q_s1_split.txt): 7^3 46.2 -> 41.3 ms on the 140^3 volume
// and 66.0 -> 62.7 ms on the 512^2 slab; 9^3 264.7 -> 268.3 ms (the second LDS instruction per pair
// costs more than the bytes save there): taken up to 7^3.
#ifndef PPP_S1V3_SPLIT
#define PPP_S1V3_SPLIT(PX) ((PX) <= 7)
#endif
// iterations of the classification whose chains are interleaved
#ifndef PPP_S1V3_CGROUP
#define PPP_S1V3_CGROUP 3
#endif
typedef const volatile __attribute__((address_space(3))) float *lds_f32_cvp;
typedef const volatile __attribute__((address_space(3))) uint32_t *lds_u32_cvp;

template <int PX, bool FLAT>
struct V3 {
    static constexpr int RX = PX / 2;
    static constexpr int NC = 64 + (FLAT ? 2 : 1) * (PX - 1);   // centres per run (both segments)
    static constexpr int NT = 64 + (FLAT ? 4 : 2) * (PX - 1);   // target pixels per run
    static constexpr int NACC = 2 * PX - 1;
    static constexpr int NEL = PX * NC;                         // elements of one operand image
    static constexpr int NIT = (NEL + 63) / 64;                 // staging iterations per image
    static constexpr int NELP = NIT * 64;
    static constexpr int FOLD = 255 / PX;                       // tiles between two count folds
};

// ---- packed helpers -----------------------------------------------------------------------
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
// codes {0, 1, 256} of a pair of indicator pairs -> two 16-bit integers in one register
__device__ __forceinline__ uint32_t pack_codes(v2f pos, v2f neg) {
    const v2f c = pk_fma(neg, splat(256.0f), pos);
    return (uint32_t)c.x | ((uint32_t)c.y << 16);
}

// all votes of one (kz, ky) for both slices: kx descending (raster order of the centre), every
// partner column.  ROW0: offset row (dz, dy) == (0, 0), where only dx > 0 exists.
// The "about w" image holds 16-byte elements {t slice 0, t slice 1, codes, -}: one ds_read_b128
// per vote pair (volatile: the compiler would narrow it to a slower 12-byte read).  Votes are
// processed in groups of GS whose chains are interleaved stage by stage; the LDS reads of the
// next group are issued before the arithmetic of the current one.
template <int PX, int NC, bool ROW0, bool EXACT>
__device__ __forceinline__ void tile_votes3(const v2f *at, const v2f *cfp, lds_v4f_cvp bt, const v2f *bt2, const uint32_t *ct,
                                            const double th2, const double den,
                                            v2f (&acc)[2 * PX - 1], uint32_t (&tc)[2 * PX - 1]) {
    // fl(4/3) and fl(4/3 - fl(4/3)): y = fma(d, c43, d * c43lo) is the correctly rounded d / 0.75
    // (= the reference's float(double(d) / 0.75)) for every float d = |x| - 0.25, x in [0.25, 2^22]
    // -- 4d/3 is never closer than 1/6 ulp to a rounding boundary; checked exhaustively by
    // tests/csrc/th05_quotient.c
    const v2f kq = splat(-0.25f), c43 = splat(0x1.555556p+0f), c43lo = splat(-0x1.555556p-25f);
    constexpr int GS = PPP_S1V3_GROUP;
    constexpr int NJG = (PX + GS - 1) / GS;     // groups per kx
    constexpr int NG = PX * NJG;                // groups per tile, n -> kx = PX-1 - n / NJG
    v4f bcur[GS], bnxt[GS];
    v2f ta, ta_nxt, fa, fa_nxt, ga, na;
    uint32_t ca = 0u;
    // (experiment PPP_S1V3_ADDRREGS=1: the reads of a group through address registers of their own)
#ifndef PPP_S1V3_ADDRREGS
#define PPP_S1V3_ADDRREGS 0
#endif
    lds_v4f_cvp btg[GS];
#pragma unroll
    for (int g = 0; g < GS; ++g) {
        btg[g] = bt;
        if (PPP_S1V3_ADDRREGS) asm volatile("" : "+v"(btg[g]));
    }
    auto used = [](int kx, int j) { return j < PX && !(ROW0 && j <= kx); };
    auto load_group = [&](int n, v4f (&b)[GS], v2f &t, v2f &f) {
        const int kx = PX - 1 - n / NJG, jg = (n % NJG) * GS;
        if (jg == 0) { t = at[kx * NC - kx]; f = cfp[-kx]; }
#pragma unroll
        for (int g = 0; g < GS; ++g)
            if (used(kx, jg + g)) {
                if constexpr (PPP_S1V3_SPLIT(PX)) {
                    const v2f t2 = bt2[(jg + g) * NC - kx];
#ifdef PPP_S1_ABL_NOCODEREAD
                    b[g] = (v4f){t2.x, t2.y, __uint_as_float(0x00010001u), 0.0f};   // (timing experiment: 8 bytes per pair)
#else
                    b[g] = (v4f){t2.x, t2.y, __uint_as_float(ct[(jg + g) * NC - kx]), 0.0f};
#endif
                } else {
#ifdef PPP_S1_ABL_NOLDS
                    b[g] = (v4f){t.x * 0.9f, t.y * 0.8f, __uint_as_float(0x00010001u), 0.0f};   // (timing experiment)
#else
                    b[g] = btg[g][(jg + g) * NC - kx];
#endif
                }
            }
    };
    // PPP_S1V3_DEPTH = 2 (experiment): the LDS reads run TWO groups ahead of the arithmetic
#ifndef PPP_S1V3_DEPTH
#define PPP_S1V3_DEPTH 1
#endif
    constexpr int DEPTH = (PPP_S1V3_DEPTH == 2 && NJG >= 2) ? 2 : 1;    // (one new kx per two groups at most)
    v4f bnx2[GS];
    load_group(0, bcur, ta, fa);
    if (DEPTH == 2 && NG > 1) load_group(1, bnxt, ta_nxt, fa_nxt);
#pragma unroll
    for (int n = 0; n < NG; ++n) {
        const int kx = PX - 1 - n / NJG, jg = (n % NJG) * GS;
        if (DEPTH == 1) { if (n + 1 < NG) load_group(n + 1, bnxt, ta_nxt, fa_nxt); }
        else if (n + 2 < NG) load_group(n + 2, bnx2, ta_nxt, fa_nxt);
        if (jg == 0) {
            // centre factor {0, 1}.  EXACT (a value outside [0, 1] in the tile): select, so that an
            // infinite prediction at a centre that votes nothing contributes 0 like the reference's
            // skipped centre (fillConsensusArray.cu:25-32), not inf * 0 = nan
            if constexpr (EXACT) ta = (v2f){fa.x != 0.0f ? ta.x : 0.0f, fa.y != 0.0f ? ta.y : 0.0f};
            else ta = ta * fa;
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
        for (int g = 0; g < GS; ++g) {
            if (!used(kx, jg + g)) continue;
            tb[g] = (v2f){bcur[g].x, bcur[g].y};
            cb[g] = __float_as_uint(bcur[g].z);
            x[g] = ta * tb[g];
        }
        if constexpr (!EXACT) {
            v2f dp[GS], dn[GS], d[GS], q0[GS];
#pragma unroll
            for (int g = 0; g < GS; ++g) if (used(kx, jg + g)) dp[g] = pk_fma_clamp(x[g], ga, kq);
#pragma unroll
            for (int g = 0; g < GS; ++g) if (used(kx, jg + g)) dn[g] = pk_nadd_clamp(x[g], kq);
#pragma unroll
            for (int g = 0; g < GS; ++g) if (used(kx, jg + g)) d[g] = dp[g] - dn[g];
#pragma unroll
            for (int g = 0; g < GS; ++g) if (used(kx, jg + g)) q0[g] = d[g] * c43lo;
#pragma unroll
            for (int g = 0; g < GS; ++g) if (used(kx, jg + g)) y[g] = pk_fma(d[g], c43, q0[g]);
        } else {
#pragma unroll
            for (int g = 0; g < GS; ++g) {
                if (!used(kx, jg + g)) continue;
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
        for (int g = 0; g < GS; ++g) {
            if (!used(kx, jg + g)) continue;
            const int i = jg + g - kx + PX - 1;
            acc[i] = acc[i] + y[g];
#ifndef PPP_S1_ABL_NOCNT
            tc[i] = pk_mad_u16(ca, cb[g], tc[i]);
#endif
        }
        if (n + 1 < NG) {
#pragma unroll
            for (int g = 0; g < GS; ++g) { bcur[g] = bnxt[g]; if (DEPTH == 2) bnxt[g] = bnx2[g]; }
            if ((n + 1) % NJG == 0) { ta = ta_nxt; fa = fa_nxt; }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Prefetch loads: global loads (scalar base + 32-bit per-lane offset), NOT raw buffer loads.
// Buffer loads need no vector instruction per load (the global form spends a v_mov per load on
// the offset) and are 4 % faster on volumes whose channel offsets stay below 2 GB (9^3 slab of a
// 48 x 512 x 512 volume: 252 vs 264 ms) -- but with the 512^3 block resident (channel stride
// 268 MB, per-lane offsets up to 2.1 GB) the same launch takes 309 ms with buffer loads against
// 281 ms with global loads and 291 ms for round 4's kernel (profiles/r05_l_s1_tile.txt).
#ifndef PPP_S1V3_BUFLOAD
#define PPP_S1V3_BUFLOAD 0
#endif
// raw buffer loads: scalar base (the resource), 32-bit per-lane byte offset, scalar byte offset.
// DATA_FORMAT_32 in word 3 (what gfx9 wants of an untyped buffer), num_records = 2^32 - 1.
typedef __amdgpu_buffer_rsrc_t BufRsrc;
__device__ __forceinline__ BufRsrc buf_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, -1, 0x00020000);
}
// the loaded bits stay as they are (float16: zero-extended) until `widen`: the conversion of a
// prefetched value belongs to the classification, after the votes the load flies behind
template <typename T>
__device__ __forceinline__ unsigned buf_ldraw(BufRsrc r, unsigned voff, unsigned soff);
template <>
__device__ __forceinline__ unsigned buf_ldraw<float>(BufRsrc r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0);
}
template <>
__device__ __forceinline__ unsigned buf_ldraw<__half>(BufRsrc r, unsigned voff, unsigned soff) {
    return (unsigned)__builtin_amdgcn_raw_buffer_load_b16(r, (int)voff, (int)soff, 0);
}
// (experiment switch PPP_S1V3_BUFLOAD=0: the same bits by global loads, saddr + 32-bit voffset)
template <typename T>
__device__ __forceinline__ unsigned glb_ldraw(const T *base, unsigned voff, unsigned soff);
// (an empty asm pins a COPY of the offset per load: it keeps the zero extension next to the load --
// hoisted out of the tile loop it would turn every load into a 64-bit VGPR address computation
// instead of saddr + voffset.  One copy shared by the four loads of an element saves 39 v_mov per
// tile and is SLOWER: 312 vs 281 ms on the 512^3 tile, 263 vs 252 on the small slab
// (profiles/r05_n_s1_tile.txt) -- like the buffer loads, which share the offset register too.)
template <>
__device__ __forceinline__ unsigned glb_ldraw<float>(const float *base, unsigned voff, unsigned soff) {
    asm volatile("" : "+v"(voff));
    return *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(base) + soff + voff);
}
template <>
__device__ __forceinline__ unsigned glb_ldraw<__half>(const __half *base, unsigned voff, unsigned soff) {
    asm volatile("" : "+v"(voff));
    return (unsigned)*reinterpret_cast<const unsigned short *>(reinterpret_cast<const char *>(base) + soff + voff);
}
template <typename T>
__device__ __forceinline__ float widen(unsigned raw);
template <>
__device__ __forceinline__ float widen<float>(unsigned raw) { return __uint_as_float(raw); }
template <>
__device__ __forceinline__ float widen<__half>(unsigned raw) {
    return (float)__builtin_bit_cast(_Float16, (unsigned short)raw);
}

template <typename T>
__device__ __forceinline__ float ldf_at3(const T *base, unsigned byte_off) {
    // (keeps the zero extension of the offset next to the load: hoisted out of the tile loop it
    // would turn every load into a 64-bit VGPR address computation instead of saddr + voffset)
    asm volatile("" : "+v"(byte_off));
    return ldf(reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off), 0);
}

// CLEAN (ppp_params.pred_clean, established by ppp_pred_check): every prediction value lies in
// [0, 1] (as a bit pattern: no negative zero, no nan) and none equals the threshold 0.5.  Then
// [v > 0.5] + [v < 0.5] = 1 and the classification of an operand is t = v - [v < 0.5], its code
// 1 + 255 [v < 0.5]: two packed operations per "about u" pair and three per "about w" pair instead
// of four and five, no range check of the staged values (four integer maxima per staging iteration)
// and no exact path in the kernel -- 12 instead of 24 vector instructions per staging iteration.
template <typename T, int PX, bool FLAT, bool CLEAN>
__global__ void __launch_bounds__(64 * PPP_S1V3_WAVES(PX), PPP_S1V3_MINWAVES(PX))
    consensus_v3_kernel(const T *__restrict__ pred, const uint8_t *__restrict__ ov,
                        float *__restrict__ cons, float *__restrict__ cnt_out, const Geo G,
                        const int n_rows, const int runs_per_line, const int bZ2,
                        const long long n_waves) {
    using K = V3<PX, FLAT>;
    constexpr int NIT = K::NIT;
    constexpr int V3_WAVES = PPP_S1V3_WAVES(PX);
    constexpr bool SPLIT = PPP_S1V3_SPLIT(PX);
    __shared__ v4f lds_bt[V3_WAVES][SPLIT ? 1 : K::NELP];
    __shared__ v2f lds_bt2[V3_WAVES][SPLIT ? K::NELP : 1];
    __shared__ uint32_t lds_ct[V3_WAVES][SPLIT ? K::NELP : 1];
    __shared__ v2f lds_at[V3_WAVES][K::NELP];
    __shared__ v2f lds_cf[V3_WAVES][K::NC];
    __shared__ uint8_t lds_valid[V3_WAVES][2][2][K::NT + 2];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // XCD-aware order (as v2): each XCD gets a contiguous range of (x-run, row) work
    long long bid = blockIdx.x;
    {
        const long long nb = gridDim.x, per = nb / 8, main = per * 8;
        if (bid < main) bid = (bid % 8) * per + bid / 8;
    }
    const long long wid = bid * V3_WAVES + wave;
    if (wid >= n_waves) return;
    const int row = (int)(wid % n_rows);
    long long run = wid / n_rows;
    int dz, dy;
    if (row < G.py) { dz = 0; dy = row; }
    else { const int t = row - G.py; dz = 1 + t / G.wy; dy = t % G.wy - (G.py - 1); }
    // run -> base voxels of slice 0 (slice 1 = one slice further).  `runs_per_line` = runs per
    // line, or per z-slice when FLAT; bZ2 = pairs of slices.
    const int xr = (int)(run % runs_per_line);
    run /= runs_per_line;
    int uy, uz, ux0, nA;
    // (base voxels are enumerated over the COMPUTE box c*, a sub-box of the cons box b* that
    // indexes the output)
    if (FLAT) {
        const int flat0 = xr * 64;
        uy = G.cy0 + flat0 / G.cX;
        uz = G.cz0 + 2 * (int)run;
        ux0 = G.cx0 + flat0 % G.cX;
        nA = min(64, G.cX - flat0 % G.cX);
    } else {
        uy = G.cy0 + (int)(run % G.cY);
        uz = G.cz0 + 2 * (int)(run / G.cY);
        ux0 = G.cx0 + xr * 64;
        nA = 64;
    }
    const bool have_s1 = uz + 1 < G.cz0 + G.cZ;                    // slice 1 exists (wave-uniform)
    const bool in_b = FLAT && lane >= nA;                          // this lane sits on line B
    const bool have_b = FLAT && nA < 64 && uy + 1 < G.cy0 + G.cY;  // (wave-uniform)
    const int ux = in_b ? G.cx0 + (lane - nA) : ux0 + lane;
    const int uy_l = in_b ? uy + 1 : uy;
    const bool lane_ok = in_b ? have_b : ux < G.cx0 + G.cX;
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
    // validity (foreground && !overlap) of the target pixels on the u row and on the w row of
    // both slices; applied when the accumulators are written
    const int ntA = nA + 2 * (PX - 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        uint8_t *uval = lds_valid[wave][s][0], *wval = lds_valid[wave][s][1];
        const bool s_ok = s == 0 || have_s1;
        const bool wz_ok = s ? wz_ok1 : wz_ok0;
        for (int i = lane; i < K::NT; i += 64) {
            const bool sb = FLAT && i >= ntA;
            const int x = sb ? G.cx0 - (PX - 1) + (i - ntA) : ux0 - (PX - 1) + i;
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
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // Per-lane description of the NIT elements of an operand image this lane stages for every
    // tile (element e = it*64 + lane -> channel column j, centre i); the same for both images
    // and both slices:
    //   el_off[it] : byte offset j * V + (line) * X + clamped centre x from the tile's row base
    // and of the (up to two) centres whose factor this lane computes:
    //   cf_off[c]  : byte offset (line) * X + clamped centre x from the centre row of `mid`
    //   cf_st bit c: centre inside the x-interior and its segment exists;  cf_sb bit c: segment B
    unsigned el_off[NIT];
    const int ncA = nA + (PX - 1);                    // centres of segment A
    auto centre_x = [&](int i, bool &sb, bool &ok) -> int {
        sb = FLAT && i >= ncA;
        const int iseg = sb ? i - ncA : i;
        const int cx = (sb ? G.cx0 : ux0) - (PX - 1) + K::RX + iseg;
        ok = cx >= K::RX && cx < G.X - K::RX && (!sb || have_b);
        return min(max(cx, 0), G.X - 1) + (sb ? G.X : 0);
    };
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = it * 64 + lane;
        const int i = e % K::NC;
        const int j = min(e / K::NC, PX - 1);         // (padding elements repeat the last column)
        bool sb, ok;
        const int cxc = centre_x(i, sb, ok);
        el_off[it] = (unsigned)(((long long)j * G.V + cxc) * (long long)sizeof(T));
    }
    unsigned cf_off[2];
    unsigned cf_st = 0, cf_sb = 0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int i = min(lane + 64 * c, K::NC - 1);
        bool sb, ok;
        const int cxc = centre_x(i, sb, ok);
        cf_off[c] = (unsigned)cxc * (unsigned)sizeof(T);
        cf_st |= ((ok && lane + 64 * c < K::NC) ? 1u : 0u) << c;
        cf_sb |= (sb ? 1u : 0u) << c;
    }

    v2f *at = lds_at[wave], *cf = lds_cf[wave];
    v4f *bt = lds_bt[wave];
    v2f *bt2 = lds_bt2[wave];
    uint32_t *ct = lds_ct[wave];

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
        unsigned ra[2][NIT], rb[2][NIT], rc[2][2];             // loaded bits, [slice]
        bool lz0 = true, lz1 = false, la = true, lb = false;   // flags of the LOADED tile
        // issue the (independent, unconditional) loads of one tile into registers: buffer loads
        // -- a scalar resource per operand image (base = the tile's channel row at the centre
        // row of slice 0), the lane's 32-bit element offset, the second slice as scalar offset;
        // no vector instruction per load.  A slice whose centre slice is not interior reads the
        // other slice's rows (in bounds; its centre factors are 0).  FLAT: both lines are inside
        // the z-slice (see v2).
        auto load_tile = [&](int z, int y) {
            lz0 = z_ok0; lz1 = z_ok1; la = row_a_ok; lb = row_b_ok;
            const int cz = uz - z + G.rz;
            const long long crow0 = vox(G, z_ok0 ? cz : cz + 1, uy - y + G.ry, 0);
            // (readfirstlane: the scalar offset of a buffer load must be known to be uniform)
            const unsigned s1 = __builtin_amdgcn_readfirstlane((z_ok1 && z_ok0) ? (unsigned)(slice * (long long)sizeof(T)) : 0u);
            const long long cha = (long long)((z * G.py + y) * PX) * G.V;
            const long long chb = (long long)(((z + dz) * G.py + (y + dy)) * PX) * G.V;
            const BufRsrc qa = buf_rsrc(pred + cha + crow0), qb = buf_rsrc(pred + chb + crow0),
                          qc = buf_rsrc(mid + crow0);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
#ifdef PPP_S1_ABL_NOLOAD
                ra[0][it] = ra[1][it] = 0x3b00u + ((el_off[it] + (unsigned)(cha + crow0)) & 15u);   // (timing experiment)
                rb[0][it] = rb[1][it] = 0x2e00u + ((el_off[it] + (unsigned)(chb + crow0)) & 15u);
#else
#if PPP_S1V3_BUFLOAD
                ra[0][it] = buf_ldraw<T>(qa, el_off[it], 0u);
                ra[1][it] = buf_ldraw<T>(qa, el_off[it], s1);
                rb[0][it] = buf_ldraw<T>(qb, el_off[it], 0u);
                rb[1][it] = buf_ldraw<T>(qb, el_off[it], s1);
#else
                ra[0][it] = glb_ldraw<T>(pred + cha + crow0, el_off[it], 0u);
                ra[1][it] = glb_ldraw<T>(pred + cha + crow0, el_off[it], s1);
                rb[0][it] = glb_ldraw<T>(pred + chb + crow0, el_off[it], 0u);
                rb[1][it] = glb_ldraw<T>(pred + chb + crow0, el_off[it], s1);
#endif
#endif
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                rc[0][c] = buf_ldraw<T>(qc, cf_off[c], 0u);
                rc[1][c] = buf_ldraw<T>(qc, cf_off[c], s1);
            }
        };
        constexpr bool PREFETCH = PPP_S1V3_PREFETCH(PX);
        const v2f big26 = splat(0x1p26f), nbig26 = splat(-0x1p26f), hb = splat(0x1p25f),
                  nhb = splat(-0x1p25f);
        int n_fold = 0;
        bool have = next_tile(kz, ky);
        if (PREFETCH && have) load_tile(kz, ky);
        while (have) {
            if (!PREFETCH) load_tile(kz, ky);
            // ---- centre factors: foreground && interior (x, y, z) && segment exists; multiplied
            //      into the "about u" operand when the votes read it
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bool st = (cf_st >> c) & 1u, sb = (cf_sb >> c) & 1u;
                const bool rok = st && (sb ? lb : la);
                v2f f;
                f.x = (rok && lz0 && widen<T>(rc[0][c]) > G.th_gt) ? 1.0f : 0.0f;
                f.y = (rok && lz1 && widen<T>(rc[1][c]) > G.th_gt) ? 1.0f : 0.0f;
                if (lane + 64 * c < K::NC) cf[lane + 64 * c] = f;
            }
            // ---- classify both images of both slices into LDS
            //      t = v (v > 0.5), v - 1 (v < 0.5), 0 (v == 0.5):  g = [v > 0.5], h = [v < 0.5]
            //      from clamp01((v - 0.5) * 2^26) / clamp01((0.5 - v) * 2^26), t = v*(g + h) - h.
            //      CG iterations at a time, their chains interleaved stage by stage (a dependent
            //      packed operation right behind its producer costs a wait state)
            unsigned bigmax = 0u;
            constexpr int CG = PPP_S1V3_CGROUP;
#pragma unroll
            for (int it = 0; it < NIT; ++it)       // (the widening stays here, not next to the load)
                asm volatile("" : "+v"(ra[0][it]), "+v"(ra[1][it]), "+v"(rb[0][it]), "+v"(rb[1][it]));
            bool any_big = false;
            if constexpr (CLEAN) {
                const v2f c255 = splat(255.0f), one = splat(1.0f);
#pragma unroll
                for (int i0 = 0; i0 < NIT; i0 += CG) {
                    v2f va[CG], vb[CG], ha[CG], hbb[CG], ta[CG], tb[CG], cc[CG];
#pragma unroll
                    for (int q = 0; q < CG; ++q) if (i0 + q < NIT) {
                        va[q] = (v2f){widen<T>(ra[0][i0 + q]), widen<T>(ra[1][i0 + q])};
                        vb[q] = (v2f){widen<T>(rb[0][i0 + q]), widen<T>(rb[1][i0 + q])};
                    }
#pragma unroll
                    for (int q = 0; q < CG; ++q) if (i0 + q < NIT) ha[q] = pk_fma_clamp(va[q], nbig26, hb);
#pragma unroll
                    for (int q = 0; q < CG; ++q) if (i0 + q < NIT) hbb[q] = pk_fma_clamp(vb[q], nbig26, hb);
#pragma unroll
                    for (int q = 0; q < CG; ++q) if (i0 + q < NIT) ta[q] = va[q] - ha[q];
#pragma unroll
                    for (int q = 0; q < CG; ++q) if (i0 + q < NIT) tb[q] = vb[q] - hbb[q];
#pragma unroll
                    for (int q = 0; q < CG; ++q) if (i0 + q < NIT) cc[q] = pk_fma(hbb[q], c255, one);
#pragma unroll
                    for (int q = 0; q < CG; ++q) if (i0 + q < NIT) {
                        const int e = (i0 + q) * 64 + lane;
                        const uint32_t code = (uint32_t)cc[q].x | ((uint32_t)cc[q].y << 16);
                        at[e] = ta[q];
                        if constexpr (SPLIT) { bt2[e] = tb[q]; ct[e] = code; }
                        else bt[e] = (v4f){tb[q].x, tb[q].y, __uint_as_float(code), 0.0f};
                    }
                }
            } else {
#pragma unroll
            for (int i0 = 0; i0 < NIT; i0 += CG) {
                v2f va[CG], vb[CG], ga[CG], ha[CG], gb[CG], hbb[CG], sa[CG], sbb[CG], ta[CG], tb[CG], cc[CG];
#pragma unroll
                for (int q = 0; q < CG; ++q) if (i0 + q < NIT) {
                    va[q] = (v2f){widen<T>(ra[0][i0 + q]), widen<T>(ra[1][i0 + q])};
                    vb[q] = (v2f){widen<T>(rb[0][i0 + q]), widen<T>(rb[1][i0 + q])};
                }
#pragma unroll
                for (int q = 0; q < CG; ++q) if (i0 + q < NIT) ga[q] = pk_fma_clamp(va[q], big26, nhb);
#pragma unroll
                for (int q = 0; q < CG; ++q) if (i0 + q < NIT) ha[q] = pk_fma_clamp(va[q], nbig26, hb);
#pragma unroll
                for (int q = 0; q < CG; ++q) if (i0 + q < NIT) gb[q] = pk_fma_clamp(vb[q], big26, nhb);
#pragma unroll
                for (int q = 0; q < CG; ++q) if (i0 + q < NIT) hbb[q] = pk_fma_clamp(vb[q], nbig26, hb);
#pragma unroll
                for (int q = 0; q < CG; ++q) if (i0 + q < NIT) sa[q] = ga[q] + ha[q];
#pragma unroll
                for (int q = 0; q < CG; ++q) if (i0 + q < NIT) sbb[q] = gb[q] + hbb[q];
#pragma unroll
                for (int q = 0; q < CG; ++q) if (i0 + q < NIT) cc[q] = pk_fma(hbb[q], splat(256.0f), gb[q]);
#pragma unroll
                for (int q = 0; q < CG; ++q) if (i0 + q < NIT) ta[q] = pk_fma(va[q], sa[q], -ha[q]);
#pragma unroll
                for (int q = 0; q < CG; ++q) if (i0 + q < NIT) tb[q] = pk_fma(vb[q], sbb[q], -hbb[q]);
#pragma unroll
                for (int q = 0; q < CG; ++q) if (i0 + q < NIT) {
                    const int e = (i0 + q) * 64 + lane;
                    bigmax = max(bigmax, max(max(__float_as_uint(va[q].x), __float_as_uint(va[q].y)),
                                             max(__float_as_uint(vb[q].x), __float_as_uint(vb[q].y))));
                    const uint32_t code = (uint32_t)cc[q].x | ((uint32_t)cc[q].y << 16);
                    at[e] = ta[q];
                    if constexpr (SPLIT) { bt2[e] = tb[q]; ct[e] = code; }
                    else bt[e] = (v4f){tb[q].x, tb[q].y, __uint_as_float(code), 0.0f};
                }
            }
            // a value outside [0, 1] (as an unsigned bit pattern: > 1.0f, negative, inf, nan) sends
            // the tile down the exact path: classification by compares (nan -> unclassified, no
            // 0 * inf), votes with compares and the double division
            any_big = __ballot(bigmax > 0x3F800000u) != 0ull;
            if (any_big) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int e = it * 64 + lane;
                    v2f ta, tb;
                    unsigned code = 0u;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const float va = widen<T>(ra[s][it]), vb = widen<T>(rb[s][it]);
                        const float xa = va > G.th_gt ? va : (va < G.bg_lt ? va - 1.0f : 0.0f);
                        const float xb = vb > G.th_gt ? vb : (vb < G.bg_lt ? vb - 1.0f : 0.0f);
                        if (s) { ta.y = xa; tb.y = xb; } else { ta.x = xa; tb.x = xb; }
                        code |= (xb > 0.0f ? 1u : (xb < 0.0f ? 256u : 0u)) << (16 * s);
                    }
                    at[e] = ta;
                    if constexpr (SPLIT) { bt2[e] = tb; ct[e] = code; }
                    else bt[e] = (v4f){tb.x, tb.y, __uint_as_float(code), 0.0f};
                }
            }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- prefetch the next tile; its latency hides behind this tile's votes
            have = next_tile(kz, ky);
            if (PREFETCH && have) load_tile(kz, ky);
            // ---- votes
            const v2f *ia = at + pos_l;
            lds_v4f_cvp ib = (lds_v4f_cvp)(bt + (SPLIT ? 0 : pos_l));
            const v2f *ib2 = bt2 + (SPLIT ? pos_l : 0);
            const uint32_t *ic = ct + (SPLIT ? pos_l : 0);
            const v2f *icf = cf + pos_l;
            if (CLEAN || !any_big) {
                if (row0) tile_votes3<PX, K::NC, true, false>(ia, icf, ib, ib2, ic, G.th2, G.den, acc, tc);
                else tile_votes3<PX, K::NC, false, false>(ia, icf, ib, ib2, ic, G.th2, G.den, acc, tc);
            } else if constexpr (!CLEAN) {
                if (row0) tile_votes3<PX, K::NC, true, true>(ia, icf, ib, ib2, ic, G.th2, G.den, acc, tc);
                else tile_votes3<PX, K::NC, false, true>(ia, icf, ib, ib2, ic, G.th2, G.den, acc, tc);
            }
            if (++n_fold == K::FOLD) {
                n_fold = 0;
#pragma unroll
                for (int i = 0; i < K::NACC; ++i) {
                    cnt[i] += (tc[i] & 0x00FF00FFu) + ((tc[i] >> 8) & 0x00FF00FFu);
                    tc[i] = 0u;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (!lane_ok) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int ti_u = (in_b ? ntA + (lane - nA) : lane) + PX - 1;   // index of u in the validity rows
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (s == 1 && !have_s1) break;
        const uint8_t *uval = lds_valid[wave][s][0], *wval = lds_valid[wave][s][1];
        const bool u_ok = uval[ti_u] != 0;
        float val[K::NACC];
#pragma unroll
        for (int i = 0; i < K::NACC; ++i) {
            const int dx = i - (PX - 1);
            val[i] = 0.0f;
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
            // symmetric voxel-major rows written directly (no compact planes, no transpose):
            // S[u][Lc + L(d)] -- the 2 PX - 1 entries of this offset row are contiguous, stored
            // four at a time -- and, when w = u + d lies in the box, the mirror S[w][Lc - L(d)].
            // Entries whose source voxel lies outside the box are zeroed by vm_zero_kernel.
            const int W = (2 * G.pz - 1) * G.wy * G.wx, Lc = (W - 1) / 2;
            const int L0 = (dz * G.wy + dy) * G.wx;
            const long long vu = ((long long)row_slice(G, uz + s) * G.bY + (uy_l - G.by0)) * G.bX + (ux - G.bx0);
            float *pos = cons + vu * W + Lc + L0 - (PX - 1);     // entry of dx = -(PX-1)
            if (row0) {
                pos[PX - 1] = 0.0f;                              // offset 0
#pragma unroll
                for (int i = PX; i < K::NACC; ++i) pos[i] = val[i];
            } else {
                typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
                for (int i = 0; i + 4 <= K::NACC; i += 4)
                    *reinterpret_cast<f4u *>(pos + i) = (f4u){val[i], val[i + 1], val[i + 2], val[i + 3]};
#pragma unroll
                for (int i = K::NACC / 4 * 4; i < K::NACC; ++i) pos[i] = val[i];
            }
            const int wz = uz + s + dz, wy2 = uy_l + dy;
            if (wz < G.bz0 + G.bZ && wy2 >= G.by0 && wy2 < G.by0 + G.bY) {
                // (the slice of w: a ring may wrap between u and w)
                const long long vw0 = ((long long)row_slice(G, wz) * G.bY + (wy2 - G.by0)) * G.bX + (ux - G.bx0);
#pragma unroll
                for (int i = 0; i < K::NACC; ++i) {
                    const int dx = i - (PX - 1);
                    if (row0 && dx <= 0) continue;
                    const int wx2 = ux + dx;
                    if (wx2 >= G.bx0 && wx2 < G.bx0 + G.bX) cons[(vw0 + dx) * W + Lc - L0 - dx] = val[i];
                }
            }
        }
    }
}

// Voxel-major output: the mirrored entries S[w][Lc - L(d)] whose source voxel w - d lies outside
// the consensus box have no wave that writes them; they are zero.  Only voxels within p - 1 of
// the low-z, low / high-y and low / high-x faces have such entries: thread per (face voxel,
// offset row (dz, dy)); the five face slabs may overlap (the same zero is then stored twice).
// (L = Lc, offset 0, is written by the waves of offset row 0.)
struct VmFaces { long long start[6]; int lo[5][3], ext[5][3]; };
__global__ void __launch_bounds__(256)
    vm_zero_kernel(float *__restrict__ S, const Geo G, const int n_rows, const VmFaces F) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= F.start[5] * n_rows) return;
    const int row = (int)(t % n_rows);
    const long long fv = t / n_rows;
    int f = 0;
#pragma unroll
    for (int k = 1; k < 5; ++k) f += fv >= F.start[k] ? 1 : 0;
    const long long r = fv - F.start[f];
    const int ex = F.ext[f][2], ey = F.ext[f][1];
    const int bx = F.lo[f][2] + (int)(r % ex), by = F.lo[f][1] + (int)((r / ex) % ey),
              bz = F.lo[f][0] + (int)(r / ((long long)ex * ey));
    int dz, dy;
    if (row < G.py) { dz = 0; dy = row; }
    else { const int q = row - G.py; dz = 1 + q / G.wy; dy = q % G.wy - (G.py - 1); }
    const int W = (2 * G.pz - 1) * G.wy * G.wx, Lc = (W - 1) / 2;
    float *Sv = S + (((long long)bz * G.bY + by) * G.bX + bx) * W;
    const bool line_out = bz - dz < 0 || by - dy < 0 || by - dy >= G.bY;
    const int L0 = (dz * G.wy + dy) * G.wx;
    for (int dx = -(G.px - 1); dx <= G.px - 1; ++dx) {
        if (row == 0 && dx <= 0) continue;
        const int sx = bx - dx;
        if (line_out || sx < 0 || sx >= G.bX) Sv[Lc - L0 - dx] = 0.0f;
    }
}

static hipError_t launch_vm_zero(float *S, const Geo &G, int n_rows, hipStream_t s) {
    VmFaces F{};
    const int gz = std::min(G.pz - 1, G.bZ), gy = std::min(G.py - 1, G.bY), gx = std::min(G.px - 1, G.bX);
    const int lo[5][3] = {{0, 0, 0}, {0, 0, 0}, {0, G.bY - gy, 0}, {0, 0, 0}, {0, 0, G.bX - gx}};
    const int ext[5][3] = {{gz, G.bY, G.bX}, {G.bZ, gy, G.bX}, {G.bZ, gy, G.bX}, {G.bZ, G.bY, gx}, {G.bZ, G.bY, gx}};
    long long acc = 0;
    for (int f = 0; f < 5; ++f) {
        F.start[f] = acc;
        for (int a = 0; a < 3; ++a) { F.lo[f][a] = lo[f][a]; F.ext[f][a] = std::max(ext[f][a], 1); }
        acc += (long long)ext[f][0] * ext[f][1] * ext[f][2];
        // (a face of extent 0 -- patch size 1 on that axis -- contributes no voxels)
    }
    F.start[5] = acc;
    const long long nt = acc * n_rows;
    if (nt == 0) return hipSuccess;
    PPP_GRID_CHECK((nt + 255) / 256, 256);
    vm_zero_kernel<<<dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s>>>(S, G, n_rows, F);
    return hipGetLastError();
}

// (also the two-wave kernel's, ppp_consensus_v4.hip)
hipError_t launch_vm_zero_faces(float *S, const Geo &G, hipStream_t s) {
    return launch_vm_zero(S, G, (G.pz - 1) * G.wy + G.py, s);
}

template <typename T, int PX, bool FLAT, bool CLEAN>
static hipError_t launch_v3f(const T *pred, const uint8_t *ov, float *cons, float *cnt,
                             const Geo &G, hipStream_t s) {
    const int n_rows = (G.pz - 1) * G.wy + G.py;
    const int runs_per_line = FLAT ? (int)(((long long)G.cX * G.cY + 63) / 64) : (G.cX + 63) / 64;
    const int bZ2 = (G.cZ + 1) / 2;
    const long long n_waves = (long long)runs_per_line * (FLAT ? 1 : G.cY) * bZ2 * n_rows;
    constexpr int V3_WAVES = PPP_S1V3_WAVES(PX);
    const long long n_blocks = (n_waves + V3_WAVES - 1) / V3_WAVES;
    if (n_blocks >= (1ll << 31)) return hipErrorInvalidValue;
    PPP_GRID_CHECK(n_blocks, 64 * V3_WAVES);
    if (((long long)(PX - 1) * G.V + 2ll * G.X) * (long long)sizeof(T) >= (1ll << 32)) return hipErrorNotSupported;
    if (G.layout == PPP_CONS_VOXEL_MAJOR && !G.vm_open) {
        const hipError_t ez = launch_vm_zero(cons, G, n_rows, s);
        if (ez != hipSuccess) return ez;
    }
    // (experiment PPP_S1_DYNLDS=<bytes>: dynamic LDS nobody uses, to LOWER the waves a CU holds -- what a
    // wave more or less per SIMD is worth: profiles/r06_d_s1_occupancy.txt)
    static EnvSwitch dyn("PPP_S1_DYNLDS");
    const char *ed = dyn.get();
    const unsigned dyn_lds = ed ? (unsigned)atoi(ed) : 0u;
    if (dyn_lds) {
        const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&consensus_v3_kernel<T, PX, FLAT, CLEAN>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds);
        if (ea != hipSuccess) return ea;
    }
    consensus_v3_kernel<T, PX, FLAT, CLEAN><<<dim3((unsigned)n_blocks), dim3(64 * V3_WAVES), dyn_lds, s>>>(
        pred, ov, cons, cnt, G, n_rows, runs_per_line, bZ2, n_waves);
    return hipGetLastError();
}

template <typename T, int PX>
static hipError_t launch_v3(const T *pred, const uint8_t *ov, float *cons, float *cnt,
                            const Geo &G, hipStream_t s) {
    static EnvSwitch sw("PPP_S1_FLAT");        // same rule as v2
    const char *e = sw.get();
    bool flat = G.cX >= 64 && G.cX % 64 != 0 && G.py >= 3 && G.cY > 1;
    if (e && e[0] == '0') flat = false;
    if (e && e[0] == '1' && G.cX >= 64 && G.py >= 3) flat = true;
    static EnvSwitch swc("PPP_S1_CLEAN");      // PPP_S1_CLEAN=0: the general kernel whatever the caller knows
    const char *ec = swc.get();
    const bool clean = G.pred_clean == 1 && !(ec && ec[0] == '0');
    if (clean)
        return flat ? launch_v3f<T, PX, true, true>(pred, ov, cons, cnt, G, s)
                    : launch_v3f<T, PX, false, true>(pred, ov, cons, cnt, G, s);
    return flat ? launch_v3f<T, PX, true, false>(pred, ov, cons, cnt, G, s)
                : launch_v3f<T, PX, false, false>(pred, ov, cons, cnt, G, s);
}

// the shapes / rules the packed kernel serves (and with them the direct voxel-major output)
bool consensus_v3_supported(const Geo &G) {
    static EnvSwitch sw("PPP_S1_V3");
    const char *e = sw.get();
    if (e && e[0] == '0') return false;
    return G.value_rule == PPP_VAL_NORM_PROB_PRODUCT && G.th2 == 0.25 && G.den == 0.75 &&
           G.th_gt == 0.5f && G.bg_lt == 0.5f && (G.px == 3 || G.px == 5 || G.px == 7 || G.px == 9);
}

// TH = 0.5, normalised probability product, px in {3,5,7,9}; hipErrorNotSupported otherwise
// (the caller falls back to v2).  PPP_S1_V3=0 switches it off.
hipError_t launch_consensus_v3(const void *pred, int dtype, const uint8_t *ov, float *cons,
                               float *cnt, const Geo &G, hipStream_t s) {
    if (!consensus_v3_supported(G)) return hipErrorNotSupported;
    if (G.layout == PPP_CONS_VOXEL_MAJOR && (cnt || !cons)) return hipErrorInvalidValue;
#define PPP_V3_CASE(P)                                                                          \
    case P:                                                                                     \
        return dtype == PPP_F16                                                                 \
                   ? launch_v3<__half, P>((const __half *)pred, ov, cons, cnt, G, s)            \
                   : launch_v3<float, P>((const float *)pred, ov, cons, cnt, G, s);
    switch (G.px) {
        PPP_V3_CASE(3)
        PPP_V3_CASE(5)
        PPP_V3_CASE(7)
        PPP_V3_CASE(9)
    default:
        return hipErrorNotSupported;
    }
#undef PPP_V3_CASE
}

}  // namespace ppp
