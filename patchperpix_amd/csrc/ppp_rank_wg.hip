// ppp_rank_wg.hip -- S2 (patch ranking) on the VOXEL-MAJOR consensus, row-stationary, one
// WORKGROUP (four waves) per tile of centres.
//
// Reference: cuda/rankPatches.cu:28-147; same sums in the same order as ppp_rank_vm.hip (whose
// header states the decomposition: walk the voxels u of the tile grown by the patch radius in
// raster order, stage the consensus row S[u][.] in LDS, every centre of the tile whose window
// holds u takes its step a = u - c + rad; lane = (centre, a); accumulators live in LDS between
// the steps of a centre).  What differs from the one-wave kernel:
//
//   * four waves share ONE row image, ONE coefficient table and a tile of 8 x 16 x 16 (or
//     8 x 8 x 16) centres: the one-wave kernel needs 26 KB of LDS per wave at 9^3 (6 waves per
//     CU, 1.5 per SIMD -- an LDS read then runs at a fraction of its rate and nothing hides
//     the dependent fma chain); here a workgroup needs 30 KB for four waves (16 waves per CU at
//     <= 128 VGPRs).  The items of a row are dealt to the waves in chunks of 64, the wave that
//     takes the odd chunk rotates from row to row.
//   * a row is staged (tz + 2 rz)(ty + 2 ry)(tx + 2 rx) / (tz ty tx) = 4.5 times per centre at
//     9^3 instead of 8 times (HBM fetches of the launch: 0.45 of the one-wave kernel's).
//   * a term is ONE instruction: v_fma_mix_f32 takes its coefficient +-1/32 / 0 as a float16 from
//     either half of a register (no v_perm_b32 to build a float32 coefficient).  fma(32 S, c, acc)
//     with c = +-2^-5 has an exact product and rounds once, like the reference's acc += / -= S.
//   * the P / N masks of a centre are stored interleaved by the pre-pass -- byte j of word k holds
//     the P bits (low nibble) and the N bits (high nibble) of the four partners 16 k + 4 j .. + 3 --
//     so that ONE byte indexes ONE 256-entry table of four float16 coefficients (2 KB; 8-byte
//     reads): 0.25 table reads and 0.25 address operations per term instead of 0.25 + 0.6.
//
// Measured and dropped on a 128^3 / 9^3 launch (this kernel: 204 ms; profiles/r03_e_pmc_s2_wg_*,
// r03_f_s2_variants_timing.txt; all bit-identical):
//   * a two-line row image of 8-byte elements {S[q], S[q + one line]} read by ds_read_b64 (256
//     instead of 128 B per LDS cycle, 12 instead of 16 waves per CU): 227 ms -- fewer LDS cycles
//     (4.6 vs 6.7 * 10^10) but the LDS is only half busy in either form;
//   * a slot-stationary form (a centre always meets the same lane, masks and accumulator stay in
//     registers for the p rows of an x-run: a ninth of the mask loads, which are 60 % of this
//     kernel's 482 GB of fetches): 402 ms -- one 12-wave workgroup per CU does one chunk per
//     barrier interval and nothing overlaps the row publication.
// Round 5, designed and not built: the masks are 60 % of the fetches and, by the
// constant-mask ablation, 39 % of the time, and every chunk waits for its twelve mask loads before
// its chain can start.  A walk along x-lines with lanes bound to (centre line, slot = cx mod 9)
// keeps a centre's masks and accumulator in registers for its nine rows (a ninth of the loads);
// with four waves a line's <= 64 centre lines need ~2.3 groups, i.e. the rows of the line staged
// 2.3 x (from L2) -- the bytes the masks save -- so what it can win is the LATENCY: only if the
// masks of the ONE centre per line that enters at the next row are prefetched (28 x 192 B per row
// and group) by a fifth, helper wave into a 10 KB LDS double buffer while the four compute waves
// run the current row.  Rows that are skipped (no valid voxel) make several centres enter at once:
// those lanes load directly, as every lane does today.
// What holds the kernel back is neither unit alone (VALU 24 %, LDS 54 % busy of which half bank
// conflicts, 2.35 TB/s of fetches) but the stalls of its dependent chains at 16 waves per CU.
//
// Round 5, measured (profiles/r05_z*; DESIGN.md section 7 has the numbers):
//   * kept: the four reads of a group are issued last element first (one s_waitcnt per group); the
//     two-bit masks lie in rows of 16 x-neighbours (a quad load of an x-run touches 144 contiguous
//     bytes); the tiles of a launch are dealt HEAVIEST FIRST inside every XCD's range (a launch ends
//     with its slowest workgroup: one round of 1 024 workgroups 146 ms, every further round 64 ms);
//   * dropped, all bit-identical: TWO x-neighbouring centres per lane sharing their row reads
//     (0.58 reads per term, two independent chains: 476 vs 253 ms -- five chunks of double work per
//     row for four waves, and the second mask set costs a wave per SIMD); three waves per workgroup;
//     ds_read2_b32 through a non-volatile pointer; 8 x 8 x 8 tiles; staggered starts of the first round;
//   * the instruction cache serves 100 % hits and s_waitcnt / s_nop cost next to nothing
//     (tools/ubench/issue_mix.hip); v_fma_mix_f32 issues in 4.6-5.0 cycles per SIMD against 2.9 for
//     v_fma_f32; a conflict-free ds_read_b32 costs the CU 2.6 cycles; with 1 / 2 / 3 / 4 resident
//     workgroups per CU the 112 x 176 x 176 launch takes 579 / 348 / 279 / 248 ms (a fifth would
//     give ~6 %); without masks, table and row reads at all it takes 180 ms.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "ppp_kernels.hpp"

namespace ppp {

// waves per SIMD the register budget must allow (experiments: -DPPP_RW_MINWAVES(PX)=n).  More
// resident waves do not help: 9^3 at 5 per SIMD (96 VGPRs, 14 spilled) 205 -> 247 ms on the 128^3
// launch, 7^3 at 6 per SIMD (75 VGPRs, nothing spilled) 72.5 -> 73.1 ms at 140^3.
#ifndef PPP_RW_MINWAVES
#define PPP_RW_MINWAVES(PX) 4
#endif
// LDS bank conflicts of the row reads.  A half-wave is 32 consecutive items; an item reads the row
// image at -(plane * az + 17 * ay + ax) + const.  Enumerated (ax, ay, az) -- x-runs of nine lanes,
// consecutive runs 17 floats apart -- the runs of ay and ay + 2 share seven banks: every read
// took two passes (half of the LDS's active cycles were conflicts, r03/r04 profiles).  Enumerated
// (ax, az, ay) with the image's PLANE stride padded from 289 to 297 (= 9 mod 32), consecutive
// runs sit 9 banks apart -- 0-8, 9-17, 18-26, 27-31 -- and conflicts remain only where a
// half-wave straddles the step from the last az of one ay to the first of the next.
// (-DPPP_RW_ZRUNS=0: the old enumeration and image.)
// Measured (tools/time_s2.py, profiles/r04_j_s2_zruns.txt): 9^3 190 -> 166 ms at 128^3, 102 -> 84 ms
// at 96^3; 7^3 (plane stride 169 = 9 mod 32 already, conflicts mild before) 62.2 -> 64.7 ms at
// 140^3 -- so only 9^3 takes it.  With all nine az present (rows in the z-interior of a tile at
// least 9 thick) the step to the next ay continues the lattice (81 = 17 mod 32 = the y stride):
// the 16 x 8 x 16 tile has half of its items in such rows.
#ifndef PPP_RW_ZRUNS
#define PPP_RW_ZRUNS(PX) ((PX) == 9)
#endif
static constexpr int RW_PAD = 8;
#ifndef PPP_RW_WAVES
#define PPP_RW_WAVES 4
#endif
#ifndef PPP_RW_REVREADS
#define PPP_RW_REVREADS 1
#endif
static constexpr int RW_WAVES = PPP_RW_WAVES;     // waves per workgroup
// Mask words per centre, padded to whole 16-byte loads.  The masks are stored CENTRE-MAJOR,
// M[centre][word]: a lane reads the 184 bytes of ITS centre with twelve 16-byte loads, and the
// centres of a wave's chunk (9-runs of x neighbours) cover their cache lines densely.  The earlier
// word-major layout M[word][centre] made each of the 46 loads of a chunk touch eight 128-byte lines
// for 36 bytes apiece: 47 KB of line traffic per chunk for 11.8 KB of masks -- and because a
// workgroup's masks (377 KB) are re-read on every one of the 729 steps while 128 workgroups per
// XCD share a 4 MB L2, most of those lines came over the fabric (the kernel ran at the fabric's
// bandwidth: profiles/r04_b_s2_ablations.txt -- 205 ms, 126 ms without the mask loads).
static constexpr int rw_mask_words(int C) { return (((C + 15) / 16) + 3) & ~3; }
// Round 5: the two-bit masks in ROWS OF 16 X-NEIGHBOURS, M[row][quad of words][centre in row][4 words]:
// the 16-byte load of quad q by the nine x-neighbours of a run touches 144 contiguous bytes (two or
// three lines) instead of nine lines 184 bytes apart -- the bytes fetched stay the same, the
// addresses the vector cache has to look up per load fall to a quarter.  PPP_RW_MASKBLK=0: centre-major.
#ifndef PPP_RW_MASKBLK
#define PPP_RW_MASKBLK 1
#endif
// index (in 16-byte units) of quad 0 of the centre at x of line zy (score-box coordinates)
__host__ __device__ __forceinline__ long long rw_mask_quad0(long long zy, int x, int sX, int quads) {
    if (!PPP_RW_MASKBLK) return (zy * sX + x) * quads;
    return ((zy * ((sX + 15) >> 4) + (x >> 4)) * quads) * 16 + (x & 15);
}
static constexpr int RW_QSTRIDE = PPP_RW_MASKBLK ? 16 : 1;      // 16-byte units between a centre's quads
static size_t rw_mask_bytes(int sZ, int sY, int sX, int C) {
    const size_t cols = PPP_RW_MASKBLK ? (size_t)((sX + 15) >> 4) * 16 : (size_t)sX;
    return (size_t)sZ * sY * cols * rw_mask_words(C) * 4;
}
#ifndef PPP_RW_NOVOLATILE
typedef const volatile __attribute__((address_space(3))) float *lds_f32_cvp2;
#else
typedef const __attribute__((address_space(3))) float *lds_f32_cvp2;
#endif

// acc + r * c with c a float16 in the low / high half of a register: the compiler selects
// v_fma_mix_f32 (op_sel picks the half; the float16 -> float32 conversion is exact and part of the
// fused operation, which rounds once)
typedef _Float16 rw_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fma_mix_lo(float r, uint32_t c, float acc) {
    return __builtin_fmaf(r, (float)__builtin_bit_cast(rw_h2, c).x, acc);
}
__device__ __forceinline__ float fma_mix_hi(float r, uint32_t c, float acc) {
    return __builtin_fmaf(r, (float)__builtin_bit_cast(rw_h2, c).y, acc);
}

template <int I, int END>
struct StaticFor {
    template <class F>
    static __device__ __forceinline__ void run(F &&f) {
        f(std::integral_constant<int, I>{});
        StaticFor<I + 1, END>::run(f);
    }
};
template <int END>
struct StaticFor<END, END> {
    template <class F>
    static __device__ __forceinline__ void run(F &&) {}
};

// 16 P bits and 16 N bits -> one word: byte j = P nibble j | N nibble j << 4
__host__ __device__ __forceinline__ uint32_t interleave16(uint32_t p, uint32_t n) {
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        out |= (((p >> (4 * j)) & 0xFu) | (((n >> (4 * j)) & 0xFu) << 4)) << (8 * j);
    return out;
}

// ---- pre-pass: interleaved masks, pair counts, border / background scores ------------------
// thread per centre of the score box; masks word-major over the box (lanes = neighbouring
// centres coalesce in the main kernel)
template <typename T>
__global__ void __launch_bounds__(256)
    rank_masks_il_kernel(const T *__restrict__ pred, const uint8_t *__restrict__ ov, const ppp_box sb,
                         uint32_t *__restrict__ M, uint32_t *__restrict__ info,
                         float *__restrict__ score, const int *__restrict__ any_e, const Geo G) {
    if (any_e && *any_e == 0) return;         // (the one-bit masks serve this launch)
    const int sX = sb.x1 - sb.x0, sY = sb.y1 - sb.y0, sZ = sb.z1 - sb.z0;
    const long long sbV = (long long)sX * sY * sZ;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= sbV) return;
    const int cx = sb.x0 + (int)(t % sX), cy = sb.y0 + (int)((t / sX) % sY), cz = sb.z0 + (int)(t / ((long long)sX * sY));
    const long long lc = vox(G, cz, cy, cx);
    const T *mid = pred + (long long)G.mid * G.V;
    uint32_t inf = 0;
    if (!interior(G, cz, cy, cx)) {
        score[lc] = G.norm_rank ? -1.0f : -9999999.0f;
    } else if (!(ldf(mid, lc) > G.th_gt)) {
        score[lc] = 0.0f;   // the reference leaves the allocation's zero
    } else {
        const int words16 = (G.C + 15) / 16;
        const long long q0 = rw_mask_quad0(t / sX, (int)(t % sX), sX, rw_mask_words(G.C) / 4);
        unsigned nP = 0, nV = 0;
        int r = 0;
        for (int w = 0; w < words16; ++w) {
            uint32_t p = 0, n = 0;
            for (int b = 0; b < 16 && r < G.C; ++b, ++r) {
                const int z = cz + r / (G.py * G.px) - G.rz, y = cy + (r / G.px) % G.py - G.ry,
                          x = cx + r % G.px - G.rx;
                const long long lz = vox(G, z, y, x);
                const bool valid = ldf(mid, lz) > G.th_gt && (!G.use_overlap || ov[lz] == 0);
                const float val = ldf(pred, (long long)r * G.V + lc);
                if (valid) ++nV;
                if (valid && val > G.th_gt) p |= 1u << b;
                if (valid && val < G.bg_lt) n |= 1u << b;
            }
            M[(q0 + (long long)(w >> 2) * RW_QSTRIDE) * 4 + (w & 3)] = interleave16(p, n);
            nP += __popc(p);
        }
        for (int w = words16; w < rw_mask_words(G.C); ++w) M[(q0 + (long long)(w >> 2) * RW_QSTRIDE) * 4 + (w & 3)] = 0u;
        // fgCnt = |P| (|V| - 1) - |P| (|P| - 1) / 2   (rankPatches.cu:139, see ppp_rank_v2.hip)
        const unsigned fg_cnt = nP ? nP * (nV - 1u) - nP * (nP - 1u) / 2u : 0u;
        inf = 0x80000000u | fg_cnt;
        if (nP == 0) score[lc] = 0.0f;   // no first pixel: acc = 0, 0 / max(1, 0)
    }
    info[t] = inf;
}

// ---- one bit per partner (round 5) ----------------------------------------------------------
// The consensus entry S[u][q] is 0 whenever u or u + q is not a valid foreground voxel (the votes
// skip such pixels: fillConsensusArray.cu:45-57, 75-83), so a term with an invalid partner adds
// +-0 to the running sum -- which never changes it (the sum is never -0).  The masks therefore need
// no validity, and with the background rule "v < TH" (the shipped one) a partner that is not in P
// is in N unless its value EQUALS the threshold: ONE bit per partner (P' = v > TH), 92 bytes per
// centre instead of 184 -- half the mask fetches, half the mask registers.  A launch in which any
// centre has a valid partner with v == TH (or a NaN) falls back to the two-bit masks: the pre-pass
// raises `any_e`, the one-bit main kernel returns at once and the two-bit pre-pass + kernel, launched
// behind it, run (they return at once otherwise).
static constexpr int rw_p1_words(int C) { return (((C + 31) / 32) + 3) & ~3; }
template <typename T>
__global__ void __launch_bounds__(256)
    rank_masks_p1_kernel(const T *__restrict__ pred, const uint8_t *__restrict__ ov, const ppp_box sb,
                         uint32_t *__restrict__ M, uint32_t *__restrict__ info, int *__restrict__ any_e,
                         float *__restrict__ score, const Geo G) {
    const int sX = sb.x1 - sb.x0, sY = sb.y1 - sb.y0, sZ = sb.z1 - sb.z0;
    const long long sbV = (long long)sX * sY * sZ;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= sbV) return;
    const int cx = sb.x0 + (int)(t % sX), cy = sb.y0 + (int)((t / sX) % sY), cz = sb.z0 + (int)(t / ((long long)sX * sY));
    const long long lc = vox(G, cz, cy, cx);
    const T *mid = pred + (long long)G.mid * G.V;
    uint32_t inf = 0;
    if (!interior(G, cz, cy, cx)) {
        score[lc] = G.norm_rank ? -1.0f : -9999999.0f;
    } else if (!(ldf(mid, lc) > G.th_gt)) {
        score[lc] = 0.0f;   // the reference leaves the allocation's zero
    } else {
        const int words = (G.C + 31) / 32, wp = rw_p1_words(G.C);
        unsigned nP = 0, nV = 0;
        bool e = false;
        int r = 0;
        for (int w = 0; w < words; ++w) {
            uint32_t p = 0;
            for (int b = 0; b < 32 && r < G.C; ++b, ++r) {
                const int z = cz + r / (G.py * G.px) - G.rz, y = cy + (r / G.px) % G.py - G.ry,
                          x = cx + r % G.px - G.rx;
                const long long lz = vox(G, z, y, x);
                const bool valid = ldf(mid, lz) > G.th_gt && (!G.use_overlap || ov[lz] == 0);
                const float val = ldf(pred, (long long)r * G.V + lc);
                const bool isp = val > G.th_gt;
                if (valid) ++nV;
                if (valid && isp) ++nP;
                if (valid && !isp && !(val < G.bg_lt)) e = true;     // neither P nor N: needs the two-bit masks
                if (isp) p |= 1u << b;
            }
            M[t * (long long)wp + w] = p;
        }
        for (int w = words; w < wp; ++w) M[t * (long long)wp + w] = 0u;
        if (e) *any_e = 1;
        const unsigned fg_cnt = nP ? nP * (nV - 1u) - nP * (nP - 1u) / 2u : 0u;
        inf = 0x80000000u | fg_cnt;
        if (nP == 0) score[lc] = 0.0f;
    }
    info[t] = inf;
}

template <typename T>
__global__ void __launch_bounds__(256)
    rank_valid2_kernel(const T *__restrict__ pred, const uint8_t *__restrict__ ov, uint8_t *__restrict__ valid,
                       const Geo G) {
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= G.V) return;
    valid[v] = (ldf(pred, (long long)G.mid * G.V + v) > G.th_gt && (!G.use_overlap || ov[v] == 0)) ? 1 : 0;
}

// ---- heavy tiles first (round 5) -----------------------------------------------------------
// A launch ends with its slowest workgroup, and the tiles of centres differ: by the valid voxels
// of their grown box (rows walked) and by their foreground centres (items with work).  On a 264^2
// column one round of 1 024 workgroups takes 146 ms, every further round 64 ms
// (profiles/r05_zl_s2_rounds.txt) -- the dispatcher hands out blocks in index order, so with the
// tiles in spatial order the heavy ones of the LAST round decide the end.  Launches of up to
// RW_ORDER_MAX tiles are therefore dealt heaviest first; every XCD keeps its contiguous range of
// tiles (neighbours share rows in its L2) and walks it in descending order of weight.
// order[slot] = tile, or -1 for the padding slots.
static constexpr int RW_ORDER_MAX = 16384;
// Tile number -> (tz, ty, tx).  Round 6: Y-MAJOR numbering (y_major = 1).  Every XCD takes a contiguous range
// of tile numbers; numbered z-major, XCD 0 held the launch's lowest z-tiles and XCD 7 its highest, and a
// launch's z-edge tiles run ~20 % longer than its inner ones -- per-workgroup start / end stamps
// (tools/s2_wg_times.py, profiles/r06_h*): the XCDs of a 2 048-tile launch finished between 134 and 178 ms.
// Numbered y-major an XCD's range is a slab of y-tiles with EVERY z-tile in it (and z-neighbours, which
// share half of their rows, still meet in one L2).  PPP_RANK_TILE_MAJOR=z: the old numbering.
__device__ __forceinline__ void rw_tile_decode(int bid, int n_tiles, int tiles_y, int tiles_x, int y_major,
                                               int &tz_i, int &ty_i, int &tx_i) {
    tx_i = bid % tiles_x;
    if (y_major) {
        const int tiles_z = n_tiles / (tiles_y * tiles_x);
        tz_i = (bid / tiles_x) % tiles_z;
        ty_i = bid / (tiles_x * tiles_z);
    } else {
        ty_i = (bid / tiles_x) % tiles_y;
        tz_i = bid / (tiles_x * tiles_y);
    }
}
// weight of a tile = the chunks of 64 items its workgroup walks: over the VALID voxels u of the tile
// grown by the radius, ceil(items of u / 64) (+ 1 for the row's staging and barriers); 0 without an
// active centre (the workgroup leaves at once).  PPP_RANK_ORDER=centres: the active centres instead.
__global__ void __launch_bounds__(256)
    rank_tile_weight_kernel(const uint32_t *__restrict__ info, const uint8_t *__restrict__ valid, const ppp_box sb,
                            const int TZ, const int TY, const int TX, const int tiles_y, const int tiles_x,
                            const int by_centres, const int y_major, int32_t *__restrict__ weight, const Geo G) {
    const int bid = blockIdx.x;
    int tx_i, ty_i, tz_i;
    rw_tile_decode(bid, (int)gridDim.x, tiles_y, tiles_x, y_major, tz_i, ty_i, tx_i);
    const int sX = sb.x1 - sb.x0, sY = sb.y1 - sb.y0, sZ = sb.z1 - sb.z0;
    int c = 0, w = 0;
    for (int cl = threadIdx.x; cl < TZ * TY * TX; cl += blockDim.x) {
        const int z = tz_i * TZ + cl / (TX * TY), y = ty_i * TY + (cl / TX) % TY, x = tx_i * TX + cl % TX;
        if (z < sZ && y < sY && x < sX) c += (int)(info[((long long)z * sY + y) * sX + x] >> 31);
    }
    if (!by_centres) {
        const int c0z = sb.z0 + tz_i * TZ, c0y = sb.y0 + ty_i * TY, c0x = sb.x0 + tx_i * TX;
        const int tz = min(TZ, sb.z1 - c0z), ty = min(TY, sb.y1 - c0y), tx = min(TX, sb.x1 - c0x);
        const int uz0 = max(c0z - G.rz, G.bz0), uz1 = min(c0z + tz - 1 + G.rz, G.bz0 + G.bZ - 1);
        const int uy0 = max(c0y - G.ry, G.by0), uy1 = min(c0y + ty - 1 + G.ry, G.by0 + G.bY - 1);
        const int ux0 = max(c0x - G.rx, G.bx0), ux1 = min(c0x + tx - 1 + G.rx, G.bx0 + G.bX - 1);
        const int nuy = uy1 - uy0 + 1, nux = ux1 - ux0 + 1, nu = (uz1 - uz0 + 1) * nuy * nux;
        for (int k = threadIdx.x; k < nu; k += blockDim.x) {
            const int uz = uz0 + k / (nuy * nux), uy = uy0 + (k / nux) % nuy, ux = ux0 + k % nux;
            if (!valid[vox(G, uz, uy, ux)]) continue;
            const int nz = min(G.pz - 1, uz + G.rz - c0z) - max(0, uz + G.rz - (c0z + tz - 1)) + 1;
            const int ny = min(G.py - 1, uy + G.ry - c0y) - max(0, uy + G.ry - (c0y + ty - 1)) + 1;
            const int nx = min(G.px - 1, ux + G.rx - c0x) - max(0, ux + G.rx - (c0x + tx - 1)) + 1;
            if (nz > 0 && ny > 0 && nx > 0) w += (nz * ny * nx + 63) / 64 + 1;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o); w += __shfl_xor(w, o); }
    __shared__ int part[2][4];
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = c; part[1][threadIdx.x >> 6] = w; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int cs = part[0][0] + part[0][1] + part[0][2] + part[0][3];
        const int ws = part[1][0] + part[1][1] + part[1][2] + part[1][3];
        weight[bid] = by_centres ? cs : (cs ? ws : 0);
    }
}
// one workgroup per XCD range: its tiles in descending order of weight (position = the number of tiles
// of the range that come before: heavier, or as heavy with a smaller index)
__global__ void __launch_bounds__(256)
    rank_tile_order_kernel(const int32_t *__restrict__ weight, const int n_tiles, const int per_xcd,
                           int32_t *__restrict__ order) {
    __shared__ int32_t w[RW_ORDER_MAX / 8 + 8];
    const int lo = blockIdx.x * per_xcd, hi = min(lo + per_xcd, n_tiles), n = max(hi - lo, 0);
    for (int i = threadIdx.x; i < n; i += blockDim.x) w[i] = weight[lo + i];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int wi = w[i];
        int before = 0;
        for (int j = 0; j < n; ++j) before += (w[j] > wi || (w[j] == wi && j < i)) ? 1 : 0;
        order[lo + before] = lo + i;
    }
    for (int t = lo + n + threadIdx.x; t < lo + per_xcd; t += blockDim.x) order[t] = -1;
}

// ---- main kernel ---------------------------------------------------------------------------
template <int PZ, int PY, int PX, int TZ, int TY, int TX, bool P1>
__global__ void __launch_bounds__(64 * RW_WAVES, PPP_RW_MINWAVES(PX))
    rank_wg_kernel(const float *__restrict__ S, const uint32_t *__restrict__ M,
                   const uint32_t *__restrict__ info, const uint8_t *__restrict__ valid,
                   float *__restrict__ score, const ppp_box sb, const Geo G, const int tiles_y,
                   const int tiles_x, const int n_tiles, const int *__restrict__ any_e,
                   const int32_t *__restrict__ order, const int y_major
#ifdef PPP_RW_STAMPS
                   , uint32_t *__restrict__ stamps
#endif
                   ) {
    // (one launch of each form per call when the one-bit masks are possible: the pre-pass decides)
    if (any_e && (*any_e != 0) == P1) return;
#ifdef PPP_RW_STAMPS
    // (diagnostic build -DPPP_RW_STAMPS, tools/s2_wg_times.py: when does every workgroup start and
    // end, and on which CU -- the tile weights' array of the workspace is free once the order is made)
    const unsigned long long t_start = __builtin_readcyclecounter();
#endif
    constexpr int C = PZ * PY * PX, W16 = (C + 15) / 16, RZ = PZ / 2, RY = PY / 2, RX = PX / 2;
    constexpr int W16P = P1 ? rw_p1_words(C) : rw_mask_words(C);   // mask words per centre in M
    constexpr int NMW = P1 ? (C + 31) / 32 : W16;                  // ... that hold bits
    constexpr int WZ = 2 * PZ - 1, WY = 2 * PY - 1, WX = 2 * PX - 1, W = WZ * WY * WX, LC = (W - 1) / 2;
    constexpr int NTHR = 64 * RW_WAVES;
    constexpr int NST = (W + NTHR - 1) / NTHR;
    constexpr int NT = TZ * TY * TX;
    constexpr int UB = (TZ + 2 * RZ) * (TY + 2 * RY) * (TX + 2 * RX);
    // plane stride of the row image (see PPP_RW_ZRUNS)
    constexpr int SZ = WY * WX;
    constexpr int SZP = PPP_RW_ZRUNS(PX) ? SZ + ((9 - SZ % 32) + 32) % 32 : SZ;
    constexpr int LCP = LC + (PZ - 1) * (SZP - SZ);
    auto img = [](int e) -> int { return SZP == SZ ? e : (e / SZ) * SZP + e % SZ; };
    __shared__ float rowbuf[WZ * SZP + 2 * RW_PAD];
    __shared__ float accs[NT];
    __shared__ uint32_t act_bits[(NT + 31) / 32];
    __shared__ uint2 coefT[256];
    __shared__ uint32_t uvalid_bits[(UB + 63) / 64 * 2];
    __shared__ int any_act;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sX = sb.x1 - sb.x0, sY = sb.y1 - sb.y0, sZ = sb.z1 - sb.z0;
    const long long sbV = (long long)sX * sY * sZ;
    // XCD-aware order: consecutive blocks go to different XCDs; give each XCD a contiguous range
    // of tiles so that x-neighbours (which share a third of their rows) meet in one L2
    const int n_blocks = gridDim.x;
    const int per_xcd = (n_blocks + 7) / 8;
    const int slot = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    // (heavy tiles first within the XCD's range when the launcher made an order)
    const int bid = order ? order[slot] : slot;
    if (bid < 0 || bid >= n_tiles) return;
    int tx_i, ty_i, tz_i;
    rw_tile_decode(bid, n_tiles, tiles_y, tiles_x, y_major, tz_i, ty_i, tx_i);
    const int c0z = sb.z0 + tz_i * TZ, c0y = sb.y0 + ty_i * TY, c0x = sb.x0 + tx_i * TX;
    const int tz = min(TZ, sb.z1 - c0z), ty = min(TY, sb.y1 - c0y), tx = min(TX, sb.x1 - c0x);
    if (tz <= 0 || ty <= 0 || tx <= 0) return;

    auto sb_index = [&](int lz, int ly, int lx) -> long long {
        return ((long long)(c0z + lz - sb.z0) * sY + (c0y + ly - sb.y0)) * sX + (c0x + lx - sb.x0);
    };
    // coefficient table: entry e -> four float16 {+2^-5 (P bit), -2^-5 (N bit), 0}
    for (int e = tid; e < 256; e += NTHR) {
        uint32_t h[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = ((e >> i) & 1) ? 0x2800u : (((e >> (4 + i)) & 1) ? 0xA800u : 0u);
        coefT[e] = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
    }
    if (tid == 0) any_act = 0;
    __syncthreads();
    static_assert(NT % 64 == 0, "tile size must be a multiple of the wave size");
    for (int cl = tid; cl < NT; cl += NTHR) {
        const int lx = cl % TX, ly = (cl / TX) % TY, lz = cl / (TX * TY);
        uint32_t v = 0;
        if (lz < tz && ly < ty && lx < tx) v = info[sb_index(lz, ly, lx)];
        const unsigned long long m = __ballot((v >> 31) != 0);
        if (lane == 0) {
            act_bits[cl >> 5] = (uint32_t)m; act_bits[(cl >> 5) + 1] = (uint32_t)(m >> 32);
            if (m) any_act = 1;
        }
        accs[cl] = 0.0f;
    }
    __syncthreads();
    if (!any_act) return;

    const long long rsY = G.bX, rsZ = (long long)G.bX * G.bY;
    const int uz0 = max(c0z - RZ, G.bz0), uz1 = min(c0z + tz - 1 + RZ, G.bz0 + G.bZ - 1);
    const int uy0 = max(c0y - RY, G.by0), uy1 = min(c0y + ty - 1 + RY, G.by0 + G.bY - 1);
    const int ux0 = max(c0x - RX, G.bx0), ux1 = min(c0x + tx - 1 + RX, G.bx0 + G.bX - 1);
    const int nuy = uy1 - uy0 + 1, nux = ux1 - ux0 + 1, nu = (uz1 - uz0 + 1) * nuy * nux;
    for (int k0 = 64 * wave; k0 < nu; k0 += NTHR) {
        const int k = k0 + lane;
        const bool ok = k < nu && valid[vox(G, uz0 + k / (nuy * nux), uy0 + (k / nux) % nuy, ux0 + k % nux)] != 0;
        const unsigned long long m = __ballot(ok);
        if (lane == 0) { uvalid_bits[k0 >> 5] = (uint32_t)m; uvalid_bits[(k0 >> 5) + 1] = (uint32_t)(m >> 32); }
    }
    __syncthreads();
    auto next_valid = [&](int k) -> int {
        while (k < nu) {
            const uint32_t wbits = uvalid_bits[k >> 5] >> (k & 31);
            if (wbits) return k + __builtin_ctz(wbits);
            k = (k | 31) + 1;
        }
        return nu;
    };
    auto row_src = [&](int k) -> const float * {
        const int uz = uz0 + k / (nuy * nux), uy = uy0 + (k / nux) % nuy, ux = ux0 + k % nux;
        return S + (((long long)row_slice(G, uz) * rsZ + (long long)(uy - G.by0) * rsY + (ux - G.bx0)) * W);
    };
    float st[NST];
    int uk = __builtin_amdgcn_readfirstlane(next_valid(0));
    if (uk < nu) {
        const float *src = row_src(uk);
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int e = tid + i * NTHR;
            if (e < W) rowbuf[RW_PAD + img(e)] = src[e] * 32.0f;
        }
    }
    __syncthreads();
    int turn = 0;   // rotates the wave that takes the first chunk of a row
    // geometry of a row (which pixels a of voxel u have their centre in the tile) and of the item a
    // lane takes in chunk i0 of it -- functions of the row index, so that the masks of the NEXT
    // chunk (of this row or of the next) can be requested ahead
    struct RowG { int uz, uy, ux, az0, ay0, ax0, nz, ny, nx, n_box; };
    struct ItemG { bool in; int cl, a, az, ay, ax; long long t, q0; };
    auto row_geom = [&](int k) -> RowG {
        RowG r;
        r.uz = uz0 + k / (nuy * nux); r.uy = uy0 + (k / nux) % nuy; r.ux = ux0 + k % nux;
        r.az0 = max(0, r.uz + RZ - (c0z + tz - 1));
        r.ay0 = max(0, r.uy + RY - (c0y + ty - 1));
        r.ax0 = max(0, r.ux + RX - (c0x + tx - 1));
        r.nz = min(PZ - 1, r.uz + RZ - c0z) - r.az0 + 1;
        r.ny = min(PY - 1, r.uy + RY - c0y) - r.ay0 + 1;
        r.nx = min(PX - 1, r.ux + RX - c0x) - r.ax0 + 1;
        r.n_box = (r.nz <= 0 || r.ny <= 0 || r.nx <= 0) ? 0 : r.nz * r.ny * r.nx;
        return r;
    };
    auto item_geom = [&](const RowG &r, int i0) -> ItemG {
        ItemG g;
        const int i = i0 + lane;
        g.in = i < r.n_box;
        const int ii = g.in ? i : 0;
        g.ax = r.ax0 + ii % r.nx;
        g.ay = PPP_RW_ZRUNS(PX) ? r.ay0 + ii / (r.nx * r.nz) : r.ay0 + (ii / r.nx) % r.ny;
        g.az = PPP_RW_ZRUNS(PX) ? r.az0 + (ii / r.nx) % r.nz : r.az0 + ii / (r.nx * r.ny);
        const int lz = r.uz + RZ - g.az - c0z, ly = r.uy + RY - g.ay - c0y, lx = r.ux + RX - g.ax - c0x;
        g.cl = (lz * TY + ly) * TX + lx;
        g.a = (g.az * PY + g.ay) * PX + g.ax;
        g.t = sb_index(lz, ly, lx);
        // (two-bit masks: quad 0 of the centre, in 16-byte units)
        g.q0 = rw_mask_quad0((long long)(c0z + lz - sb.z0) * sY + (c0y + ly - sb.y0), c0x + lx - sb.x0, sX, W16P / 4);
        return g;
    };
    // one-bit masks: the first PFQ 16-byte pieces of a centre's words + the word with P'[a], requested
    // one chunk ahead (PPP_RW_PREFETCH=0: every chunk loads its masks when it starts, as the two-bit form)
#ifndef PPP_RW_PREFETCH
#define PPP_RW_PREFETCH 1
#endif
    constexpr bool PF = PPP_RW_PREFETCH != 0;
    constexpr int PFQ = (W16P / 4) / 2;
    uint4 pfa[PFQ > 0 ? PFQ : 1];
    uint32_t pfw = 0u;
    bool pf_have = false;
    auto prefetch = [&](const RowG &r, int i0) {
        const ItemG g = item_geom(r, i0);
        const uint32_t *mrow = M + g.t * (long long)W16P;
        pfw = mrow[g.a >> 5];
        const uint4 *mc = reinterpret_cast<const uint4 *>(mrow);
#pragma unroll
        for (int q = 0; q < PFQ; ++q) pfa[q] = mc[q];
    };
    (void)prefetch; (void)pf_have; (void)pfw;
    while (uk < nu) {
        const int uk_next = __builtin_amdgcn_readfirstlane(next_valid(uk + 1));
        if (uk_next < nu) {
            const float *src = row_src(uk_next);
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int e = tid + i * NTHR;
#ifdef PPP_RW_ABL_NOROW
                st[i] = (float)e;                      // (timing experiment: no row fetches)
#else
                st[i] = e < W ? src[e] : 0.0f;
#endif
            }
        }
        // pixels a of this voxel whose centre c = u + R - a lies in the tile
        const RowG R = row_geom(uk);
        const int n_box = R.n_box;
        const int first = (wave + RW_WAVES - turn) % RW_WAVES;
        turn = (turn + ((n_box + 63) >> 6)) % RW_WAVES;
        if constexpr (P1 && PF) {
            // (the first chunk of a row is normally prefetched at the end of the row before)
            if (!pf_have && 64 * first < n_box) prefetch(R, 64 * first);
            pf_have = false;
        }
        for (int i0 = 64 * first; i0 < n_box; i0 += NTHR) {
            const ItemG it = item_geom(R, i0);
            const bool in = it.in;
            const int ax = it.ax, ay = it.ay, az = it.az, cl = it.cl, a = it.a;
            const long long t = it.t, q0 = it.q0;
            (void)in; (void)q0;
            bool active = it.in && ((act_bits[cl >> 5] >> (cl & 31)) & 1u) != 0;
            uint32_t mw[NMW];
            if constexpr (P1 && PF) {
                // ---- one-bit masks, software pipelined: the first half of this chunk's words and the
                // word that holds P'[a] were requested a chunk ago; the second half is requested now
                // and arrives behind the first half of the chain; then the next chunk's first half
                // (this row's next chunk of the wave, or the first chunk of the next row)
                const uint32_t aword = pfw;
#pragma unroll
                for (int q = 0; q < PFQ; ++q) {
                    if (4 * q < NMW) mw[4 * q] = pfa[q].x;
                    if (4 * q + 1 < NMW) mw[4 * q + 1] = pfa[q].y;
                    if (4 * q + 2 < NMW) mw[4 * q + 2] = pfa[q].z;
                    if (4 * q + 3 < NMW) mw[4 * q + 3] = pfa[q].w;
                }
                const uint4 *mc = reinterpret_cast<const uint4 *>(M + t * (long long)W16P);
#pragma unroll
                for (int q = PFQ; q < W16P / 4; ++q) {
                    const uint4 v = mc[q];
                    if (4 * q < NMW) mw[4 * q] = v.x;
                    if (4 * q + 1 < NMW) mw[4 * q + 1] = v.y;
                    if (4 * q + 2 < NMW) mw[4 * q + 2] = v.z;
                    if (4 * q + 3 < NMW) mw[4 * q + 3] = v.w;
                }
                if (i0 + NTHR < n_box) {
                    prefetch(R, i0 + NTHR);
                } else if (uk_next < nu) {
                    const RowG Rn = row_geom(uk_next);
                    const int first_n = (wave + RW_WAVES - turn) % RW_WAVES;
                    if (64 * first_n < Rn.n_box) { prefetch(Rn, 64 * first_n); pf_have = true; }
                }
                if (active) active = ((aword >> (a & 31)) & 1u) != 0;
                if (__ballot(active) == 0) continue;
            } else {
            // is a in P?  P bit of partner s: bit 8 (s >> 2 & 3) + (s & 3) of word s >> 4 (two-bit
            // masks), bit s & 31 of word s >> 5 (one-bit masks)
            if (active) {
                if constexpr (P1) active = ((M[t * (long long)W16P + (a >> 5)] >> (a & 31)) & 1u) != 0;
                else active = ((M[(q0 + (long long)(a >> 6) * RW_QSTRIDE) * 4 + ((a >> 4) & 3)] >> (8 * ((a >> 2) & 3) + (a & 3))) & 1u) != 0;
            }
            if (__ballot(active) == 0) continue;
            // (unconditional loads -- t is a centre of the tile for every lane -- then one select
            // per word: a predicated load is a branch per word)
#ifdef PPP_RW_ABL_NOMASK
            // (timing experiment: no mask loads -- every partner in P)
#pragma unroll
            for (int w = 0; w < NMW; ++w) mw[w] = (P1 ? 0xFFFFFFFFu : 0x0F0F0F0Fu) + (uint32_t)(t & 0);
#else
            {
                const uint4 *mc = reinterpret_cast<const uint4 *>(M) + (P1 ? t * (long long)(W16P / 4) : q0);
#pragma unroll
                for (int q = 0; q < W16P / 4; ++q) {
                    const uint4 v = mc[q * (P1 ? 1 : RW_QSTRIDE)];
                    if (4 * q < NMW) mw[4 * q] = v.x;
                    if (4 * q + 1 < NMW) mw[4 * q + 1] = v.y;
                    if (4 * q + 2 < NMW) mw[4 * q + 2] = v.z;
                    if (4 * q + 3 < NMW) mw[4 * q + 3] = v.w;
                }
            }
#endif
            }
            if constexpr (!P1) {
#pragma unroll
                for (int w = 0; w < W16; ++w) mw[w] = active ? mw[w] : 0u;
            }
            float acc = active ? accs[cl] : 0.0f;
            lds_f32_cvp2 row = (lds_f32_cvp2)(rowbuf + RW_PAD + LCP - (az * SZP + ay * WX + ax));
            // (experiment PPP_RW_ADDRREGS=1: the four reads of a group through address registers of
            // their own instead of one register + immediate offsets)
#ifndef PPP_RW_ADDRREGS
#define PPP_RW_ADDRREGS 0
#endif
            lds_f32_cvp2 rowc[4];
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) {
                rowc[i2] = row;
                if (PPP_RW_ADDRREGS) asm volatile("" : "+v"(rowc[i2]));
            }
            const int aw = a >> 4;
            // b in P counts only for b > a: P bits of the partners <= a go (N bits stay)
            const uint32_t above16 = ~((2u << (a & 15)) - 1u) & 0xFFFFu;
            const uint32_t keep_a = interleave16(above16, 0xFFFFu);
            // (a compile-time recursion, not a loop: at 9^3 the 46 x 16 terms exceed the size up to
            // which the compiler honours an unroll pragma, and a rolled loop computes every LDS
            // offset at run time)
            StaticFor<0, W16>::run([&](auto wc) {
                constexpr int w = decltype(wc)::value;
                uint32_t m;
                if constexpr (P1) {
                    // the two-bit word of these 16 partners from their P' bits: P counts for b > a,
                    // N = not P' for every b != a (partners beyond C do not exist)
                    constexpr uint32_t last = (w == W16 - 1 && (C & 15)) ? ((1u << (C & 15)) - 1u) : 0xFFFFu;
                    const uint32_t pw = (mw[w >> 1] >> (16 * (w & 1))) & 0xFFFFu;
                    const uint32_t pos = pw & (w < aw ? 0u : (w > aw ? 0xFFFFu : above16));
                    const uint32_t neg = ~pw & (w == aw ? ~(1u << (a & 15)) : 0xFFFFFFFFu) & last;
                    auto spread = [](uint32_t x) -> uint32_t {      // nibble j of x -> low nibble of byte j
                        x = (x | (x << 8)) & 0x00FF00FFu;
                        return (x | (x << 4)) & 0x0F0F0F0Fu;
                    };
                    m = active ? (spread(pos) | (spread(neg) << 4)) : 0u;
                } else {
                    m = mw[w] & (w < aw ? 0xF0F0F0F0u : (w > aw ? 0xFFFFFFFFu : keep_a));
                }
                if (__ballot(m != 0u) == 0) return;
                // table reads of the word first; then the four groups of four partners, the row
                // values of the next group in flight while the current chain runs
                uint2 cf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#ifdef PPP_RW_ABL_NOTABLE
                    // (timing experiment: no table reads)
                    if (w * 16 + j * 4 < C) { const uint32_t q = (m >> (8 * j)) & 0xFFu; cf[j] = make_uint2(0x28002800u ^ (q << 15), 0xA800A800u ^ (q << 7)); }
#else
                    if (w * 16 + j * 4 < C) cf[j] = coefT[(m >> (8 * j)) & 0xFFu];
#endif
                float rv[2][4];
                // (the reads of a group are issued LAST ELEMENT FIRST: the LDS returns in order, so the
                // wait of the group's first term covers the other three -- one s_waitcnt per group
                // instead of one per term, and an s_waitcnt costs its wave an issue slot like any
                // instruction; PPP_RW_REVREADS=0: first element first)
                auto load_rows = [&](int j, float (&r)[4]) {
#pragma unroll
                    for (int i3 = 0; i3 < 4; ++i3) {
                        const int i2 = PPP_RW_REVREADS ? 3 - i3 : i3;
                        const int b = w * 16 + j * 4 + i2;
#ifdef PPP_RW_ABL_NOLDS
                        if (b < C) r[i2] = __builtin_bit_cast(float, 0x3F800000u + (uint32_t)(lane + b));   // (timing experiment)
#else
                        if (b < C) r[i2] = rowc[i2][(b / (PY * PX)) * SZP + ((b / PX) % PY) * WX + b % PX];
#endif
                    }
                };
                load_rows(0, rv[0]);
                if (w * 16 + 4 < C) load_rows(1, rv[1]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (w * 16 + j * 4 < C) {
                        float r0[4];
#pragma unroll
                        for (int i2 = 0; i2 < 4; ++i2) r0[i2] = rv[j & 1][i2];
                        if (j + 2 < 4 && w * 16 + (j + 2) * 4 < C) load_rows(j + 2, rv[j & 1]);
#pragma unroll
                        for (int i2 = 0; i2 < 4; ++i2) {
                            const int b = w * 16 + j * 4 + i2;
                            if (b < C) {
                                const uint32_t c2 = i2 < 2 ? cf[j].x : cf[j].y;
                                acc = (i2 & 1) ? fma_mix_hi(r0[i2], c2, acc) : fma_mix_lo(r0[i2], c2, acc);
                            }
                        }
                    }
                }
            });
            if (active) accs[cl] = acc;
        }
        // ---- publish the next row
        __syncthreads();
        if (uk_next < nu) {
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int e = tid + i * NTHR;
                if (e < W) rowbuf[RW_PAD + img(e)] = st[i] * 32.0f;
            }
        }
        __syncthreads();
        uk = uk_next;
    }
    // ---- scores of the tile
    for (int cl = tid; cl < NT; cl += NTHR) {
        const int lx = cl % TX, ly = (cl / TX) % TY, lz = cl / (TX * TY);
        if (lz < tz && ly < ty && lx < tx && ((act_bits[cl >> 5] >> (cl & 31)) & 1u)) {
            const unsigned fg_cnt = info[sb_index(lz, ly, lx)] & 0x7FFFFFFFu;
            const float acc = accs[cl];
            score[vox(G, c0z + lz, c0y + ly, c0x + lx)] = G.norm_rank ? acc / (float)(fg_cnt > 1u ? fg_cnt : 1u) : acc;
        }
    }
#ifdef PPP_RW_STAMPS
    if (tid == 0 && stamps && blockIdx.x < RW_ORDER_MAX / 4) {
        const unsigned long long t_end = __builtin_readcyclecounter();
        unsigned hw_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        unsigned xcc_id;          // (s_memtime is a counter per XCD: the tool needs to know whose clock a stamp is)
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        stamps[4 * blockIdx.x + 0] = (uint32_t)(t_start >> 8);     // (256-cycle units: 32 bits hold minutes)
        stamps[4 * blockIdx.x + 1] = (uint32_t)(t_end >> 8);
        stamps[4 * blockIdx.x + 2] = hw_id;
        stamps[4 * blockIdx.x + 3] = (uint32_t)bid | ((xcc_id & 0xFu) << 24);
    }
#endif
}

static size_t up256w(size_t v) { return (v + 255) / 256 * 256; }

bool rank_wg_supported(const Geo &G) {
    static EnvSwitch wg_sw("PPP_RANK_WG");
    const bool off = wg_sw.get() && wg_sw.get()[0] == '0';
    if (off) return false;       // (PPP_RANK_WG=0: the one-wave kernel of ppp_rank_vm.hip, read once)
    return G.pz == G.py && G.py == G.px && (G.px == 5 || G.px == 7 || G.px == 9) && !G.count_pos_neg &&
           G.layout == PPP_CONS_VOXEL_MAJOR;
}

size_t rank_wg_workspace_bytes(const ppp_box &sb, const Geo &G) {
    const size_t sbV = (size_t)(sb.x1 - sb.x0) * (sb.y1 - sb.y0) * (sb.z1 - sb.z0);
    return up256w(rw_mask_bytes(sb.z1 - sb.z0, sb.y1 - sb.y0, sb.x1 - sb.x0, G.C)) + up256w(sbV * 4) + up256w((size_t)G.V) + 256 +
           2 * (size_t)RW_ORDER_MAX * 4;
}

template <typename T>
static hipError_t launch_rwg(const T *pred, const float *S, const uint8_t *ov, float *score,
                             const ppp_box &sb, void *work, const Geo &G, hipStream_t s) {
    const int sX = sb.x1 - sb.x0, sY = sb.y1 - sb.y0, sZ = sb.z1 - sb.z0;
    const size_t sbV = (size_t)sX * sY * sZ;
    char *p = (char *)work;
    uint32_t *M = (uint32_t *)p;    p += up256w(rw_mask_bytes(sZ, sY, sX, G.C));
    uint32_t *info = (uint32_t *)p; p += up256w(sbV * 4);
    uint8_t *valid = (uint8_t *)p;  p += up256w((size_t)G.V);
    int *any_e = (int *)p;          // one-bit masks: "a partner equals the threshold somewhere"
    if (G.bz0 > (sb.z0 - G.rz > 0 ? sb.z0 - G.rz : 0) || G.by0 > (sb.y0 - G.ry > 0 ? sb.y0 - G.ry : 0) ||
        G.bx0 > (sb.x0 - G.rx > 0 ? sb.x0 - G.rx : 0) ||
        G.bz0 + G.bZ < (sb.z1 + G.rz < G.Z ? sb.z1 + G.rz : G.Z) ||
        G.by0 + G.bY < (sb.y1 + G.ry < G.Y ? sb.y1 + G.ry : G.Y) ||
        G.bx0 + G.bX < (sb.x1 + G.rx < G.X ? sb.x1 + G.rx : G.X))
        return hipErrorInvalidValue;
    PPP_GRID_CHECK((G.V + 255) / 256, 256);
    PPP_GRID_CHECK((sbV + 255) / 256, 256);
    rank_valid2_kernel<T><<<dim3((unsigned)((G.V + 255) / 256)), dim3(256), 0, s>>>(pred, ov, valid, G);
    // one bit per partner where N is the complement of P up to values that equal the threshold
    // (background rule "v < TH": bg_lt >= th_gt) -- only with PPP_RANK_P1=1: it halves the mask
    // fetches and, with the prefetch, takes their latency off the chain's start, and is SLOWER
    // (112 x 176 x 176 / 9^3: 285 ms, 272 ms without the prefetch, against 254 ms for the two-bit masks;
    // 140^3 / 7^3: 72 vs 62 ms; profiles/r05_x_s2_one_bit_masks.txt): the 12 vector instructions that
    // rebuild a two-bit word from 16 mask bits cost more than the fetches they save -- the kernel is
    // not waiting for its masks.
    static EnvSwitch p1_sw("PPP_RANK_P1");
    const bool p1 = G.bg_lt >= G.th_gt && p1_sw.get() && p1_sw.get()[0] == '1';
    if (p1) {
        hipError_t em = hipMemsetAsync(any_e, 0, 4, s);
        if (em != hipSuccess) return em;
        rank_masks_p1_kernel<T><<<dim3((unsigned)((sbV + 255) / 256)), dim3(256), 0, s>>>(pred, ov, sb, M, info, any_e,
                                                                                          score, G);
    }
    // Tile of centres per workgroup: 8 x 16 x 16 when that still gives every CU four workgroups
    // (a row is then staged 4.5 times per centre at 9^3), else 8 x 8 x 16 (6 times).
    // PPP_RANK_WG_TILE=8x8x16 | 8x16x16 overrides.
    const long long big_tiles = (long long)((sZ + 7) / 8) * ((sY + 15) / 16) * ((sX + 15) / 16);
    bool big = big_tiles >= 4 * 256;
    static EnvSwitch tile_sw("PPP_RANK_WG_TILE");
    if (const char *e = tile_sw.get()) big = strcmp(e, "8x16x16") == 0 ? true : (strcmp(e, "8x8x16") == 0 ? false : big);
    // 9^3 with the z-run enumeration: 8 x 8 x 16 also at large boxes (a 112 x 176 x 176 launch,
    // the size of the 512^3 tiles: 256 ms against 265 ms for 8 x 16 x 16 -- and for a 16 x 8 x 16 tile,
    // which has half of its items in conflict-free rows: profiles/r04_k_s2_tiles*.txt)
    if (G.px == 9 && PPP_RW_ZRUNS(9) && !tile_sw.get()) big = false;
    bool tall = tile_sw.get() && strcmp(tile_sw.get(), "16x8x16") == 0;
    // the caller's choice (ppp_params.rank_tile; the environment switch, a development aid, wins): which
    // shape is faster depends on the box -- 8 x 8 x 16 by 3 % where the ranking kernel runs at 161 ms per
    // 2 048-tile launch, 16 x 8 x 16 by 12 % where it runs at 215 ms (profiles/r06_l_bench_ab_tiles.txt)
    if (G.rank_tile && !tile_sw.get()) { big = G.rank_tile == 2; tall = G.rank_tile == 3; }
    const int TZ = tall ? 16 : 8, TY = tall ? 8 : (big ? 16 : 8), TX = 16;
    const int tiles_z = (sZ + TZ - 1) / TZ, tiles_y = (sY + TY - 1) / TY, tiles_x = (sX + TX - 1) / TX;
    const long long n_tiles = (long long)tiles_z * tiles_y * tiles_x;
    const long long n_blocks = (n_tiles + 7) / 8 * 8;
    PPP_GRID_CHECK(n_blocks, 64 * RW_WAVES);
    // heavy tiles first (PPP_RANK_ORDER=0: spatial order)
    static EnvSwitch order_sw("PPP_RANK_ORDER");
    int32_t *weight = (int32_t *)((char *)any_e + 256), *order = weight + RW_ORDER_MAX;
    if (n_tiles > RW_ORDER_MAX || (order_sw.get() && order_sw.get()[0] == '0')) order = nullptr;
    const bool by_centres = order_sw.get() && order_sw.get()[0] == 'c';
    static EnvSwitch major_sw("PPP_RANK_TILE_MAJOR");
    const int y_major = (major_sw.get() && major_sw.get()[0] == 'z') ? 0 : 1;
    // (occupancy experiment: PPP_RANK_WG_DYNLDS=<bytes> of unused dynamic LDS per workgroup)
    static EnvSwitch dyn_sw("PPP_RANK_WG_DYNLDS");
    const unsigned dyn_lds = dyn_sw.get() ? (unsigned)atoi(dyn_sw.get()) : 0u;
#ifdef PPP_RW_STAMPS
#define PPP_RW_STAMP_ARG , (uint32_t *)weight
#else
#define PPP_RW_STAMP_ARG
#endif
#define PPP_RW_LAUNCH1(A_, D_, E_, F_, P1_)                                                                 \
    rank_wg_kernel<A_, A_, A_, D_, E_, F_, P1_><<<dim3((unsigned)n_blocks), dim3(64 * RW_WAVES), dyn_lds, s>>>( \
        S, M, info, valid, score, sb, G, tiles_y, tiles_x, (int)n_tiles, p1 ? any_e : nullptr, order, y_major PPP_RW_STAMP_ARG)
#define PPP_RW_LAUNCH(A_, D_, E_, F_)                                                                       \
    do {                                                                                                    \
        if (p1 && pass == 0) PPP_RW_LAUNCH1(A_, D_, E_, F_, true);                                          \
        else PPP_RW_LAUNCH1(A_, D_, E_, F_, false);                                                         \
    } while (0)
#define PPP_RW_CASE(P)                                                                                      \
    case P:                                                                                                 \
        if (tall) PPP_RW_LAUNCH(P, 16, 8, 16);                                                              \
        else if (big) PPP_RW_LAUNCH(P, 8, 16, 16);                                                          \
        else PPP_RW_LAUNCH(P, 8, 8, 16);                                                                    \
        break;
    // pass 0: the one-bit form (when possible); then the two-bit pre-pass and kernel -- both return
    // at once unless the one-bit pre-pass found a partner that equals the threshold
    for (int pass = p1 ? 0 : 1; pass < 2; ++pass) {
        if (pass == 1)
            rank_masks_il_kernel<T><<<dim3((unsigned)((sbV + 255) / 256)), dim3(256), 0, s>>>(pred, ov, sb, M, info, score,
                                                                                              p1 ? any_e : nullptr, G);
        if (order && pass == (p1 ? 0 : 1)) {      // (`info` is the same from either pre-pass)
            rank_tile_weight_kernel<<<dim3((unsigned)n_tiles), dim3(256), 0, s>>>(info, valid, sb, TZ, TY, TX, tiles_y, tiles_x,
                                                                                  by_centres ? 1 : 0, y_major, weight, G);
            rank_tile_order_kernel<<<dim3(8), dim3(256), 0, s>>>(weight, (int)n_tiles, (int)(n_blocks / 8), order);
        }
#ifdef PPP_RW_STAMPS
        (void)hipMemsetAsync(weight, 0, (size_t)RW_ORDER_MAX * 4, s);      // (the weights are spent: room for the stamps)
#endif
        switch (G.px) {
            PPP_RW_CASE(5)
            PPP_RW_CASE(7)
            PPP_RW_CASE(9)
        default:
            return hipErrorNotSupported;
        }
    }
#undef PPP_RW_LAUNCH
#undef PPP_RW_LAUNCH1
#undef PPP_RW_STAMP_ARG
#undef PPP_RW_CASE
    return hipGetLastError();
}

hipError_t launch_rank_wg(const void *pred, int dtype, const float *S, const uint8_t *ov, float *score,
                          const ppp_box &sb, void *work, const Geo &G, hipStream_t s) {
    if (!rank_wg_supported(G)) return hipErrorNotSupported;
    return dtype == PPP_F16 ? launch_rwg<__half>((const __half *)pred, S, ov, score, sb, work, G, s)
                            : launch_rwg<float>((const float *)pred, S, ov, score, sb, work, G, s);
}

}  // namespace ppp
