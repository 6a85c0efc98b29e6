// ppp_rank_vm.hip -- S2 (patch ranking) on the VOXEL-MAJOR consensus, row-stationary.
//
// Reference: cuda/rankPatches.cu:28-147.  For a foreground centre c, with P / N / V the sets of
// patch pixels that are foreground / background / valid for c (see ppp_rank_v2.hip),
//     score(c) = sum over a in P (raster)  sum over b != a, V[b] (raster)
//                    + cons(z_a, z_b)   if b in P and b > a
//                    - cons(z_a, z_b)   if b in N
// accumulated sequentially in float.  cons(z_a, z_b) is entry q = b - a of the voxel-major row
// S[z_a][.], z_a = c + a - rad: for a fixed first pixel the whole inner loop reads ONE row.
//
// The gather kernel (ppp_rank_v2.hip, lane = centre) fetches every consensus entry from HBM about
// 50 times (0.70 TB per launch on the 140^3 / 7^3 volume): a row is needed by the 343 centres
// whose window holds its voxel, at 343 different moments.  Here the row is the stationary
// operand: a wave owns a TILE of centres, walks the voxels u of the tile grown by the patch
// radius in raster order, stages S[u][.] in LDS once, and lets every centre of the tile whose
// window holds u take its step a = u - c + rad -- one lane per (centre, a), all lanes running the
// same 343-long inner loop with compile-time LDS offsets.  Raster order of u is raster order of
// a for every centre, so each centre's float chain is exactly the reference's.  Accumulators live
// in LDS between the steps of a centre.  The per-centre P / N bit masks and the closed-form
// pair count come from a coalesced pre-pass over the prediction.
#include <stdlib.h>
#include <string.h>

#include "ppp_kernels.hpp"

namespace ppp {

static constexpr int RV_PAD = 8;
// row reads through a volatile LDS pointer: one ds_read_b32 with a 16-bit immediate offset each.
// (Merged into ds_read2_b32 -- 8-bit offsets -- the compiler needs an address add for most pairs.)
typedef const volatile __attribute__((address_space(3))) float *lds_f32_cvp;

// ---- pre-pass: per-centre masks, pair counts, border / background scores ---------------------
// thread per centre of the score box; masks are stored word-major over the box so that the main
// kernel's loads (lanes = neighbouring centres) coalesce
template <typename T>
__global__ void __launch_bounds__(256)
    rank_masks_kernel(const T *__restrict__ pred, const uint8_t *__restrict__ ov, const ppp_box sb,
                      uint32_t *__restrict__ Pb, uint32_t *__restrict__ Nb, uint32_t *__restrict__ info,
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
        const int words = (G.C + 31) / 32;
        unsigned nP = 0, nV = 0;
        int r = 0;
        for (int w = 0; w < words; ++w) {
            uint32_t p = 0, n = 0;
            for (int b = 0; b < 32 && r < G.C; ++b, ++r) {
                const int z = cz + r / (G.py * G.px) - G.rz, y = cy + (r / G.px) % G.py - G.ry,
                          x = cx + r % G.px - G.rx;
                const long long lz = vox(G, z, y, x);
                const bool valid = ldf(mid, lz) > G.th_gt && (!G.use_overlap || ov[lz] == 0);
                const float val = ldf(pred, (long long)r * G.V + lc);
                if (valid) ++nV;
                if (valid && val > G.th_gt) p |= 1u << b;
                if (valid && val < G.bg_lt) n |= 1u << b;
            }
            Pb[(long long)w * sbV + t] = p;
            Nb[(long long)w * sbV + t] = n;
            nP += __popc(p);
        }
        // fgCnt = |P| (|V| - 1) - |P| (|P| - 1) / 2   (rankPatches.cu:139, see ppp_rank_v2.hip)
        const unsigned fg_cnt = nP ? nP * (nV - 1u) - nP * (nP - 1u) / 2u : 0u;
        inf = 0x80000000u | fg_cnt;
        if (nP == 0) score[lc] = 0.0f;   // no first pixel: acc = 0, 0 / max(1, 0)
    }
    info[t] = inf;
}

// voxels that can be a first pixel at all: pred[mid] > TH and not an overlap voxel
template <typename T>
__global__ void __launch_bounds__(256)
    rank_valid_kernel(const T *__restrict__ pred, const uint8_t *__restrict__ ov, uint8_t *__restrict__ valid,
                      const Geo G) {
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= G.V) return;
    valid[v] = (ldf(pred, (long long)G.mid * G.V + v) > G.th_gt && (!G.use_overlap || ov[v] == 0)) ? 1 : 0;
}

// ---- main kernel ---------------------------------------------------------------------------------
template <int PZ, int PY, int PX, int TZ, int TY, int TX>
__global__ void __launch_bounds__(64, PZ * PY * PX <= 343 ? 3 : 2)
    rank_vm_kernel(const float *__restrict__ S, const uint32_t *__restrict__ Pb,
                   const uint32_t *__restrict__ Nb, const uint32_t *__restrict__ info,
                   const uint8_t *__restrict__ valid, float *__restrict__ score, const ppp_box sb,
                   const Geo G, const int tiles_y, const int tiles_x) {
    constexpr int C = PZ * PY * PX, WORDS = (C + 31) / 32, RZ = PZ / 2, RY = PY / 2, RX = PX / 2;
    constexpr int WZ = 2 * PZ - 1, WY = 2 * PY - 1, WX = 2 * PX - 1, W = WZ * WY * WX, LC = (W - 1) / 2;
    constexpr int NST = (W + 63) / 64;
    constexpr int NT = TZ * TY * TX;
    __shared__ float rowbuf[W + 2 * RV_PAD];
    __shared__ float accs[NT];
    constexpr int UB = (TZ + 2 * RZ) * (TY + 2 * RY) * (TX + 2 * RX);
    __shared__ uint32_t act_bits[(NT + 31) / 32];      // centre of the tile takes part
    // byte of mask bits -> 8 coefficient bytes (byte i = bit i ? code : 0); code 0x3D / 0xBD is
    // the top byte of +-1/32 as a float (all lower bits zero)
    __shared__ uint2 coefP[256], coefN[256];
    __shared__ uint32_t uvalid_bits[(UB + 63) / 64 * 2];   // voxel of the grown tile can be a first pixel
    const int lane = threadIdx.x;
    const int sX = sb.x1 - sb.x0, sY = sb.y1 - sb.y0, sZ = sb.z1 - sb.z0;
    const long long sbV = (long long)sX * sY * sZ;
    // XCD-aware order: consecutive blocks are spread over the 8 XCDs by the dispatcher; give
    // each XCD a contiguous range of tiles so that neighbouring tiles (which share halo rows)
    // meet in the same L2
    const int n_blocks = gridDim.x;
    const int per_xcd = (n_blocks + 7) / 8;
    int bid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (bid >= n_blocks) return;   // (per_xcd * 8 >= n_blocks: the tail ids of the last XCDs are empty)
    const int tx_i = bid % tiles_x, ty_i = (bid / tiles_x) % tiles_y, tz_i = bid / (tiles_x * tiles_y);
    const int c0z = sb.z0 + tz_i * TZ, c0y = sb.y0 + ty_i * TY, c0x = sb.x0 + tx_i * TX;
    const int tz = min(TZ, sb.z1 - c0z), ty = min(TY, sb.y1 - c0y), tx = min(TX, sb.x1 - c0x);
    if (tz <= 0 || ty <= 0 || tx <= 0) return;

    // sb-linear index of tile centre cl = (lz * TY + ly) * TX + lx
    auto sb_index = [&](int lz, int ly, int lx) -> long long {
        return ((long long)(c0z + lz - sb.z0) * sY + (c0y + ly - sb.y0)) * sX + (c0x + lx - sb.x0);
    };
    for (int e = lane; e < 256; e += 64) {
        uint32_t lo = 0, hi = 0;
        for (int i = 0; i < 4; ++i) {
            lo |= ((e >> i) & 1u) << (8 * i);
            hi |= ((e >> (4 + i)) & 1u) << (8 * i);
        }
        coefP[e] = make_uint2(lo * 0x3Du, hi * 0x3Du);
        coefN[e] = make_uint2(lo * 0xBDu, hi * 0xBDu);
    }
    bool any_active = false;
    static_assert(NT % 64 == 0, "tile size must be a multiple of the wave size");
    for (int cl = lane; cl < NT; cl += 64) {
        const int lx = cl % TX, ly = (cl / TX) % TY, lz = cl / (TX * TY);
        uint32_t v = 0;
        if (lz < tz && ly < ty && lx < tx) v = info[sb_index(lz, ly, lx)];
        const unsigned long long m = __ballot((v >> 31) != 0);
        if (lane == 0) { act_bits[cl >> 5] = (uint32_t)m; act_bits[(cl >> 5) + 1] = (uint32_t)(m >> 32); }
        accs[cl] = 0.0f;
        any_active |= m != 0ull;
    }
    if (!any_active) return;

    const long long rsY = G.bX, rsZ = (long long)G.bX * G.bY;
    // voxels of the tile grown by the radius, clipped to the consensus box (rows outside it are
    // rows of voxels outside the volume: no centre of the interior has them in its window)
    const int uz0 = max(c0z - RZ, G.bz0), uz1 = min(c0z + tz - 1 + RZ, G.bz0 + G.bZ - 1);
    const int uy0 = max(c0y - RY, G.by0), uy1 = min(c0y + ty - 1 + RY, G.by0 + G.bY - 1);
    const int ux0 = max(c0x - RX, G.bx0), ux1 = min(c0x + tx - 1 + RX, G.bx0 + G.bX - 1);
    const int nuy = uy1 - uy0 + 1, nux = ux1 - ux0 + 1, nu = (uz1 - uz0 + 1) * nuy * nux;
    for (int k0 = 0; k0 < nu; k0 += 64) {
        const int k = k0 + lane;
        const bool ok = k < nu && valid[vox(G, uz0 + k / (nuy * nux), uy0 + (k / nux) % nuy, ux0 + k % nux)] != 0;
        const unsigned long long m = __ballot(ok);
        if (lane == 0) { uvalid_bits[k0 >> 5] = (uint32_t)m; uvalid_bits[(k0 >> 5) + 1] = (uint32_t)(m >> 32); }
    }
    __syncthreads();
    // next voxel >= k of the grown tile that can be a first pixel (nu if none); wave-uniform
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
        return S + (((long long)(uz - G.bz0) * rsZ + (long long)(uy - G.by0) * rsY + (ux - G.bx0)) * W);
    };
    // software pipeline: the row of the next voxel travels HBM -> registers while the current
    // row (in LDS) is consumed
    float st[NST];
    int uk = __builtin_amdgcn_readfirstlane(next_valid(0));
    if (uk < nu) {
        const float *src = row_src(uk);
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int e = lane + i * 64;
            if (e < W) rowbuf[RV_PAD + e] = src[e] * 32.0f;
        }
    }
    __syncthreads();
    while (uk < nu) {
                const int uz = uz0 + uk / (nuy * nux), uy = uy0 + (uk / nux) % nuy, ux = ux0 + uk % nux;
                const int uk_next = __builtin_amdgcn_readfirstlane(next_valid(uk + 1));
                if (uk_next < nu) {
                    const float *src = row_src(uk_next);
#pragma unroll
                    for (int i = 0; i < NST; ++i) {
                        const int e = lane + i * 64;
                        st[i] = e < W ? src[e] : 0.0f;
                    }
                }
                // pixels a of this voxel whose centre c = u + R - a lies in the tile
                const int az0 = max(0, uz + RZ - (c0z + tz - 1)), az1 = min(PZ - 1, uz + RZ - c0z);
                const int ay0 = max(0, uy + RY - (c0y + ty - 1)), ay1 = min(PY - 1, uy + RY - c0y);
                const int ax0 = max(0, ux + RX - (c0x + tx - 1)), ax1 = min(PX - 1, ux + RX - c0x);
                const int nz = az1 - az0 + 1, ny = ay1 - ay0 + 1, nx = ax1 - ax0 + 1;
                const int n_box = (nz <= 0 || ny <= 0 || nx <= 0) ? 0 : nz * ny * nx;
                for (int i0 = 0; i0 < n_box; i0 += 64) {
                    const int i = i0 + lane;
                    const bool in = i < n_box;
                    const int ii = in ? i : 0;
                    const int ax = ax0 + ii % nx, ay = ay0 + (ii / nx) % ny, az = az0 + ii / (nx * ny);
                    const int lz = uz + RZ - az - c0z, ly = uy + RY - ay - c0y, lx = ux + RX - ax - c0x;
                    const int cl = (lz * TY + ly) * TX + lx;
                    const int a = (az * PY + ay) * PX + ax;
                    const long long t = sb_index(lz, ly, lx);
                    // (prefetching the next chunk's mask words -- for every item, before it is known
                    // whether a is in P -- was measured slower: 73 -> 80 ms at 140^3 / 7^3)
                    bool active = in && ((act_bits[cl >> 5] >> (cl & 31)) & 1u) != 0;
                    if (active) active = ((Pb[(long long)(a >> 5) * sbV + t] >> (a & 31)) & 1u) != 0;
                    if (__ballot(active) == 0) continue;
                    uint32_t pw[WORDS], nw[WORDS];
#pragma unroll
                    for (int w = 0; w < WORDS; ++w) {
                        pw[w] = active ? Pb[(long long)w * sbV + t] : 0u;
                        nw[w] = active ? Nb[(long long)w * sbV + t] : 0u;
                    }
                    float acc = active ? accs[cl] : 0.0f;
                    lds_f32_cvp row = (lds_f32_cvp)(rowbuf + RV_PAD + LC - ((az * WY + ay) * WX + ax));
                    const int aw = a >> 5;
                    const uint32_t above = ~((2u << (a & 31)) - 1u);   // bits of a's word above a
#pragma unroll
                    for (int w = 0; w < WORDS; ++w) {
                        // b in P counts only for b > a
                        const uint32_t pos = pw[w] & (w < aw ? 0u : (w > aw ? 0xFFFFFFFFu : above));
                        const uint32_t neg = nw[w];
                        if (__ballot((pos | neg) != 0u) == 0) continue;
                        // The staged row holds 32 * S (exact).  A term is fma(32 S, c, acc) with
                        // c = +1/32 (b in P, b > a), -1/32 (b in N) or 0: the product is exactly
                        // +-S or 0, so the fma rounds once, like the reference's acc += / -= S.
                        // The coefficients of 8 partners come as bytes from two table reads; a
                        // byte moved to the top of a register IS the float (v_perm_b32).
                        // all table reads of the word first, then the groups of 8 partners with the
                        // row values of the next group already in flight
                        uint32_t c_lo[4], c_hi[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            if (w * 32 + g * 8 < C) {
                                const uint2 cp = coefP[(pos >> (8 * g)) & 0xFFu], cn = coefN[(neg >> (8 * g)) & 0xFFu];
                                c_lo[g] = cp.x | cn.x;
                                c_hi[g] = cp.y | cn.y;
                            }
                        }
                        float rv[2][8];
                        auto load_rows = [&](int g, float (&r)[8]) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const int b = w * 32 + g * 8 + i;
                                if (b < C) r[i] = row[((b / (PY * PX)) * WY + (b / PX) % PY) * WX + b % PX];
                            }
                        };
                        load_rows(0, rv[0]);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            if (w * 32 + g * 8 < C) {
                                if (g + 1 < 4 && w * 32 + (g + 1) * 8 < C) load_rows(g + 1, rv[(g + 1) & 1]);
#pragma unroll
                                for (int i = 0; i < 8; ++i) {
                                    const int b = w * 32 + g * 8 + i;
                                    if (b < C) {
                                        const uint32_t c = __builtin_amdgcn_perm(0u, i < 4 ? c_lo[g] : c_hi[g],
                                                                                 0x000C0C0Cu | ((uint32_t)(i & 3) << 24));
                                        acc = __builtin_fmaf(rv[g & 1][i], __uint_as_float(c), acc);
                                    }
                                }
                            }
                        }
                    }
                    if (active) accs[cl] = acc;
                }
                // ---- publish the next row
                __syncthreads();
                if (uk_next < nu) {
#pragma unroll
                    for (int i = 0; i < NST; ++i) {
                        const int e = lane + i * 64;
                        if (e < W) rowbuf[RV_PAD + e] = st[i] * 32.0f;
                    }
                }
                __syncthreads();
                uk = uk_next;
    }
    // ---- scores of the tile
    for (int cl = lane; cl < NT; cl += 64) {
        const int lx = cl % TX, ly = (cl / TX) % TY, lz = cl / (TX * TY);
        if (lz < tz && ly < ty && lx < tx && ((act_bits[cl >> 5] >> (cl & 31)) & 1u)) {
            const unsigned fg_cnt = info[sb_index(lz, ly, lx)] & 0x7FFFFFFFu;
            const float acc = accs[cl];
            score[vox(G, c0z + lz, c0y + ly, c0x + lx)] = G.norm_rank ? acc / (float)(fg_cnt > 1u ? fg_cnt : 1u) : acc;
        }
    }
}

static size_t up256r(size_t v) { return (v + 255) / 256 * 256; }

// 0 = no kernel; else an id of the instantiated patch shape
static int rank_vm_shape(const Geo &G) {
    if (G.pz == G.py && G.py == G.px && (G.px == 3 || G.px == 5 || G.px == 7 || G.px == 9)) return G.px;
    if (G.pz == 1 && G.py == G.px && (G.px == 5 || G.px == 7 || G.px == 9 || G.px == 25)) return 100 + G.px;
    return 0;
}
bool rank_vm_supported(const Geo &G) {
    return rank_vm_shape(G) != 0 && !G.count_pos_neg && G.layout == PPP_CONS_VOXEL_MAJOR;
}

size_t rank_vm_workspace_bytes(const ppp_box &sb, const Geo &G) {
    const size_t sbV = (size_t)(sb.x1 - sb.x0) * (sb.y1 - sb.y0) * (sb.z1 - sb.z0);
    const size_t words = (size_t)(G.C + 31) / 32;
    const size_t one_wave = 2 * up256r(words * sbV * 4) + up256r(sbV * 4) + up256r((size_t)G.V);
    // (either kernel may serve the call: ppp_rank_wg.hip takes the cubic 5 / 7 / 9 patches)
    const size_t wg = rank_wg_supported(G) ? rank_wg_workspace_bytes(sb, G) : 0;
    return one_wave > wg ? one_wave : wg;
}

template <typename T>
static hipError_t launch_rv(const T *pred, const float *S, const uint8_t *ov, float *score,
                            const ppp_box &sb, void *work, const Geo &G, hipStream_t s) {
    const int sX = sb.x1 - sb.x0, sY = sb.y1 - sb.y0, sZ = sb.z1 - sb.z0;
    const size_t sbV = (size_t)sX * sY * sZ;
    const size_t words = (size_t)(G.C + 31) / 32;
    char *p = (char *)work;
    uint32_t *Pb = (uint32_t *)p;   p += up256r(words * sbV * 4);
    uint32_t *Nb = (uint32_t *)p;   p += up256r(words * sbV * 4);
    uint32_t *info = (uint32_t *)p; p += up256r(sbV * 4);
    uint8_t *valid = (uint8_t *)p;
    // the consensus box must hold every voxel of the volume within the radius of the score box
    if (G.bz0 > (sb.z0 - G.rz > 0 ? sb.z0 - G.rz : 0) || G.by0 > (sb.y0 - G.ry > 0 ? sb.y0 - G.ry : 0) ||
        G.bx0 > (sb.x0 - G.rx > 0 ? sb.x0 - G.rx : 0) ||
        G.bz0 + G.bZ < (sb.z1 + G.rz < G.Z ? sb.z1 + G.rz : G.Z) ||
        G.by0 + G.bY < (sb.y1 + G.ry < G.Y ? sb.y1 + G.ry : G.Y) ||
        G.bx0 + G.bX < (sb.x1 + G.rx < G.X ? sb.x1 + G.rx : G.X))
        return hipErrorInvalidValue;
    PPP_GRID_CHECK((G.V + 255) / 256, 256);
    PPP_GRID_CHECK((sbV + 255) / 256, 256);
    rank_valid_kernel<T><<<dim3((unsigned)((G.V + 255) / 256)), dim3(256), 0, s>>>(pred, ov, valid, G);
    rank_masks_kernel<T><<<dim3((unsigned)((sbV + 255) / 256)), dim3(256), 0, s>>>(pred, ov, sb, Pb, Nb, info, score, G);
    // tile of centres per wave: 8 x 8 x 8 by default; PPP_RANK_TILE=8x8x16 doubles it along x
    // (fewer row stagings per centre and fuller waves, half as many tiles to spread over the CUs)
    // Tile of centres per wave.  One wave per tile, ~4 waves per SIMD resident: 4096 tiles run at
    // a time, so a volume of few tiles is cut finer (4 x 8 x 8: 140^3 -> 10.7 k tiles, 2.6 rounds
    // instead of 1.3 rounds of which the second is a third full: 95 -> 86 ms), a large one coarser
    // (8 x 8 x 8: 5.4 instead of 7.7 row stagings per centre).  PPP_RANK_TILE overrides.  2-d
    // patches (pz = 1): 1 x 8 x 8.
    const int shape_id = rank_vm_shape(G);
    int tile_kind = ((long long)((sZ + 7) / 8) * ((sY + 7) / 8) * ((sX + 7) / 8) < 3 * 4096) ? 2 : 0;
    // 7^3: five slices of centres instead of four -- an inner voxel then serves 5 * 49 = 245
    // (centre, a) items = 3.8 chunks of 64 lanes instead of 196 = 3.06 (the fourth chunk 6 % full)
    if (tile_kind == 2 && shape_id == 7) tile_kind = 4;
    static EnvSwitch tile_sw("PPP_RANK_TILE");
    if (const char *e = tile_sw.get())
        tile_kind = strcmp(e, "8x8x16") == 0 ? 1 : (strcmp(e, "4x8x8") == 0 ? 2 : (strcmp(e, "8x8x8") == 0 ? 0 :
                    (strcmp(e, "5x8x8") == 0 ? 4 : (strcmp(e, "7x8x8") == 0 ? 5 : tile_kind))));
    if (shape_id > 100) tile_kind = 3;
    const int TZ = tile_kind == 3 ? 1 : (tile_kind == 2 ? 4 : (tile_kind == 4 ? 5 : (tile_kind == 5 ? 7 : 8))), TY = 8,
              TX = tile_kind == 1 ? 16 : 8;
    const int tiles_z = (sZ + TZ - 1) / TZ, tiles_y = (sY + TY - 1) / TY, tiles_x = (sX + TX - 1) / TX;
    const long long n_tiles = (long long)tiles_z * tiles_y * tiles_x;
    const long long n_blocks = (n_tiles + 7) / 8 * 8;
    PPP_GRID_CHECK(n_blocks, 64);
#define PPP_RV_LAUNCH(A_, B_, C_, D_, E_, F_)                                                          \
    rank_vm_kernel<A_, B_, C_, D_, E_, F_><<<dim3((unsigned)n_blocks), dim3(64), 0, s>>>(              \
        S, Pb, Nb, info, valid, score, sb, G, tiles_y, tiles_x)
#define PPP_RV_CASE(P)                                                                                 \
    case P:                                                                                            \
        if (tile_kind == 1) PPP_RV_LAUNCH(P, P, P, 8, 8, 16);                                          \
        else if (tile_kind == 2) PPP_RV_LAUNCH(P, P, P, 4, 8, 8);                                      \
        else if (tile_kind == 4) PPP_RV_LAUNCH(P, P, P, 5, 8, 8);                                      \
        else if (tile_kind == 5) PPP_RV_LAUNCH(P, P, P, 7, 8, 8);                                      \
        else PPP_RV_LAUNCH(P, P, P, 8, 8, 8);                                                          \
        break;
#define PPP_RV_CASE2D(P)                                                                               \
    case 100 + P:                                                                                      \
        PPP_RV_LAUNCH(1, P, P, 1, 8, 8);                                                               \
        break;
    switch (shape_id) {
        PPP_RV_CASE(3)
        PPP_RV_CASE(5)
        PPP_RV_CASE(7)
        PPP_RV_CASE(9)
        PPP_RV_CASE2D(5)
        PPP_RV_CASE2D(7)
        PPP_RV_CASE2D(9)
        PPP_RV_CASE2D(25)
    default:
        return hipErrorNotSupported;
    }
#undef PPP_RV_LAUNCH
#undef PPP_RV_CASE2D
#undef PPP_RV_CASE
    return hipGetLastError();
}

hipError_t launch_rank_vm(const void *pred, int dtype, const float *S, const uint8_t *ov, float *score,
                          const ppp_box &sb, void *work, const Geo &G, hipStream_t s) {
    if (!rank_vm_supported(G)) return hipErrorNotSupported;
    if (rank_wg_supported(G)) return launch_rank_wg(pred, dtype, S, ov, score, sb, work, G, s);
    return dtype == PPP_F16 ? launch_rv<__half>((const __half *)pred, S, ov, score, sb, work, G, s)
                            : launch_rv<float>((const float *)pred, S, ov, score, sb, work, G, s);
}

}  // namespace ppp
