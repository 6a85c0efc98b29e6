// ppp_cover.hip -- S3, the greedy foreground cover, as an EXACT priority-parallel algorithm.
//
// Reference: foreground_cover.py:111-180 (computeForegroundCoverLoop) walks the ranked patch
// list sequentially: a patch is selected iff more than pixTh voxels of the still uncovered mask
// lie inside its window where its prediction is > fc_threshold; those voxels are then cleared.
//
// The outcome for patch i only depends on the selected patches of HIGHER rank whose windows
// overlap win(i) (centres within p-1 per axis).  Hence rounds of
//   count   : h_i = |mask & win_i & bits_i| for every undecided patch; h_i <= pixTh can never
//             recover (the mask only shrinks) -> decided "not selected" at once;
//   ready   : an undecided patch is ready when no undecided patch of higher rank lies within
//             p-1 of it (3-d min filter over the volume of undecided ranks);
//   select  : every ready patch is selected and clears its voxels.  Ready patches never
//             overlap each other, and when i is decided every higher-ranked overlapping patch
//             is decided and no lower-ranked overlapping one has touched the mask, so win(i)
//             is exactly what the sequential loop would see at i's turn.
// reproduce the sequential result.  The loop's stop rule (it ends as soon as the interior of
// the mask is empty) is applied afterwards from the per-patch "cleared interior voxels"
// counts, in rank order (host side, foreground_cover.py driver).
//
// The undecided patches live in a volume (rank k at the patch centre), and every kernel of a
// round is a coalesced sweep over that volume: no lists, no compaction, no atomics on the data
// path.  A patch is only recounted when a patch selected in the previous round overlaps it
// (the selecting wave marks the (2p-1)^3 box of centres it can affect in a byte volume), the
// recount stops at the first window row that settles "more than pixTh", and the host only
// synchronises once per batch of rounds.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "ppp_kernels.hpp"

namespace ppp {

static constexpr int32_t RANK_NONE = 0x7F7F7F7F;
static constexpr int COVER_BATCH = 8;   // rounds per host synchronisation

struct CoverWork {
    int32_t *rank_vol, *nbr_min, *tmp;   // [V] each
    uint8_t *dirty;                      // [V]
    uint32_t *mbits;                     // [Z*Y][XW] running mask, one bit per voxel
    int32_t *counters;                   // [COVER_BATCH]
    int32_t *loc_vol;                    // [V] sharded cover only: index into the rank's own
                                         // state / cleared / bits tables, -1 off the own centres
    uint16_t *witness;                   // [V] pix_th == 0: ONE voxel of the window that the patch
                                         // centred here still covered at its last recount: window
                                         // row (dz * py + dy) << 5 | x offset (px <= 32: five bits; the row
                                         // keeps eleven, witness_ok); 0xFFFF = none known
};
static constexpr uint16_t WIT_NONE = 0xFFFFu;
// the 16-bit witness holds a window row in eleven bits (0x7FF | 31 would read as WIT_NONE); taller
// windows run without witnesses: dirty marks + windowed recounts, as the pix_th > 0 passes do
static inline bool witness_ok(const Geo &G) { return G.pz * G.py < 2047 && G.px <= 32; }

// words per row of the bit mask: one spare word so a window may be read as two words
__host__ __device__ __forceinline__ int row_words(const Geo &G) { return (G.X + 31) / 32 + 1; }

__device__ __forceinline__ void centre_of(const Geo &G, long long c, int &cz, int &cy, int &cx) {
    cx = (int)(c % G.X);
    cy = (int)((c / G.X) % G.Y);
    cz = (int)(c / ((long long)G.X * G.Y));
}

// n (<= 32) bits starting at bit `start` of a little-endian bit string held in 32-bit words;
// `limit` = number of readable words
__device__ __forceinline__ uint32_t bit_window(const uint32_t *w, int start, int n, int limit) {
    const int i = start >> 5, sh = start & 31;
    const uint32_t lo = w[i];
    const uint32_t hi = (i + 1 < limit) ? w[i + 1] : 0u;
    const uint32_t v = (uint32_t)((((unsigned long long)hi << 32) | lo) >> sh);
    return n >= 32 ? v : (v & ((1u << n) - 1u));
}

// byte mask -> bit mask (thread per word) and back (thread per voxel)
__global__ void __launch_bounds__(256)
    cover_pack_kernel(const uint8_t *__restrict__ mask, uint32_t *__restrict__ mbits, const Geo G) {
    const int XW = row_words(G);
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (t >= (long long)G.Z * G.Y * XW) return;
    const int w = (int)(t % XW);
    const long long row = t / XW;
    uint32_t v = 0;
    for (int i = 0; i < 32; ++i) {
        const int x = w * 32 + i;
        if (x < G.X && mask[row * G.X + x] != 0) v |= 1u << i;
    }
    mbits[t] = v;
}
__global__ void __launch_bounds__(256)
    cover_unpack_kernel(const uint32_t *__restrict__ mbits, uint8_t *__restrict__ mask, const Geo G) {
    const long long v = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (v >= G.V) return;
    const int x = (int)(v % G.X);
    const long long row = v / G.X;
    // keep the caller's byte values where the bit survived
    if (!((mbits[row * row_words(G) + (x >> 5)] >> (x & 31)) & 1u)) mask[v] = 0;
}

// state[k] == 0: undecided.  Enter every undecided patch into the rank volume.
__global__ void __launch_bounds__(256)
    cover_init_kernel(const long long *__restrict__ lin, const int32_t *__restrict__ state, int n,
                      int32_t *__restrict__ rank_vol) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && state[k] == 0) rank_vol[lin[k]] = k;
}

// Count step.  Thread per voxel; a workgroup of 1024 voxels first COMPACTS its undecided patches
// whose neighbourhood changed into a list in LDS (and consumes the dirty marks; "any patch still
// undecided" is flagged), then its first lanes recount them (per row of the window: mask bits AND
// patch bits, popcount) and reject those that can cover <= pix_th voxels.  Recounting inside the
// per-voxel sweep left one or two lanes of almost every wave walking the 49-row window while the
// others idled -- a few per cent of the voxels are dirty candidates, scattered.  (A global list
// costs a same-address atomic per wave: 4x slower than no compaction at all.)
static constexpr int COUNT_THREADS = 1024;
static constexpr int COUNT_VPT = 4;     // voxels per thread of the cover's count sweep
__global__ void __launch_bounds__(COUNT_THREADS)
    cover_count_kernel(const uint32_t *__restrict__ mbits, const uint32_t *__restrict__ bits,
                       uint8_t *__restrict__ dirty, const int pix_th, int32_t *__restrict__ state,
                       int32_t *__restrict__ rank_vol, int32_t *__restrict__ n_alive,
                       const int32_t *__restrict__ loc_vol, uint16_t *__restrict__ witness,
                       const long long bits_vox, const Geo G) {
    // COUNT_VPT voxels per thread (v = block base + j * 1024 + thread: every pass stays coalesced).  The
    // sweep is a chain of dependent loads per voxel -- rank, then witness, then one word of the mask --
    // and its time was their latency (1.17 ms for the 134 M voxels of 512^3 with one voxel per thread:
    // 0.8 TB/s); the loads of a thread's voxels are issued pass by pass, four in flight each.
    __shared__ uint16_t s_list[COUNT_THREADS * COUNT_VPT];
    __shared__ int s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const long long v0 = blockIdx.x * (long long)(COUNT_THREADS * COUNT_VPT);
    {
        const bool use_wit = pix_th == 0 && witness != nullptr;
        long long vv[COUNT_VPT];
        int kk[COUNT_VPT];
        bool alive[COUNT_VPT], marked[COUNT_VPT];
        unsigned wv[COUNT_VPT];
#pragma unroll
        for (int j = 0; j < COUNT_VPT; ++j) {
            vv[j] = v0 + j * COUNT_THREADS + threadIdx.x;
            kk[j] = vv[j] < G.V ? rank_vol[vv[j]] : RANK_NONE;
        }
        // sharded: rank_vol holds GLOBAL ranks (also of the neighbour's patches in the halo);
        // only the own centres are worked on, through their local table index
        if (loc_vol) {
#pragma unroll
            for (int j = 0; j < COUNT_VPT; ++j)
                if (kk[j] != RANK_NONE) { const int l = loc_vol[vv[j]]; kk[j] = l < 0 ? RANK_NONE : l; }
        }
        // pix_th == 0 ("does the patch still cover ANY voxel"): a patch stays undecided as long as
        // its witness voxel is uncovered -- one bit of the running mask (16 MB at 512^3: cache
        // resident) decides it, no patch bits are read and no dirty marks are needed; only a patch
        // whose witness was cleared (or that has none yet) is recounted.  On a dense volume a patch
        // is looked at in hundreds of rounds and nearly always survives.
#pragma unroll
        for (int j = 0; j < COUNT_VPT; ++j) {
            alive[j] = kk[j] != RANK_NONE;
            wv[j] = WIT_NONE;
            if (use_wit) { if (alive[j]) wv[j] = witness[vv[j]]; }
            else { marked[j] = vv[j] < G.V && dirty[vv[j]] != 0; if (marked[j]) dirty[vv[j]] = 0; }
        }
        if (use_wit) {
            uint32_t mw[COUNT_VPT];
            int xb[COUNT_VPT];
#pragma unroll
            for (int j = 0; j < COUNT_VPT; ++j) {
                mw[j] = 0u; xb[j] = 0;
                if (alive[j] && wv[j] != WIT_NONE) {
                    const int wr = (int)(wv[j] >> 5), xo = (int)(wv[j] & 0x1Fu);
                    const long long row = vv[j] / G.X + (long long)(wr / G.py - G.rz) * G.Y + (wr % G.py - G.ry);
                    const int x = (int)(vv[j] % G.X) - G.rx + xo;
                    mw[j] = mbits[row * row_words(G) + (x >> 5)];
                    xb[j] = x & 31;
                }
            }
#pragma unroll
            for (int j = 0; j < COUNT_VPT; ++j) marked[j] = alive[j] && ((mw[j] >> xb[j]) & 1u) == 0u;
        }
        bool rest = false;
#pragma unroll
        for (int j = 0; j < COUNT_VPT; ++j) {
            const unsigned long long m = __ballot(alive[j] && marked[j]);
            if (m != 0ull) {
                const int lane = threadIdx.x & 63;
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_n, __popcll(m));
                base = __shfl(base, 0);
                if (alive[j] && marked[j])
                    s_list[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(j * COUNT_THREADS + threadIdx.x);
            }
            rest = rest || (alive[j] && !marked[j]);
        }
        // "is any patch still undecided" AFTER this step: a plain store of the same value from
        // every wave that has one that is not recounted now (same-address atomics from ~V/64
        // waves would dominate the kernel); the recounted ones report below if they survive
        if (__ballot(rest) != 0 && (threadIdx.x & 63) == 0) *n_alive = 1;
    }
    __syncthreads();
    const int n = s_n;
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        const long long v = v0 + s_list[t];
        int k = rank_vol[v];
        if (loc_vol) k = loc_vol[v];
        const int words = (G.C + 31) / 32, XW = row_words(G);
        int cz, cy, cx;
        centre_of(G, v, cz, cy, cx);
        // (bits_vox >= 0: the table has a row per VOXEL, its first row belongs to voxel bits_vox)
        const uint32_t *b = bits + (bits_vox >= 0 ? v - bits_vox : (long long)k) * words;
        const int start = cx - G.rx, wi = start >> 5, sh = start & 31;
        const bool two = sh + G.px > 32;
        const uint32_t pmask = G.px >= 32 ? 0xFFFFFFFFu : ((1u << G.px) - 1u);
        const bool use_wit = pix_th == 0 && witness != nullptr;
        unsigned wit_new = WIT_NONE;      // first covered voxel found: row << 5 | x offset
        // sliding 64-bit window over the patch's bit string
        unsigned long long win = b[0] | ((unsigned long long)(words > 1 ? b[1] : 0u) << 32);
        int have = 64, next = 2, hits = 0;
        // only "more than pix_th" matters: stop at the first plane of rows that settles it
        // (with pix_th = 0 a surviving patch is usually done after the first plane).  The mask
        // words of a whole plane are loaded before any of them is used.
        constexpr int MAXPY = 9;
        for (int dz = 0; dz < G.pz && hits <= pix_th; ++dz) {
            const uint32_t *row = mbits + ((long long)(cz + dz - G.rz) * G.Y + (cy - G.ry)) * XW + wi;
            if (G.py <= MAXPY) {
                uint32_t lo[MAXPY], hi[MAXPY];
#pragma unroll
                for (int dy = 0; dy < MAXPY; ++dy) {
                    const bool in_p = dy < G.py;
                    lo[dy] = in_p ? row[(long long)dy * XW] : 0u;
                    hi[dy] = (in_p && two) ? row[(long long)dy * XW + 1] : 0u;
                }
#pragma unroll
                for (int dy = 0; dy < MAXPY; ++dy) {
                    if (dy < G.py) {
                        const unsigned long long mw = lo[dy] | ((unsigned long long)hi[dy] << 32);
                        const uint32_t hm = (uint32_t)(mw >> sh) & (uint32_t)win & pmask;
                        const int rh = __popc(hm);
                        if (rh && wit_new == WIT_NONE) wit_new = ((unsigned)(dz * G.py + dy) << 5) | (unsigned)__builtin_ctz(hm);
                        hits += rh;
                        win >>= G.px;
                        have -= G.px;
                        if (have <= 32) {
                            win |= (unsigned long long)(next < words ? b[next] : 0u) << have;
                            ++next;
                            have += 32;
                        }
                    }
                }
            } else {
                for (int dy = 0; dy < G.py; ++dy, row += XW) {
                    unsigned long long mw = row[0];
                    if (two) mw |= (unsigned long long)row[1] << 32;
                    const uint32_t hm = (uint32_t)(mw >> sh) & (uint32_t)win & pmask;
                    const int rh = __popc(hm);
                    if (rh && wit_new == WIT_NONE) wit_new = ((unsigned)(dz * G.py + dy) << 5) | (unsigned)__builtin_ctz(hm);
                    hits += rh;
                    win >>= G.px;
                    have -= G.px;
                    if (have <= 32) {
                        win |= (unsigned long long)(next < words ? b[next] : 0u) << have;
                        ++next;
                        have += 32;
                    }
                }
            }
        }
        if (hits <= pix_th) {
            state[k] = 2;
            rank_vol[v] = RANK_NONE;
        } else {
            if (use_wit) witness[v] = (uint16_t)wit_new;
            *n_alive = 1;
        }
    }
}

// "no patch here" of a filter volume: the largest value of the element type's byte pattern 0x7F..
template <typename T> struct FilterNone;
template <> struct FilterNone<int32_t> { static constexpr int32_t value = 0x7F7F7F7F; };
template <> struct FilterNone<long long> { static constexpr long long value = 0x7F7F7F7F7F7F7F7Fll; };

// 1-d running minimum of width 2*radius+1 along one axis (stride in elements, n along axis)
template <typename T>
__global__ void __launch_bounds__(256)
    cover_minfilter_kernel(const T *__restrict__ in, T *__restrict__ out,
                           const long long V, const int n, const long long stride, const int radius) {
    const long long v = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (v >= V) return;
    const int pos = (int)((v / stride) % n);
    T m = FilterNone<T>::value;
    for (int d = -radius; d <= radius; ++d) {
        const int q = pos + d;
        if (q >= 0 && q < n) m = min(m, in[v + (long long)d * stride]);
    }
    out[v] = m;
}

// x and y passes of the running minimum in one kernel: a block owns 8 rows x 64 columns of one
// z-slice, stages them with their (ry, rx) apron in LDS, takes the minimum along x into a second
// LDS array and then along y (one launch and one volume round trip less per round).
// (rows per block by element size: 32 rows of int32 / 16 of int64 -- with the (ry, rx) apron a block
// reads (TY + 2 ry)(64 + 2 rx) elements for TY * 64 results: 1.9 x at 32 rows, 3.75 x at 8)
static constexpr int MF_TX = 64;
template <typename T> struct MfRows { static constexpr int value = sizeof(T) == 4 ? 32 : 16; };
template <typename T, int MF_TY>
__global__ void __launch_bounds__(256)
    cover_minfilter_xy_kernel(const T *__restrict__ in, T *__restrict__ out, const Geo G,
                              const int rx, const int ry) {
    extern __shared__ long long mf_lds_raw[];
    const T RANK_NONE = FilterNone<T>::value;
    T *mf_lds = reinterpret_cast<T *>(mf_lds_raw);
    const int W = MF_TX + 2 * rx, H = MF_TY + 2 * ry;
    T *a = mf_lds;            // [H][W]   input tile with apron
    T *b = mf_lds + H * W;    // [H][MF_TX] minimum along x
    const int x0 = blockIdx.x * MF_TX, y0 = blockIdx.y * MF_TY, z = blockIdx.z;
    const long long zbase = (long long)z * G.Y * G.X;
    for (int i = threadIdx.x; i < H * W; i += 256) {
        const int yy = y0 - ry + i / W, xx = x0 - rx + i % W;
        a[i] = (yy >= 0 && yy < G.Y && xx >= 0 && xx < G.X) ? in[zbase + (long long)yy * G.X + xx] : RANK_NONE;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < H * MF_TX; i += 256) {
        const int r = i / MF_TX, c = i % MF_TX;
        T m = RANK_NONE;
        for (int d = 0; d <= 2 * rx; ++d) m = min(m, a[r * W + c + d]);
        b[i] = m;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MF_TY * MF_TX; i += 256) {
        const int r = i / MF_TX, c = i % MF_TX;
        const int yy = y0 + r, xx = x0 + c;
        if (yy >= G.Y || xx >= G.X) continue;
        T m = RANK_NONE;
        for (int d = 0; d <= 2 * ry; ++d) m = min(m, b[(r + d) * MF_TX + c]);
        out[zbase + (long long)yy * G.X + xx] = m;
    }
}

// x and y passes of the neighbourhood minimum: in -> out (scratch: the x pass when the fused
// kernel's tile does not fit LDS).  The z pass is NOT a sweep of its own: only the few voxels that
// hold an undecided patch need the 3-d minimum, and the select kernels take it over the 2 pz - 1
// slices themselves (one volume write + read and one launch less per round).
template <typename T>
static void minfilter_xy(const T *in, T *scratch, T *out, const Geo &G, hipStream_t s) {
    const int rx = G.px - 1, ry = G.py - 1;
    constexpr int TYB = MfRows<T>::value, TYS = 8;      // rows per block: the tall tile where it fits LDS
    auto lds_of = [&](int ty) { return (size_t)((ty + 2 * ry) * (MF_TX + 2 * rx) + (ty + 2 * ry) * MF_TX) * sizeof(T); };
    auto grid_of = [&](int ty) { return dim3((unsigned)((G.X + MF_TX - 1) / MF_TX), (unsigned)((G.Y + ty - 1) / ty), (unsigned)G.Z); };
    const dim3 vgrid((unsigned)((G.V + 255) / 256)), block(256);
    if (lds_of(TYB) <= 48 * 1024 && G.Z <= 65535) {
        cover_minfilter_xy_kernel<T, TYB><<<grid_of(TYB), block, lds_of(TYB), s>>>(in, out, G, rx, ry);
    } else if (lds_of(TYS) <= 48 * 1024 && G.Y <= 65535 * TYS && G.Z <= 65535) {
        cover_minfilter_xy_kernel<T, TYS><<<grid_of(TYS), block, lds_of(TYS), s>>>(in, out, G, rx, ry);
    } else {
        cover_minfilter_kernel<T><<<vgrid, block, 0, s>>>(in, scratch, G.V, G.X, 1, rx);
        cover_minfilter_kernel<T><<<vgrid, block, 0, s>>>(scratch, out, G.V, G.Y, G.X, ry);
    }
}
// minimum over the slices z - (pz-1) .. z + (pz-1) of the xy-filtered volume at voxel v
template <typename T>
__device__ __forceinline__ T zmin_at(const T *__restrict__ xy, long long v, const Geo &G) {
    const long long plane = (long long)G.X * G.Y;
    const int z = (int)(v / plane);
    T m = FilterNone<T>::value;
    for (int d = -(G.pz - 1); d <= G.pz - 1; ++d)
        if (z + d >= 0 && z + d < G.Z) m = min(m, xy[v + (long long)d * plane]);
    return m;
}

// "is the value `mine` at voxel v the minimum over the slices z - (pz-1) .. z + (pz-1) of the
// xy-filtered volume": the same answer as zmin_at(...) == mine, but the slices are asked nearest
// first and a wave stops as soon as every candidate lane has met a smaller value -- on a dense
// volume a patch has a better ranked undecided neighbour in its own slice in all but a few
// thousand cases per round, so a wave reads ONE slice instead of 2 pz - 1.
template <typename T>
__device__ __forceinline__ bool is_zmin(const T *__restrict__ xy, long long v, T mine, bool cand, const Geo &G) {
    const long long plane = (long long)G.X * G.Y;
    const int z = cand ? (int)(v / plane) : 0;
    for (int i = 0; i <= 2 * (G.pz - 1); ++i) {
        if (__ballot(cand) == 0ull) break;
        const int d = (i & 1) ? -((i + 1) >> 1) : (i >> 1);            // 0, -1, +1, -2, +2, ...
        if (cand && z + d >= 0 && z + d < G.Z && xy[v + (long long)d * plane] < mine) cand = false;
    }
    return cand;
}

// Thread per voxel: the best ranked undecided patch of its neighbourhood selects itself; its
// wave clears the voxels (lane per window row) and marks the centres whose counts may have
// changed.  Selected patches never share a voxel, but they may share a mask word.
__global__ void __launch_bounds__(256)
    cover_select_kernel(uint32_t *__restrict__ mbits, const uint32_t *__restrict__ bits,
                        const int32_t *__restrict__ nbr_min, int32_t *__restrict__ state,
                        int32_t *__restrict__ rank_vol, int32_t *__restrict__ cleared_interior,
                        uint8_t *__restrict__ dirty, const int32_t *__restrict__ loc_vol,
                        const int gZ, const long long bits_vox, const int mark_dirty, const Geo G) {
    const long long v = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int k = v < G.V ? rank_vol[v] : RANK_NONE;
    bool ready = is_zmin<int32_t>(nbr_min, v, k, k != RANK_NONE, G);          // nbr_min: xy-filtered ranks
    if (loc_vol && ready) { k = loc_vol[v]; ready = k >= 0; }   // own centres only; local index
    unsigned long long todo = __ballot(ready);
    const int words = (G.C + 31) / 32, XW = row_words(G);
    while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int kk = __shfl(k, src);
        const long long cc = v - lane + src;
        int cz, cy, cx;
        centre_of(G, cc, cz, cy, cx);
        const uint32_t *b = bits + (bits_vox >= 0 ? cc - bits_vox : (long long)kk) * words;
        const int start = cx - G.rx, sh = start & 31;
        // window bits whose voxel is an interior x position
        uint32_t xin = 0;
        for (int i = 0; i < G.px; ++i)
            if (start + i >= G.rx && start + i < G.X - G.rx) xin |= 1u << i;
        int cleared = 0;
        for (int r = lane; r < G.pz * G.py; r += 64) {
            const int z = cz + r / G.py - G.rz, y = cy + r % G.py - G.ry;
            uint32_t *row = mbits + ((long long)z * G.Y + y) * XW;
            const uint32_t cl = bit_window(row, start, G.px, XW) & bit_window(b, r * G.px, G.px, words);
            if (cl) {
                atomicAnd(row + (start >> 5), ~(cl << sh));
                if (sh && (cl >> (32 - sh))) atomicAnd(row + (start >> 5) + 1, ~(cl >> (32 - sh)));
                // (interior of the WHOLE volume: gZ slices, this buffer starts at slice G.oz)
                if (z + G.oz >= G.rz && z + G.oz < gZ - G.rz && y >= G.ry && y < G.Y - G.ry)
                    cleared += __popc(cl & xin);
            }
        }
        for (int o = 32; o > 0; o >>= 1) cleared += __shfl_xor(cleared, o);
        // every centre within p-1 of this one has a window that overlaps the cleared voxels
        const int z0 = max(cz - (G.pz - 1), 0), z1 = min(cz + G.pz - 1, G.Z - 1);
        const int y0 = max(cy - (G.py - 1), 0), y1 = min(cy + G.py - 1, G.Y - 1);
        const int x0 = max(cx - (G.px - 1), 0), x1 = min(cx + G.px - 1, G.X - 1);
        const int ny = y1 - y0 + 1, nx = x1 - x0 + 1;
        // (pix_th == 0: the count step asks every undecided patch's witness voxel instead)
        const int rows = mark_dirty ? (z1 - z0 + 1) * ny : 0;
        for (int row = lane; row < rows; row += 64) {
            uint8_t *d = dirty + vox(G, z0 + row / ny, y0 + row % ny, x0);
            for (int x = 0; x < nx; ++x) d[x] = 1;
        }
        if (lane == src) {
            state[kk] = 1;
            rank_vol[cc] = RANK_NONE;
            cleared_interior[kk] = cleared;
        }
    }
}

static size_t up256(size_t v) { return (v + 255) / 256 * 256; }

size_t cover_workspace_bytes(long long n, const Geo &G) {
    (void)n;
    return 4 * up256((size_t)G.V * 4) + up256((size_t)G.V) + up256((size_t)G.V * 2) +
           up256((size_t)G.Z * G.Y * row_words(G) * 4) + 256;
}

static CoverWork carve(void *work, const Geo &G) {
    CoverWork W;
    char *p = (char *)work;
    W.rank_vol = (int32_t *)p; p += up256((size_t)G.V * 4);
    W.nbr_min = (int32_t *)p;  p += up256((size_t)G.V * 4);
    W.tmp = (int32_t *)p;      p += up256((size_t)G.V * 4);
    W.dirty = (uint8_t *)p;    p += up256((size_t)G.V);
    W.mbits = (uint32_t *)p;   p += up256((size_t)G.Z * G.Y * row_words(G) * 4);
    W.counters = (int32_t *)p; p += 256;
    W.loc_vol = (int32_t *)p;  p += up256((size_t)G.V * 4);
    W.witness = (uint16_t *)p;
    return W;
}

// One pass of the cover loop without the stop rule.  Returns the number of rounds in *rounds.
hipError_t run_cover_pass(uint8_t *mask, const uint32_t *bits, long long bits_vox, const long long *lin,
                          long long n, int pix_th, int32_t *state, int32_t *cleared, void *work,
                          const Geo &G, hipStream_t s, int *rounds) {
    *rounds = 0;
    if (n <= 0) return hipSuccess;
    PPP_GRID_CHECK((G.V + 255) / 256, 256);
    CoverWork W = carve(work, G);
    hipError_t e;
    if ((e = hipMemsetD32Async((hipDeviceptr_t)W.rank_vol, RANK_NONE, (size_t)G.V, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(W.dirty, 1, (size_t)G.V, s)) != hipSuccess) return e;   // count everything once
    if ((e = hipMemsetAsync(W.witness, 0xFF, (size_t)G.V * 2, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(cleared, 0, (size_t)n * 4, s)) != hipSuccess) return e;
    const dim3 vgrid((unsigned)((G.V + 255) / 256)), block(256);
    const long long n_words = (long long)G.Z * G.Y * row_words(G);
    cover_pack_kernel<<<dim3((unsigned)((n_words + 255) / 256)), block, 0, s>>>(mask, W.mbits, G);
    cover_init_kernel<<<dim3((unsigned)((n + 255) / 256)), block, 0, s>>>(lin, state, (int)n, W.rank_vol);
    int32_t n_alive = 1;
    while (n_alive > 0) {
        if ((e = hipMemsetAsync(W.counters, 0, COVER_BATCH * 4, s)) != hipSuccess) return e;
        const dim3 cgrid((unsigned)((G.V + COUNT_THREADS * COUNT_VPT - 1) / (COUNT_THREADS * COUNT_VPT))), cblock(COUNT_THREADS);
        for (int r = 0; r < COVER_BATCH; ++r) {
            cover_count_kernel<<<cgrid, cblock, 0, s>>>(W.mbits, bits, W.dirty, pix_th, state, W.rank_vol,
                                                       W.counters + r, nullptr, witness_ok(G) ? W.witness : nullptr, bits_vox, G);
            // x, y, z; radius p-1: two windows overlap iff |dc| <= p-1 on every axis
            minfilter_xy<int32_t>(W.rank_vol, W.tmp, W.nbr_min, G, s);
            cover_select_kernel<<<vgrid, block, 0, s>>>(W.mbits, bits, W.nbr_min, state, W.rank_vol, cleared,
                                                        W.dirty, nullptr, G.Z + G.oz, bits_vox, (pix_th != 0 || !witness_ok(G)) ? 1 : 0, G);
        }
        *rounds += COVER_BATCH;
        // "any patch undecided" at the start of the batch's last round; if none, that round
        // was a no-op and nothing is left
        if ((e = hipMemcpyAsync(&n_alive, W.counters + COVER_BATCH - 1, 4, hipMemcpyDeviceToHost, s)) != hipSuccess)
            return e;
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
        if ((e = hipGetLastError()) != hipSuccess) return e;
        static EnvSwitch trace("PPP_COVER_TRACE");
        if (trace.get()) {   // development aid: undecided / selected patches per batch
            std::vector<int32_t> h((size_t)n);
            (void)hipMemcpy(h.data(), state, (size_t)n * 4, hipMemcpyDeviceToHost);
            long long a = 0, sel = 0;
            for (long long i = 0; i < n; ++i) { a += h[i] == 0; sel += h[i] == 1; }
            fprintf(stderr, "cover rounds %d: undecided %lld selected %lld\n", *rounds, a, sel);
        }
    }
    cover_unpack_kernel<<<vgrid, block, 0, s>>>(W.mbits, mask, G);
    return hipGetLastError();
}

// ---- S4: set-cover thinning of the selected patches, on the device --------------------------
//
// Reference: foreground_cover.py:183-256 (thinOutForegroundCover, sample == 1.0).  Its loop picks,
// while the interior of the running mask is not empty, the FIRST patch with the largest number
// of still uncovered voxels (np.argmax over len(set_i), sets = window & (pred > fc_threshold) &
// running mask), keeps it and clears its voxels.  That is a greedy cover with the DYNAMIC
// priority key_i = (count_i descending, index_i ascending).
//
// Priority-parallel form (exact).  Counts only shrink, so the best key of the sequential loop
// never improves over time.  A patch whose key beats the key of every undecided patch within
// p-1 of it (the only patches whose windows intersect its own) can not be overtaken: none of
// them can become the global best while it is undecided with an unchanged count, and its count
// only changes when one of them is kept.  So it is kept by the sequential loop with exactly its
// current count, and all such local winners of a round may be kept at once (they never overlap:
// keys are distinct).  The sequential ORDER of the kept patches is the order of their keys at
// the time they were kept (strictly worsening along the loop), so the loop's stop rule ("interior
// empty", tested before every pick) is applied afterwards: sort the kept patches by (count
// descending, index ascending), cut after the one at which the cumulative number of cleared
// interior voxels reaches the initial interior count.  A patch with count 0 is never a winner;
// when every count is 0 but the interior is not empty the reference's argmax returns patch 0 and
// its empty index tuple zeroes the whole mask (:210-216), which ends the loop: keep[0] = 1.
//
// Kernels: same volume sweeps as the cover above; the rank volume becomes a 64-bit key volume
//   key = (THIN_MAXC - count) << 32 | index       (smaller is better; NONE = 0x7F7F..)
// and the ready test is "my key is the minimum of my (p-1)-neighbourhood".
static constexpr long long THIN_NONE = FilterNone<long long>::value;
static constexpr long long THIN_MAXC = 1ll << 20;

struct ThinWork {
    long long *key_vol, *nbr_min, *tmp;  // [V] each
    uint8_t *dirty;                      // [V]
    uint32_t *mbits;                     // [Z*Y][XW]
    int32_t *counters;                   // [COVER_BATCH]
    unsigned long long *interior;        // [1] set interior voxels of the mask
    int32_t *state, *sel_count, *cleared;  // [n] each
};

size_t thin_workspace_bytes(long long n, const Geo &G) {
    return 3 * up256((size_t)G.V * 8) + up256((size_t)G.V) +
           up256((size_t)G.Z * G.Y * row_words(G) * 4) + 256 + 256 + 3 * up256((size_t)(n > 0 ? n : 1) * 4);
}

static ThinWork carve_thin(void *work, long long n, const Geo &G) {
    ThinWork W;
    char *p = (char *)work;
    W.key_vol = (long long *)p; p += up256((size_t)G.V * 8);
    W.nbr_min = (long long *)p; p += up256((size_t)G.V * 8);
    W.tmp = (long long *)p;     p += up256((size_t)G.V * 8);
    W.dirty = (uint8_t *)p;     p += up256((size_t)G.V);
    W.mbits = (uint32_t *)p;    p += up256((size_t)G.Z * G.Y * row_words(G) * 4);
    W.counters = (int32_t *)p;  p += 256;
    W.interior = (unsigned long long *)p; p += 256;
    const size_t per = up256((size_t)(n > 0 ? n : 1) * 4);
    W.state = (int32_t *)p;     p += per;
    W.sel_count = (int32_t *)p; p += per;
    W.cleared = (int32_t *)p;
    return W;
}

__global__ void __launch_bounds__(256)
    thin_init_kernel(const long long *__restrict__ lin, int n, long long *__restrict__ key_vol) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    // (count not known yet.)  A centre listed twice -- possible in a list from outside, never in
    // one made by the greedy cover -- keeps its FIRST index, like the reference's np.argmax: the
    // later copy has the same voxel set, which is empty once the first is kept.
    if (k < n) atomicMin(&key_vol[lin[k]], (THIN_MAXC << 32) | (long long)k);
}
// ... and the later copies are retired at once
__global__ void __launch_bounds__(256)
    thin_dups_kernel(const long long *__restrict__ lin, int n, const long long *__restrict__ key_vol,
                     int32_t *__restrict__ state) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && (int)(key_vol[lin[k]] & 0xFFFFFFFFll) != k) state[k] = 2;
}

// set voxels of the bit mask that are interior voxels of the volume (thread per mask word)
__global__ void __launch_bounds__(256)
    thin_interior_kernel(const uint32_t *__restrict__ mbits, unsigned long long *__restrict__ total,
                         const Geo G) {
    const int XW = row_words(G);
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    int c = 0;
    if (t < (long long)G.Z * G.Y * XW) {
        const int w = (int)(t % XW);
        const long long row = t / XW;
        const int y = (int)(row % G.Y), z = (int)(row / G.Y);
        if (z >= G.rz && z < G.Z - G.rz && y >= G.ry && y < G.Y - G.ry) {
            const int lo = max(G.rx - w * 32, 0), hi = min(G.X - G.rx - w * 32, 32);   // bits [lo, hi)
            if (lo < hi) {
                const uint32_t m = (hi >= 32 ? 0xFFFFFFFFu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
                c = __popc(mbits[t] & m);
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    __shared__ int part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int sum = part[0] + part[1] + part[2] + part[3];
        if (sum) atomicAdd(total, (unsigned long long)sum);
    }
}

// Thread per voxel: the undecided patches of a workgroup whose neighbourhood changed are compacted
// in LDS (as in cover_count_kernel), then recounted in full; a patch that covers nothing any more
// is retired.
__global__ void __launch_bounds__(COUNT_THREADS)
    thin_count_kernel(const uint32_t *__restrict__ mbits, const uint32_t *__restrict__ bits,
                      uint8_t *__restrict__ dirty, int32_t *__restrict__ state,
                      long long *__restrict__ key_vol, int32_t *__restrict__ n_alive,
                      const int32_t *__restrict__ loc_vol, const Geo G) {
    __shared__ uint16_t s_list[COUNT_THREADS];
    __shared__ int s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const long long v0 = blockIdx.x * (long long)blockDim.x;
    {
        const long long v = v0 + threadIdx.x;
        const bool in = v < G.V;
        // (sharded: loc_vol = local list index of an OWN centre, -1 elsewhere -- the keys a rank sees in
        // its halo slices belong to patches their owner counts)
        const bool alive = in && key_vol[v] != THIN_NONE && (!loc_vol || loc_vol[v] >= 0);
        const bool marked = in && dirty[v] != 0;
        if (marked) dirty[v] = 0;
        const unsigned long long m = __ballot(alive && marked);
        if (m != 0ull) {
            const int lane = threadIdx.x & 63;
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_n, __popcll(m));
            base = __shfl(base, 0);
            if (alive && marked) s_list[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)threadIdx.x;
        }
        // undecided after this step: those not recounted now; recounted survivors report below
        if (__ballot(alive && !marked) != 0 && (threadIdx.x & 63) == 0) *n_alive = 1;
    }
    __syncthreads();
    const int n = s_n;
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        const long long v = v0 + s_list[t];
        const int k = (int)(key_vol[v] & 0xFFFFFFFFll);          // position in the (global) list: the key's tie-break
        const int kl = loc_vol ? loc_vol[v] : k;                 // row of `bits` / `state`
        const int words = (G.C + 31) / 32, XW = row_words(G);
        int cz, cy, cx;
        centre_of(G, v, cz, cy, cx);
        const uint32_t *b = bits + (long long)kl * words;
        const int start = cx - G.rx, wi = start >> 5, sh = start & 31;
        const bool two = sh + G.px > 32;
        const uint32_t pmask = G.px >= 32 ? 0xFFFFFFFFu : ((1u << G.px) - 1u);
        unsigned long long win = b[0] | ((unsigned long long)(words > 1 ? b[1] : 0u) << 32);
        int have = 64, next = 2, hits = 0;
        for (int dz = 0; dz < G.pz; ++dz) {
            const uint32_t *row = mbits + ((long long)(cz + dz - G.rz) * G.Y + (cy - G.ry)) * XW + wi;
            for (int dy = 0; dy < G.py; ++dy, row += XW) {
                unsigned long long mw = row[0];
                if (two) mw |= (unsigned long long)row[1] << 32;
                hits += __popc((uint32_t)(mw >> sh) & (uint32_t)win & pmask);
                win >>= G.px;
                have -= G.px;
                if (have <= 32) {
                    win |= (unsigned long long)(next < words ? b[next] : 0u) << have;
                    ++next;
                    have += 32;
                }
            }
        }
        if (hits == 0) {
            state[kl] = 2;
            key_vol[v] = THIN_NONE;
        } else {
            key_vol[v] = ((THIN_MAXC - (long long)hits) << 32) | (long long)k;
            *n_alive = 1;
        }
    }
}

// Thread per voxel: the patch with the best key of its neighbourhood keeps itself; its wave
// clears the voxels and marks the centres whose counts may have changed.
__global__ void __launch_bounds__(256)
    thin_select_kernel(uint32_t *__restrict__ mbits, const uint32_t *__restrict__ bits,
                       const long long *__restrict__ nbr_min, int32_t *__restrict__ state,
                       long long *__restrict__ key_vol, int32_t *__restrict__ sel_count,
                       int32_t *__restrict__ cleared_interior, uint8_t *__restrict__ dirty,
                       const int32_t *__restrict__ loc_vol, const int oz, const int gZ, const Geo G) {
    const long long v = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const long long key = v < G.V ? key_vol[v] : THIN_NONE;
    bool ready = is_zmin<long long>(nbr_min, v, key, key != THIN_NONE, G);      // (xy-filtered keys)
    int k = (int)(key & 0xFFFFFFFFll);
    if (loc_vol && ready) { k = loc_vol[v]; ready = k >= 0; }                   // own centres only; local index
    unsigned long long todo = __ballot(ready);
    const int words = (G.C + 31) / 32, XW = row_words(G);
    while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int kk = __shfl(k, src);
        const long long cc = v - lane + src;
        int cz, cy, cx;
        centre_of(G, cc, cz, cy, cx);
        const uint32_t *b = bits + (long long)kk * words;
        const int start = cx - G.rx, sh = start & 31;
        uint32_t xin = 0;
        for (int i = 0; i < G.px; ++i)
            if (start + i >= G.rx && start + i < G.X - G.rx) xin |= 1u << i;
        int cleared = 0;
        for (int r = lane; r < G.pz * G.py; r += 64) {
            const int z = cz + r / G.py - G.rz, y = cy + r % G.py - G.ry;
            uint32_t *row = mbits + ((long long)z * G.Y + y) * XW;
            const uint32_t cl = bit_window(row, start, G.px, XW) & bit_window(b, r * G.px, G.px, words);
            if (cl) {
                atomicAnd(row + (start >> 5), ~(cl << sh));
                if (sh && (cl >> (32 - sh))) atomicAnd(row + (start >> 5) + 1, ~(cl >> (32 - sh)));
                // (interior of the WHOLE volume: gZ slices, this buffer starts at its slice oz)
                if (z + oz >= G.rz && z + oz < gZ - G.rz && y >= G.ry && y < G.Y - G.ry)
                    cleared += __popc(cl & xin);
            }
        }
        for (int o = 32; o > 0; o >>= 1) cleared += __shfl_xor(cleared, o);
        const int z0 = max(cz - (G.pz - 1), 0), z1 = min(cz + G.pz - 1, G.Z - 1);
        const int y0 = max(cy - (G.py - 1), 0), y1 = min(cy + G.py - 1, G.Y - 1);
        const int x0 = max(cx - (G.px - 1), 0), x1 = min(cx + G.px - 1, G.X - 1);
        const int ny = y1 - y0 + 1, nx = x1 - x0 + 1;
        const int rows = (z1 - z0 + 1) * ny;
        for (int row = lane; row < rows; row += 64) {
            uint8_t *d = dirty + vox(G, z0 + row / ny, y0 + row % ny, x0);
            for (int x = 0; x < nx; ++x) d[x] = 1;
        }
        if (lane == src) {
            state[kk] = 1;
            key_vol[cc] = THIN_NONE;
            sel_count[kk] = (int32_t)(THIN_MAXC - (key >> 32));
            cleared_interior[kk] = cleared;
        }
    }
}

// keep u8 [n] (device).  Synchronises the stream.
hipError_t run_thin_cover(const uint8_t *mask, const uint32_t *bits, const long long *lin, long long n,
                          uint8_t *keep, void *work, const Geo &G, hipStream_t s, int *rounds) {
    *rounds = 0;
    if (n <= 0) return hipSuccess;
    if (n >= (1ll << 31) || G.C >= THIN_MAXC) return hipErrorInvalidValue;
    PPP_GRID_CHECK((G.V + 255) / 256, 256);
    ThinWork W = carve_thin(work, n, G);
    hipError_t e;
    if ((e = hipMemsetD32Async((hipDeviceptr_t)W.key_vol, 0x7F7F7F7F, (size_t)G.V * 2, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(W.dirty, 1, (size_t)G.V, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(W.state, 0, (size_t)n * 4, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(W.sel_count, 0, (size_t)n * 4, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(W.cleared, 0, (size_t)n * 4, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(W.interior, 0, 8, s)) != hipSuccess) return e;
    const dim3 vgrid((unsigned)((G.V + 255) / 256)), block(256);
    const long long n_words = (long long)G.Z * G.Y * row_words(G);
    cover_pack_kernel<<<dim3((unsigned)((n_words + 255) / 256)), block, 0, s>>>(mask, W.mbits, G);
    thin_interior_kernel<<<dim3((unsigned)((n_words + 255) / 256)), block, 0, s>>>(W.mbits, W.interior, G);
    thin_init_kernel<<<dim3((unsigned)((n + 255) / 256)), block, 0, s>>>(lin, (int)n, W.key_vol);
    thin_dups_kernel<<<dim3((unsigned)((n + 255) / 256)), block, 0, s>>>(lin, (int)n, W.key_vol, W.state);
    int32_t n_alive = 1;
    while (n_alive > 0) {
        if ((e = hipMemsetAsync(W.counters, 0, COVER_BATCH * 4, s)) != hipSuccess) return e;
        for (int r = 0; r < COVER_BATCH; ++r) {
            thin_count_kernel<<<dim3((unsigned)((G.V + COUNT_THREADS - 1) / COUNT_THREADS)), dim3(COUNT_THREADS), 0, s>>>(
                W.mbits, bits, W.dirty, W.state, W.key_vol, W.counters + r, nullptr, G);
            minfilter_xy<long long>(W.key_vol, W.tmp, W.nbr_min, G, s);
            thin_select_kernel<<<vgrid, block, 0, s>>>(W.mbits, bits, W.nbr_min, W.state, W.key_vol,
                                                       W.sel_count, W.cleared, W.dirty, nullptr, 0, G.Z, G);
        }
        *rounds += COVER_BATCH;
        if ((e = hipMemcpyAsync(&n_alive, W.counters + COVER_BATCH - 1, 4, hipMemcpyDeviceToHost, s)) != hipSuccess)
            return e;
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    // ---- the stop rule: kept patches in the order the sequential loop picks them
    std::vector<int32_t> st((size_t)n), cnt((size_t)n), clr((size_t)n);
    unsigned long long interior = 0;
    if ((e = hipMemcpyAsync(st.data(), W.state, (size_t)n * 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(cnt.data(), W.sel_count, (size_t)n * 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(clr.data(), W.cleared, (size_t)n * 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(&interior, W.interior, 8, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
    std::vector<long long> order;
    for (long long i = 0; i < n; ++i)
        if (st[(size_t)i] == 1) order.push_back(((THIN_MAXC - (long long)cnt[(size_t)i]) << 32) | i);
    std::sort(order.begin(), order.end());
    std::vector<uint8_t> k8((size_t)n, 0);
    long long remaining = (long long)interior;
    for (size_t j = 0; j < order.size() && remaining > 0; ++j) {
        const long long i = order[j] & 0xFFFFFFFFll;
        k8[(size_t)i] = 1;
        remaining -= clr[(size_t)i];
    }
    if (remaining > 0) k8[0] = 1;   // every count is 0 with voxels left: np.argmax picks patch 0
    if ((e = hipMemcpyAsync(keep, k8.data(), (size_t)n, hipMemcpyHostToDevice, s)) != hipSuccess) return e;
    return hipStreamSynchronize(s);
}

// ---- the same rounds, one step at a time, on the z-range of one rank (sharded cover) -------
//
// Every rank keeps the round state (rank volume, bit mask, dirty marks) for its own slices
// plus a halo of p-1 slices and works on its OWN patch centres only.  What a round needs from
// the neighbours lives in the "zones" of 2(p-1) slices around every slab boundary; they are
// made consistent twice per round by the caller (all-reduces of small buffers):
//   after count  : rank volume -- every slice is owned by one rank, the others contribute
//                  INT_MAX, MIN combines;
//   after select : mask (a voxel stays set only if nobody cleared it: MIN of 0/1 bytes) and
//                  dirty marks (kept inverted so that MIN combines them as well).
// Ready patches of two ranks never overlap (each sees the other's rank in its halo), so the
// combined mask is exactly what one device would hold.
__global__ void __launch_bounds__(256)
    cover_init_local_kernel(const long long *__restrict__ lin, const int32_t *__restrict__ rankid,
                            const int32_t *__restrict__ state, int n, int32_t *__restrict__ rank_vol,
                            int32_t *__restrict__ loc_vol) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    loc_vol[lin[i]] = i;
    if (state[i] == 0) rank_vol[lin[i]] = rankid[i];
}

hipError_t cover_open(const uint8_t *mask, const long long *lin, const int32_t *rankid, long long n,
                      const int32_t *state, int32_t *cleared, void *work, const Geo &G, hipStream_t s) {
    PPP_GRID_CHECK((G.V + 255) / 256, 256);
    CoverWork W = carve(work, G);
    hipError_t e;
    if ((e = hipMemsetD32Async((hipDeviceptr_t)W.rank_vol, RANK_NONE, (size_t)G.V, s)) != hipSuccess) return e;
    if ((e = hipMemsetD32Async((hipDeviceptr_t)W.loc_vol, 0xFFFFFFFF, (size_t)G.V, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(W.dirty, 1, (size_t)G.V, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(W.witness, 0xFF, (size_t)G.V * 2, s)) != hipSuccess) return e;
    if (n && (e = hipMemsetAsync(cleared, 0, (size_t)n * 4, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(W.counters, 0, COVER_BATCH * 4, s)) != hipSuccess) return e;
    const dim3 block(256);
    const long long n_words = (long long)G.Z * G.Y * row_words(G);
    cover_pack_kernel<<<dim3((unsigned)((n_words + 255) / 256)), block, 0, s>>>(mask, W.mbits, G);
    if (n)
        cover_init_local_kernel<<<dim3((unsigned)((n + 255) / 256)), block, 0, s>>>(
            lin, rankid, state, (int)n, W.rank_vol, W.loc_vol);
    return hipGetLastError();
}

hipError_t cover_step_count(const uint32_t *bits, int pix_th, int32_t *state, void *work,
                            const Geo &G, hipStream_t s) {
    CoverWork W = carve(work, G);
    hipError_t e;
    if ((e = hipMemsetAsync(W.counters, 0, 4, s)) != hipSuccess) return e;
    cover_count_kernel<<<dim3((unsigned)((G.V + COUNT_THREADS * COUNT_VPT - 1) / (COUNT_THREADS * COUNT_VPT))), dim3(COUNT_THREADS), 0, s>>>(
        W.mbits, bits, W.dirty, pix_th, state, W.rank_vol, W.counters, W.loc_vol, witness_ok(G) ? W.witness : nullptr, -1ll, G);
    return hipGetLastError();
}

hipError_t cover_step_filter(void *work, const Geo &G, hipStream_t s) {
    CoverWork W = carve(work, G);
    minfilter_xy<int32_t>(W.rank_vol, W.tmp, W.nbr_min, G, s);
    return hipGetLastError();
}

hipError_t cover_step_select(const uint32_t *bits, int pix_th, int32_t *state, int32_t *cleared, void *work,
                             int gZ, const Geo &G, hipStream_t s) {
    CoverWork W = carve(work, G);
    cover_select_kernel<<<dim3((unsigned)((G.V + 255) / 256)), dim3(256), 0, s>>>(
        W.mbits, bits, W.nbr_min, state, W.rank_vol, cleared, W.dirty, W.loc_vol, gZ, -1ll, (pix_th != 0 || !witness_ok(G)) ? 1 : 0, G);
    return hipGetLastError();
}

hipError_t cover_alive(void *work, const Geo &G, int32_t *alive, hipStream_t s) {
    CoverWork W = carve(work, G);
    hipError_t e;
    if ((e = hipMemcpyAsync(alive, W.counters, 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    return hipStreamSynchronize(s);
}

hipError_t cover_close(uint8_t *mask, void *work, const Geo &G, hipStream_t s) {
    CoverWork W = carve(work, G);
    cover_unpack_kernel<<<dim3((unsigned)((G.V + 255) / 256)), dim3(256), 0, s>>>(W.mbits, mask, G);
    return hipGetLastError();
}

// zone = local slices [z_lo, z_hi); the rank owns the local slices [own_lo, own_hi)
__global__ void __launch_bounds__(256)
    cover_zone_export_kernel(const int32_t *__restrict__ rank_vol, const uint32_t *__restrict__ mbits,
                             const uint8_t *__restrict__ dirty, const int z_lo, const int z_hi,
                             const int own_lo, const int own_hi, int32_t *__restrict__ out_rank,
                             uint8_t *__restrict__ out_mask, uint8_t *__restrict__ out_clean,
                             const Geo G) {
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long plane = (long long)G.Y * G.X;
    if (t >= (long long)(z_hi - z_lo) * plane) return;
    const int z = z_lo + (int)(t / plane);
    const long long v = (long long)z * plane + t % plane;
    const int x = (int)(v % G.X);
    const long long row = v / G.X;
    if (out_rank) out_rank[t] = (z >= own_lo && z < own_hi) ? rank_vol[v] : 0x7FFFFFFF;
    if (out_mask) {
        out_mask[t] = (mbits[row * row_words(G) + (x >> 5)] >> (x & 31)) & 1u;
        out_clean[t] = dirty[v] ? 0 : 1;
    }
}
__global__ void __launch_bounds__(256)
    cover_zone_import_kernel(int32_t *__restrict__ rank_vol, uint32_t *__restrict__ mbits,
                             uint8_t *__restrict__ dirty, const int z_lo, const int z_hi,
                             const int32_t *__restrict__ in_rank, const uint8_t *__restrict__ in_mask,
                             const uint8_t *__restrict__ in_clean, const Geo G) {
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long plane = (long long)G.Y * G.X;
    if (t >= (long long)(z_hi - z_lo) * plane) return;
    const int z = z_lo + (int)(t / plane);
    const long long v = (long long)z * plane + t % plane;
    const int x = (int)(v % G.X);
    const long long row = v / G.X;
    if (in_rank) rank_vol[v] = in_rank[t] == 0x7FFFFFFF ? RANK_NONE : in_rank[t];
    if (in_mask) {
        if (!in_mask[t]) atomicAnd(&mbits[row * row_words(G) + (x >> 5)], ~(1u << (x & 31)));
        if (!in_clean[t]) dirty[v] = 1;
    }
}
hipError_t cover_zone_export(void *work, int z_lo, int z_hi, int own_lo, int own_hi,
                             int32_t *out_rank, uint8_t *out_mask, uint8_t *out_clean,
                             const Geo &G, hipStream_t s) {
    CoverWork W = carve(work, G);
    const long long n = (long long)(z_hi - z_lo) * G.Y * G.X;
    if (n <= 0) return hipSuccess;
    cover_zone_export_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(
        W.rank_vol, W.mbits, W.dirty, z_lo, z_hi, own_lo, own_hi, out_rank, out_mask, out_clean, G);
    return hipGetLastError();
}
hipError_t cover_zone_import(void *work, int z_lo, int z_hi, const int32_t *in_rank,
                             const uint8_t *in_mask, const uint8_t *in_clean, const Geo &G,
                             hipStream_t s) {
    CoverWork W = carve(work, G);
    const long long n = (long long)(z_hi - z_lo) * G.Y * G.X;
    if (n <= 0) return hipSuccess;
    cover_zone_import_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(
        W.rank_vol, W.mbits, W.dirty, z_lo, z_hi, in_rank, in_mask, in_clean, G);
    return hipGetLastError();
}

// ---- set-cover thinning, one round step at a time on the z-range of one rank (round 6) --------
//
// The same rounds as run_thin_cover with the volume split by z, after the pattern of the sharded greedy
// cover above: every rank keeps key volume, bit mask and dirty marks for its own slices + p-1 halo slices
// and decides its OWN patches; the caller makes the zones of 2(p-1) slices around every slab boundary
// consistent twice per round -- keys after the count step (every slice has one owner, the others
// contribute "none", MIN combines), mask and dirty marks after the select step.  A key's low word is the
// patch's position in the GLOBAL selected list (the sequential loop's tie-break, foreground_cover.py:210);
// loc_vol maps an own centre to its row in the rank's own lists.  The stop rule (foreground_cover.py:
// 206-216) needs every rank's kept patches: the caller gathers (key at selection, cleared interior voxels).
struct ThinShardWork {
    long long *key_vol, *nbr_min, *tmp;  // [V] each
    uint8_t *dirty;                      // [V]
    uint32_t *mbits;                     // [Z*Y][XW]
    int32_t *counters;                   // [COVER_BATCH]
    int32_t *loc_vol;                    // [V]
};
size_t thin_shard_workspace_bytes(const Geo &G) {
    return 3 * up256((size_t)G.V * 8) + up256((size_t)G.V) + up256((size_t)G.Z * G.Y * row_words(G) * 4) + 256 +
           up256((size_t)G.V * 4);
}
static ThinShardWork carve_thin_shard(void *work, const Geo &G) {
    ThinShardWork W;
    char *p = (char *)work;
    W.key_vol = (long long *)p; p += up256((size_t)G.V * 8);
    W.nbr_min = (long long *)p; p += up256((size_t)G.V * 8);
    W.tmp = (long long *)p;     p += up256((size_t)G.V * 8);
    W.dirty = (uint8_t *)p;     p += up256((size_t)G.V);
    W.mbits = (uint32_t *)p;    p += up256((size_t)G.Z * G.Y * row_words(G) * 4);
    W.counters = (int32_t *)p;  p += 256;
    W.loc_vol = (int32_t *)p;
    return W;
}
__global__ void __launch_bounds__(256)
    thin_init_local_kernel(const long long *__restrict__ lin, const int32_t *__restrict__ gidx, int n,
                           long long *__restrict__ key_vol, int32_t *__restrict__ loc_vol) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    key_vol[lin[i]] = (THIN_MAXC << 32) | (long long)gidx[i];     // (count not known yet; centres are distinct)
    loc_vol[lin[i]] = i;
}
hipError_t thin_open(const uint8_t *mask, const long long *lin, const int32_t *gidx, long long n, int32_t *state,
                     int32_t *sel_count, int32_t *cleared, void *work, const Geo &G, hipStream_t s) {
    if (n >= (1ll << 31) || G.C >= THIN_MAXC) return hipErrorInvalidValue;
    PPP_GRID_CHECK((G.V + 255) / 256, 256);
    ThinShardWork W = carve_thin_shard(work, G);
    hipError_t e;
    if ((e = hipMemsetD32Async((hipDeviceptr_t)W.key_vol, 0x7F7F7F7F, (size_t)G.V * 2, s)) != hipSuccess) return e;
    if ((e = hipMemsetD32Async((hipDeviceptr_t)W.loc_vol, 0xFFFFFFFF, (size_t)G.V, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(W.dirty, 1, (size_t)G.V, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(W.counters, 0, COVER_BATCH * 4, s)) != hipSuccess) return e;
    if (n) {
        if ((e = hipMemsetAsync(state, 0, (size_t)n * 4, s)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(sel_count, 0, (size_t)n * 4, s)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(cleared, 0, (size_t)n * 4, s)) != hipSuccess) return e;
    }
    const dim3 block(256);
    const long long n_words = (long long)G.Z * G.Y * row_words(G);
    cover_pack_kernel<<<dim3((unsigned)((n_words + 255) / 256)), block, 0, s>>>(mask, W.mbits, G);
    if (n) thin_init_local_kernel<<<dim3((unsigned)((n + 255) / 256)), block, 0, s>>>(lin, gidx, (int)n, W.key_vol, W.loc_vol);
    return hipGetLastError();
}
hipError_t thin_step_count(const uint32_t *bits, int32_t *state, void *work, const Geo &G, hipStream_t s) {
    ThinShardWork W = carve_thin_shard(work, G);
    hipError_t e;
    if ((e = hipMemsetAsync(W.counters, 0, 4, s)) != hipSuccess) return e;
    thin_count_kernel<<<dim3((unsigned)((G.V + COUNT_THREADS - 1) / COUNT_THREADS)), dim3(COUNT_THREADS), 0, s>>>(
        W.mbits, bits, W.dirty, state, W.key_vol, W.counters, W.loc_vol, G);
    return hipGetLastError();
}
hipError_t thin_step_filter(void *work, const Geo &G, hipStream_t s) {
    ThinShardWork W = carve_thin_shard(work, G);
    minfilter_xy<long long>(W.key_vol, W.tmp, W.nbr_min, G, s);
    return hipGetLastError();
}
hipError_t thin_step_select(const uint32_t *bits, int32_t *state, int32_t *sel_count, int32_t *cleared, void *work,
                            int gZ, const Geo &G, hipStream_t s) {
    ThinShardWork W = carve_thin_shard(work, G);
    thin_select_kernel<<<dim3((unsigned)((G.V + 255) / 256)), dim3(256), 0, s>>>(
        W.mbits, bits, W.nbr_min, state, W.key_vol, sel_count, cleared, W.dirty, W.loc_vol, G.oz, gZ, G);
    return hipGetLastError();
}
hipError_t thin_alive(void *work, const Geo &G, int32_t *alive, hipStream_t s) {
    ThinShardWork W = carve_thin_shard(work, G);
    hipError_t e;
    if ((e = hipMemcpyAsync(alive, W.counters, 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    return hipStreamSynchronize(s);
}
hipError_t thin_close(uint8_t *mask, void *work, const Geo &G, hipStream_t s) {
    ThinShardWork W = carve_thin_shard(work, G);
    cover_unpack_kernel<<<dim3((unsigned)((G.V + 255) / 256)), dim3(256), 0, s>>>(W.mbits, mask, G);
    return hipGetLastError();
}
static constexpr long long THIN_ZONE_NONE = 0x7FFFFFFFFFFFFFFFll;      // "not my slice" in an exported zone of keys
__global__ void __launch_bounds__(256)
    thin_zone_export_kernel(const long long *__restrict__ key_vol, const uint32_t *__restrict__ mbits,
                            const uint8_t *__restrict__ dirty, const int z_lo, const int z_hi, const int own_lo,
                            const int own_hi, long long *__restrict__ out_key, uint8_t *__restrict__ out_mask,
                            uint8_t *__restrict__ out_clean, const Geo G) {
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long plane = (long long)G.Y * G.X;
    if (t >= (long long)(z_hi - z_lo) * plane) return;
    const int z = z_lo + (int)(t / plane);
    const long long v = (long long)z * plane + t % plane;
    const int x = (int)(v % G.X);
    const long long row = v / G.X;
    if (out_key) out_key[t] = (z >= own_lo && z < own_hi) ? key_vol[v] : THIN_ZONE_NONE;
    if (out_mask) {
        out_mask[t] = (mbits[row * row_words(G) + (x >> 5)] >> (x & 31)) & 1u;
        out_clean[t] = dirty[v] ? 0 : 1;
    }
}
__global__ void __launch_bounds__(256)
    thin_zone_import_kernel(long long *__restrict__ key_vol, uint32_t *__restrict__ mbits, uint8_t *__restrict__ dirty,
                            const int z_lo, const int z_hi, const long long *__restrict__ in_key,
                            const uint8_t *__restrict__ in_mask, const uint8_t *__restrict__ in_clean, const Geo G) {
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long plane = (long long)G.Y * G.X;
    if (t >= (long long)(z_hi - z_lo) * plane) return;
    const int z = z_lo + (int)(t / plane);
    const long long v = (long long)z * plane + t % plane;
    const int x = (int)(v % G.X);
    const long long row = v / G.X;
    if (in_key) key_vol[v] = in_key[t] == THIN_ZONE_NONE ? THIN_NONE : in_key[t];
    if (in_mask) {
        if (!in_mask[t]) atomicAnd(&mbits[row * row_words(G) + (x >> 5)], ~(1u << (x & 31)));
        if (!in_clean[t]) dirty[v] = 1;
    }
}
hipError_t thin_zone_export(void *work, int z_lo, int z_hi, int own_lo, int own_hi, long long *out_key,
                            uint8_t *out_mask, uint8_t *out_clean, const Geo &G, hipStream_t s) {
    ThinShardWork W = carve_thin_shard(work, G);
    const long long n = (long long)(z_hi - z_lo) * G.Y * G.X;
    if (n <= 0) return hipSuccess;
    thin_zone_export_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(
        W.key_vol, W.mbits, W.dirty, z_lo, z_hi, own_lo, own_hi, out_key, out_mask, out_clean, G);
    return hipGetLastError();
}
hipError_t thin_zone_import(void *work, int z_lo, int z_hi, const long long *in_key, const uint8_t *in_mask,
                            const uint8_t *in_clean, const Geo &G, hipStream_t s) {
    ThinShardWork W = carve_thin_shard(work, G);
    const long long n = (long long)(z_hi - z_lo) * G.Y * G.X;
    if (n <= 0) return hipSuccess;
    thin_zone_import_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(
        W.key_vol, W.mbits, W.dirty, z_lo, z_hi, in_key, in_mask, in_clean, G);
    return hipGetLastError();
}

}  // namespace ppp
