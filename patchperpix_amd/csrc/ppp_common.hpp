// ppp_common.hpp -- shared device-side geometry / arithmetic helpers (gfx950 only).
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ppp_mi355x.h"

namespace ppp {

// Device view of ppp_params with everything pre-digested on the host.
struct Geo {
    int Z, Y, X;
    int pz, py, px;
    int rz, ry, rx;
    int C, mid;
    long long V;  // Z*Y*X
    // Threshold compares.  The reference compares float values against DOUBLE
    // literals (TH, THI, TH/2).  For a float v and a double t:
    //     v >  t  <=>  v >  round_down_to_float(t)
    //     v <  t  <=>  v <  round_up_to_float(t)
    // so the double compares are replaced by exact float compares.
    float th_gt;  // v > TH   <=> v > th_gt
    float bg_lt;  // v < BG   <=> v < bg_lt
    float th_rn;  // (float)TH, round to nearest: NumPy's float32 compare of host stages
    double th2;   // TH*TH
    double den;   // 1.0 - TH*TH
    int value_rule, use_overlap, normalise, norm_rank, count_pos_neg, norm_aff;
    int layout;
    int bz0, by0, bx0;  // cons box origin
    int bZ, bY, bX;     // cons box extent
    long long BV;       // bZ*bY*bX
    int nsy, nsx;       // reference layout plane grid
    int wy, wx;         // 2py-1, 2px-1 (compact plane grid)
    int n_planes;       // compact: ((2pz-1)(2py-1)(2px-1)-1)/2
    int oz, oy, ox;     // global coordinate of local voxel (0,0,0)
    int vm_open;        // voxel-major output: entries with a source outside the box stay undefined
    // S1 only: the base voxels a launch COMPUTES (a sub-box of the cons box, which stays the
    // box the output buffer is indexed by); the whole cons box unless ppp_consensus_part asks
    int cz0, cy0, cx0, cZ, cY, cX;
    int ring;           // voxel-major rows in a ring of `ring` z-slices (0: the plain box), ppp_params.ring_z
    int pred_clean;     // 1: every prediction value in [0, 1] and != TH (ppp_params.pred_clean, ppp_pred_check)
    int rank_tile;      // ppp_params.rank_tile: 0 = the launcher's rule, 1 / 2 / 3 = 8x8x16 / 8x16x16 / 16x8x16
};

// A HIP grid is limited to 2^32 - 1 work-items per dimension (blocks x threads): a larger launch
// does NOT fail, it silently runs part of the grid.  Every launcher whose grid grows with the
// volume either chunks its work or refuses with this check.
static inline bool grid_too_big(unsigned long long blocks, unsigned threads) {
    return blocks * (unsigned long long)threads >= (1ull << 32);
}
#define PPP_GRID_CHECK(blocks, threads) \
    do { if (::ppp::grid_too_big((unsigned long long)(blocks), (unsigned)(threads))) return hipErrorInvalidConfiguration; } while (0)

template <typename T>
__device__ __forceinline__ float ldf(const T *p, long long i);
template <>
__device__ __forceinline__ float ldf<float>(const float *p, long long i) {
    return p[i];
}
template <>
__device__ __forceinline__ float ldf<__half>(const __half *p, long long i) {
    return __half2float(p[i]);
}

__device__ __forceinline__ long long vox(const Geo &G, int z, int y, int x) {
    return ((long long)z * G.Y + y) * G.X + x;
}
__device__ __forceinline__ bool interior(const Geo &G, int z, int y, int x) {
    return x >= G.rx && x < G.X - G.rx && y >= G.ry && y < G.Y - G.ry && z >= G.rz &&
           z < G.Z - G.rz;
}
// element offset of consensus entry (offset d lexicographically positive, base voxel
// in GLOBAL coordinates)
__device__ __forceinline__ long long cons_at(const Geo &G, int dz, int dy, int dx, int z, int y,
                                             int x) {
    if (G.layout == PPP_CONS_REFERENCE) {
        const long long plane =
            ((long long)(dz + G.pz - 1) * G.nsy + (dy + G.py - 1)) * G.nsx + (dx + G.px - 1);
        return plane * G.V + vox(G, z, y, x);
    }
    const long long plane = ((long long)dz * G.wy + dy) * G.wx + dx - 1;
    return plane * G.BV + ((long long)(z - G.bz0) * G.bY + (y - G.by0)) * G.bX + (x - G.bx0);
}
// slice of the voxel-major row buffer that holds the rows of (local) slice z: box-relative, or
// (z + origin) mod ring when the buffer is a ring
__device__ __forceinline__ int row_slice(const Geo &G, int z) {
    return G.ring ? (z + G.oz) % G.ring : z - G.bz0;
}
// value of one vote from the float product x = v1*v2 or v1*(1-v2)
// (fillConsensusArray.cu:104-110,127-133): normalisation in double, rounded to float.
__device__ __forceinline__ float vote_value(const Geo &G, float x) {
    if (G.value_rule == PPP_VAL_NORM_PROB_PRODUCT) return (float)(((double)x - G.th2) / G.den);
    if (G.value_rule == PPP_VAL_PROB_PRODUCT) return x;
    return 1.0f;
}

}  // namespace ppp
