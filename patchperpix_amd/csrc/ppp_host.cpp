// ppp_host.cpp -- the HOST stages of vote_instances (they are host code in the reference
// too: Python loops between the kernel launches).  Plain C++, no device code; results are
// integer-exact restatements of:
//   rank_patches_by_score        ranked_patches.py:21-30  + vote_instances.py:276,286-287
//   computeForegroundCover(Loop) foreground_cover.py:15-180
//   thinOutForegroundCover       foreground_cover.py:183-256
//   computeAndStorePatchPairs    aff_patch_graph.py:43-110
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <set>
#include <unordered_map>
#include <vector>

#include "../../include/ppp_mi355x.h"

namespace {

struct Dims {
    int Z, Y, X, pz, py, px, rz, ry, rx, C, words;
    Dims(const int32_t *vol, const int32_t *ps)
        : Z(vol[0]), Y(vol[1]), X(vol[2]), pz(ps[0]), py(ps[1]), px(ps[2]), rz(ps[0] / 2),
          ry(ps[1] / 2), rx(ps[2] / 2), C(ps[0] * ps[1] * ps[2]), words((C + 31) / 32) {}
    bool interior(int z, int y, int x) const {
        return z >= rz && z < Z - rz && y >= ry && y < Y - ry && x >= rx && x < X - rx;
    }
    int64_t lin(int z, int y, int x) const { return ((int64_t)z * Y + y) * X + x; }
};

inline bool bit(const uint32_t *b, int r) { return (b[r >> 5] >> (r & 31)) & 1u; }

// bits [r, r + n) of a packed bit table (n <= 32)
inline uint32_t bit_field(const uint32_t *b, int r, int n) {
    const int w = r >> 5, sh = r & 31;
    uint64_t v = b[w];
    if (sh + n > 32) v |= (uint64_t)b[w + 1] << 32;
    return (uint32_t)(v >> sh) & (n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u));
}

// 8 mask bytes (each 0 or 1) -> 8 bits, byte k -> bit k
inline uint32_t pack8(const uint8_t *p) {
    uint64_t w;
    memcpy(&w, p, 8);
    return (uint32_t)((w * 0x0102040810204080ull) >> 56);
}

// number of window voxels that are set in `mask` (bytes 0/1) and in the patch bits; optionally
// clears them (returns how many of the cleared voxels were interior through *cleared_interior).
// Rows are tested 8 voxels at a time: pack the mask bytes to bits, AND with the patch bits of
// that row, popcount.  `mask` must have 8 readable bytes past the last voxel (callers pad).
template <bool CLEAR>
inline int window_hits(const Dims &D, uint8_t *mask, const uint32_t *bits, int cz, int cy,
                       int cx, int64_t *cleared_interior) {
    int hits = 0, r = 0;
    for (int dz = 0; dz < D.pz; ++dz)
        for (int dy = 0; dy < D.py; ++dy, r += D.px) {
            const int z = cz + dz - D.rz, y = cy + dy - D.ry;
            uint8_t *row = mask + D.lin(z, y, cx - D.rx);
            for (int x0 = 0; x0 < D.px; x0 += 8) {
                const int n = D.px - x0 < 8 ? D.px - x0 : 8;
                uint32_t m = pack8(row + x0) & bit_field(bits, r + x0, n);
                hits += __builtin_popcount(m);
                if (CLEAR)
                    while (m) {
                        const int k = __builtin_ctz(m);
                        m &= m - 1;
                        row[x0 + k] = 0;
                        if (D.interior(z, y, cx + x0 + k - D.rx)) ++*cleared_interior;
                    }
            }
        }
    return hits;
}

}  // namespace

extern "C" {

// Interior foreground voxels in raster order, stably sorted by score descending.
// out_lin must hold Z*Y*X entries; returns the count (or <0).
int64_t ppp_host_rank_order(const float *h_score, const uint8_t *h_foreground, const int32_t *vol,
                            const int32_t *patchshape, int64_t *out_lin) {
    const Dims D(vol, patchshape);
    int64_t n = 0;
    for (int z = D.rz; z < D.Z - D.rz; ++z)
        for (int y = D.ry; y < D.Y - D.ry; ++y) {
            const int64_t base = D.lin(z, y, 0);
            for (int x = D.rx; x < D.X - D.rx; ++x)
                if (h_foreground[base + x]) out_lin[n++] = base + x;
        }
    std::stable_sort(out_lin, out_lin + n,
                     [h_score](int64_t a, int64_t b) { return h_score[a] > h_score[b]; });
    return n;
}

// One pass of computeForegroundCoverLoop over the ranked list (foreground_cover.py:111-180).
//   h_mask_running : uint8 (Z,Y,X) with values 0/1, updated in place; the buffer must have 8
//                    readable bytes after the last voxel (rows are tested 8 voxels at a time)
//   h_overlap      : uint8 (Z,Y,X) or NULL
//   ranked_lin     : linear voxel index of every ranked patch centre (all interior)
//   bits           : [n][words] patch bits (pred[:,c] > fc_threshold), see ppp_patch_bits
//   selected       : uint8 [n] in/out
//   remaining      : in/out, number of set interior voxels of the running mask
//   score_threshold: NaN = disabled (foreground_cover.py:136-138)
// Returns the number of newly selected patches.
int64_t ppp_host_cover_pass_marked(uint8_t *h_mask_running, const uint8_t *h_overlap, const int32_t *vol,
                                   const int32_t *patchshape, const int64_t *ranked_lin,
                                   const float *ranked_score, const uint32_t *bits, int64_t n,
                                   int32_t pix_th, double score_threshold, uint8_t *selected,
                                   int64_t *remaining, int32_t *stopped, uint8_t *h_marked);

int64_t ppp_host_cover_pass(uint8_t *h_mask_running, const uint8_t *h_overlap, const int32_t *vol,
                            const int32_t *patchshape, const int64_t *ranked_lin,
                            const float *ranked_score, const uint32_t *bits, int64_t n,
                            int32_t pix_th, double score_threshold, uint8_t *selected,
                            int64_t *remaining, int32_t *stopped) {
    return ppp_host_cover_pass_marked(h_mask_running, h_overlap, vol, patchshape, ranked_lin, ranked_score, bits, n,
                                      pix_th, score_threshold, selected, remaining, stopped, nullptr);
}

// The same pass with `mark_close_neighboorhood` (foreground_cover.py:141-143, 162-168): h_marked
// (uint8 (Z,Y,X), in/out, shared by the passes of a cover) -- a ranked patch whose centre is marked
// is skipped, and a selected patch marks the box (0, +-3, +-3) around its centre.  The reference
// builds that box as NumPy slices: a NEGATIVE start (centre closer than 3 to the low y / x border)
// makes the slice wrap around and select nothing, so such a patch marks nothing; the high end clips.
int64_t ppp_host_cover_pass_marked(uint8_t *h_mask_running, const uint8_t *h_overlap, const int32_t *vol,
                                   const int32_t *patchshape, const int64_t *ranked_lin,
                                   const float *ranked_score, const uint32_t *bits, int64_t n,
                                   int32_t pix_th, double score_threshold, uint8_t *selected,
                                   int64_t *remaining, int32_t *stopped, uint8_t *h_marked) {
    const Dims D(vol, patchshape);
    int64_t picked = 0;
    if (stopped) *stopped = 0;
    // Coarse occupancy grid (8^3 blocks) of the running mask: once most of the foreground is
    // covered, almost every remaining candidate has an empty window and is rejected by looking
    // at <= 8 block counters instead of its p^3 window.
    constexpr int B = 8;
    const int GZ = (D.Z + B - 1) / B, GY = (D.Y + B - 1) / B, GX = (D.X + B - 1) / B;
    std::vector<int32_t> occ((size_t)GZ * GY * GX, 0);
    for (int z = 0; z < D.Z; ++z)
        for (int y = 0; y < D.Y; ++y) {
            const uint8_t *row = h_mask_running + D.lin(z, y, 0);
            int32_t *orow = occ.data() + ((size_t)(z / B) * GY + y / B) * GX;
            for (int x = 0; x < D.X; ++x) orow[x / B] += row[x] != 0;
        }
    auto window_empty = [&](int cz, int cy, int cx) {
        for (int gz = (cz - D.rz) / B; gz <= (cz + D.rz) / B; ++gz)
            for (int gy = (cy - D.ry) / B; gy <= (cy + D.ry) / B; ++gy)
                for (int gx = (cx - D.rx) / B; gx <= (cx + D.rx) / B; ++gx)
                    if (occ[((size_t)gz * GY + gy) * GX + gx]) return false;
        return true;
    };
    for (int64_t i = 0; i < n && *remaining > 0; ++i) {
        if (selected[i]) continue;
        if (!std::isnan(score_threshold) && (double)ranked_score[i] < score_threshold) {
            if (stopped) *stopped = 1;  // the reference breaks out of the whole pass
            break;
        }
        const int64_t c = ranked_lin[i];
        if (h_overlap && h_overlap[c] > 0) continue;
        if (h_marked && h_marked[c]) continue;
        const int cx = (int)(c % D.X), cy = (int)((c / D.X) % D.Y), cz = (int)(c / ((int64_t)D.X * D.Y));
        if (pix_th >= 0 && window_empty(cz, cy, cx)) continue;   // count 0 is never > pix_th
        const uint32_t *b = bits + i * D.words;
        if (window_hits<false>(D, h_mask_running, b, cz, cy, cx, nullptr) > pix_th) {
            selected[i] = 1;
            ++picked;
            // clear row by row (8 voxels at a time), keeping the occupancy grid in step
            int64_t cleared = 0;
            int r = 0;
            for (int dz = 0; dz < D.pz; ++dz)
                for (int dy = 0; dy < D.py; ++dy, r += D.px) {
                    const int z = cz + dz - D.rz, y = cy + dy - D.ry;
                    const bool zy_int = z >= D.rz && z < D.Z - D.rz && y >= D.ry && y < D.Y - D.ry;
                    uint8_t *row = h_mask_running + D.lin(z, y, cx - D.rx);
                    int32_t *orow = occ.data() + ((size_t)(z / B) * GY + y / B) * GX;
                    for (int x0 = 0; x0 < D.px; x0 += 8) {
                        const int nb = D.px - x0 < 8 ? D.px - x0 : 8;
                        uint32_t m = pack8(row + x0) & bit_field(b, r + x0, nb);
                        while (m) {
                            const int k = __builtin_ctz(m);
                            m &= m - 1;
                            const int x = cx - D.rx + x0 + k;
                            row[x0 + k] = 0;
                            --orow[x / B];
                            cleared += zy_int && x >= D.rx && x < D.X - D.rx;
                        }
                    }
                }
            *remaining -= cleared;
            if (h_marked) {
                // marked[z, cy-3:cy+4, cx-3:cx+4] = True with NumPy's slice rule: a negative
                // start counts from the end (an empty slice on any axis wider than 7)
                auto bounds = [](int start, int stop, int size, int &a, int &b) {
                    a = start < 0 ? std::max(start + size, 0) : std::min(start, size);
                    b = stop < 0 ? std::max(stop + size, 0) : std::min(stop, size);
                };
                int ya, yb, xa, xb;
                bounds(cy - 3, cy + 4, D.Y, ya, yb);
                bounds(cx - 3, cx + 4, D.X, xa, xb);
                for (int y = ya; y < yb; ++y)
                    for (int x = xa; x < xb; ++x) h_marked[D.lin(cz, y, x)] = 1;
            }
        }
    }
    return picked;
}

// Greedy set-cover thinning (foreground_cover.py:183-256, sample == 1.0).
//   sel_lin [n], bits [n][words] (same bit definition), h_mask: uint8 (Z,Y,X) = mask_to_cover
//   (not modified).  keep: uint8 [n] out.  Returns number kept.
int64_t ppp_host_thin_cover(const uint8_t *h_mask, const int32_t *vol, const int32_t *patchshape,
                            const int64_t *sel_lin, const uint32_t *bits, int64_t n,
                            uint8_t *keep) {
    const Dims D(vol, patchshape);
    const int64_t V = (int64_t)D.Z * D.Y * D.X;
    std::vector<uint8_t> running(V + 8, 0);   // + 8 readable bytes for the 8-at-a-time row test
    memcpy(running.data(), h_mask, (size_t)V);
    memset(keep, 0, (size_t)n);
    if (n == 0) return 0;
    int64_t remaining = 0;
    for (int z = D.rz; z < D.Z - D.rz; ++z)
        for (int y = D.ry; y < D.Y - D.ry; ++y)
            for (int x = D.rx; x < D.X - D.rx; ++x) remaining += running[D.lin(z, y, x)] != 0;

    std::vector<int> cz(n), cy(n), cx(n);
    std::vector<int64_t> count(n);
    // coarse grid over patch centres to find the patches whose windows overlap a window
    const int gz = std::max(1, D.pz), gy = std::max(1, D.py), gx = std::max(1, D.px);
    const int GZ = D.Z / gz + 1, GY = D.Y / gy + 1, GX = D.X / gx + 1;
    std::unordered_map<int64_t, std::vector<int64_t>> grid;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t c = sel_lin[i];
        cx[i] = (int)(c % D.X); cy[i] = (int)((c / D.X) % D.Y); cz[i] = (int)(c / ((int64_t)D.X * D.Y));
        count[i] = window_hits<false>(D, running.data(), bits + i * D.words, cz[i], cy[i], cx[i], nullptr);
        grid[((int64_t)(cz[i] / gz) * GY + cy[i] / gy) * GX + cx[i] / gx].push_back(i);
    }
    // ordered by (count desc, index asc): begin() is numpy's first argmax
    std::set<std::pair<int64_t, int64_t>> order;
    for (int64_t i = 0; i < n; ++i) order.insert({-count[i], i});
    int64_t kept = 0;
    while (remaining > 0) {
        const int64_t best = order.begin()->second;
        if (!keep[best]) { keep[best] = 1; ++kept; }
        if (count[best] == 0) break;  // empty best set: the reference zeroes the whole mask
        int64_t cleared = 0;
        window_hits<true>(D, running.data(), bits + best * D.words, cz[best], cy[best], cx[best], &cleared);
        remaining -= cleared;
        // refresh every patch whose window can intersect the cleared window
        for (int bz = (cz[best] - D.pz + 1) / gz - 1; bz <= (cz[best] + D.pz - 1) / gz + 1; ++bz)
            for (int by = (cy[best] - D.py + 1) / gy - 1; by <= (cy[best] + D.py - 1) / gy + 1; ++by)
                for (int bx = (cx[best] - D.px + 1) / gx - 1; bx <= (cx[best] + D.px - 1) / gx + 1; ++bx) {
                    if (bz < 0 || by < 0 || bx < 0 || bz >= GZ || by >= GY || bx >= GX) continue;
                    auto it = grid.find(((int64_t)bz * GY + by) * GX + bx);
                    if (it == grid.end()) continue;
                    for (int64_t j : it->second) {
                        if (std::abs(cz[j] - cz[best]) >= D.pz || std::abs(cy[j] - cy[best]) >= D.py ||
                            std::abs(cx[j] - cx[best]) >= D.px)
                            continue;
                        const int64_t c2 = window_hits<false>(D, running.data(), bits + j * D.words,
                                                              cz[j], cy[j], cx[j], nullptr);
                        if (c2 != count[j]) {
                            order.erase({-count[j], j});
                            count[j] = c2;
                            order.insert({-count[j], j});
                        }
                    }
                }
    }
    return kept;
}

// Patch pairs (aff_patch_graph.py:43-110).  sel_zyx int32 [n][3] in selection order.
//  - sorted_zyx out [n][3]: the list stably sorted by x (the reference sorts in place)
//  - pairs out: rows (A, B) with A before B in the x-sorted list and |dA-B|_i <=
//    max_ps_dist * p_i on every axis, ordered by (index A, index B); followed by the n
//    self pairs when include_single.  Call with pairs == NULL to get the row count.
int64_t ppp_host_patch_pairs(const int32_t *sel_zyx, int64_t n, const int32_t *patchshape,
                             int32_t max_ps_dist, int32_t include_single, int32_t *sorted_zyx,
                             uint32_t *pairs) {
    std::vector<int64_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [sel_zyx](int64_t a, int64_t b) {
        return sel_zyx[a * 3 + 2] < sel_zyx[b * 3 + 2];
    });
    std::vector<int32_t> pts((size_t)n * 3);
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) pts[i * 3 + k] = sel_zyx[idx[i] * 3 + k];
    if (sorted_zyx) memcpy(sorted_zyx, pts.data(), sizeof(int32_t) * (size_t)n * 3);
    const int bz = std::max(1, max_ps_dist * patchshape[0]), by = std::max(1, max_ps_dist * patchshape[1]),
              bx = std::max(1, max_ps_dist * patchshape[2]);
    // grid hash with cell = box half-width: neighbours are in the 3x3x3 surrounding cells
    std::unordered_map<int64_t, std::vector<int64_t>> grid;
    auto key = [](int64_t a, int64_t b, int64_t c) { return (a * 2097152 + b) * 2097152 + c; };
    for (int64_t i = 0; i < n; ++i)
        grid[key(pts[i * 3] / bz, pts[i * 3 + 1] / by, pts[i * 3 + 2] / bx)].push_back(i);
    int64_t rows = 0;
    std::vector<int64_t> nb;
    for (int64_t i = 0; i < n; ++i) {
        nb.clear();
        const int z = pts[i * 3], y = pts[i * 3 + 1], x = pts[i * 3 + 2];
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int64_t gz = z / bz + dz, gy = y / by + dy, gx = x / bx + dx;
                    if (gz < 0 || gy < 0 || gx < 0) continue;
                    auto it = grid.find(key(gz, gy, gx));
                    if (it == grid.end()) continue;
                    for (int64_t j : it->second)
                        if (j > i && std::abs(pts[j * 3] - z) <= max_ps_dist * patchshape[0] &&
                            std::abs(pts[j * 3 + 1] - y) <= max_ps_dist * patchshape[1] &&
                            std::abs(pts[j * 3 + 2] - x) <= max_ps_dist * patchshape[2] &&
                            // cKDTree.query_pairs(2*sum(p), p=1): implied by the box for the
                            // default max_ps_dist = 2, binding for larger values
                            std::abs(pts[j * 3] - z) + std::abs(pts[j * 3 + 1] - y) +
                                    std::abs(pts[j * 3 + 2] - x) <=
                                2 * (patchshape[0] + patchshape[1] + patchshape[2]))
                            nb.push_back(j);
                }
        std::sort(nb.begin(), nb.end());
        if (pairs)
            for (int64_t j : nb) {
                uint32_t *row = pairs + rows * 6;
                row[0] = (uint32_t)z; row[1] = (uint32_t)y; row[2] = (uint32_t)x;
                row[3] = (uint32_t)pts[j * 3]; row[4] = (uint32_t)pts[j * 3 + 1]; row[5] = (uint32_t)pts[j * 3 + 2];
                ++rows;
            }
        else
            rows += (int64_t)nb.size();
    }
    if (include_single) {
        if (pairs)
            for (int64_t i = 0; i < n; ++i) {
                uint32_t *row = pairs + rows * 6;
                for (int k = 0; k < 3; ++k) row[k] = row[3 + k] = (uint32_t)pts[i * 3 + k];
                ++rows;
            }
        else
            rows += n;
    }
    return rows;
}

}  // extern "C"
