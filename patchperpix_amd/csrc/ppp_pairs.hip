// ppp_pairs.hip -- patch-pair enumeration on the device.
//
// Reference: aff_patch_graph.py:43-110 (computeAndStorePatchPairs): the selected patches are
// stably sorted by x; cKDTree.query_pairs(2*sum(p), p=1) proposes index pairs (i < j), those
// with |pts[i] - pts[j]|_k > max_ps_dist * p_k on some axis k are dropped; rows are
// (pts[i], pts[j]); self pairs (c, c) follow when includeSinglePatchCCS.
//
// Here: the x-sorted list lives on the device.  Because it is sorted by x, the partners
// j > i of patch i form a contiguous index range (x_j <= x_i + max_ps_dist*px); one wave
// scans that range 64 candidates at a time, tests the box (and the L1 ball), and compacts the
// hits with ballot/popcount -- hits come out in ascending j, i.e. rows are produced directly
// in the canonical (i, j) order, with no sort and no atomics.  Two passes: count, then fill at
// the offsets given by an exclusive scan of the counts.
#include "ppp_kernels.hpp"

namespace ppp {

template <bool FILL>
__global__ void __launch_bounds__(256)
    pairs_kernel(const int32_t *__restrict__ pts, const int64_t n, const int bz, const int by,
                 const int bx, const int l1max, int64_t *__restrict__ counts,
                 const int64_t *__restrict__ offsets, uint32_t *__restrict__ rows,
                 const int64_t *__restrict__ subset, const int64_t m) {
    // (subset != nullptr: only the m listed patches are scanned -- the count pass of a rank
    // that owns part of the volume)
    const int64_t k = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (k >= (subset ? m : n)) return;
    const int64_t i = subset ? subset[k] : k;
    const int lane = threadIdx.x & 63;
    const int z = pts[i * 3], y = pts[i * 3 + 1], x = pts[i * 3 + 2];
    int64_t out = FILL ? offsets[i] : 0;
    int64_t found = 0;
    for (int64_t j0 = i + 1; j0 < n; j0 += 64) {
        const int64_t j = j0 + lane;
        bool hit = false;
        int jz = 0, jy = 0, jx = 0;
        bool beyond = true;
        if (j < n) {
            jz = pts[j * 3]; jy = pts[j * 3 + 1]; jx = pts[j * 3 + 2];
            beyond = jx - x > bx;
            const int az = abs(jz - z), ay = abs(jy - y), ax = abs(jx - x);
            hit = !beyond && az <= bz && ay <= by && ax <= bx && az + ay + ax <= l1max;
        }
        const unsigned long long m = __ballot(hit);
        if (FILL && hit) {
            uint32_t *row = rows + (out + __popcll(m & ((1ull << lane) - 1ull))) * 6;
            row[0] = (uint32_t)z; row[1] = (uint32_t)y; row[2] = (uint32_t)x;
            row[3] = (uint32_t)jz; row[4] = (uint32_t)jy; row[5] = (uint32_t)jx;
        }
        out += __popcll(m);
        found += __popcll(m);
        // sorted by x: once a whole chunk lies beyond x + bx nothing further can match
        if (__ballot(beyond) == ~0ull) break;
    }
    if (!FILL && lane == 0) counts[i] = found;
}

__global__ void self_pairs_kernel(const int32_t *__restrict__ pts, const int64_t n,
                                  uint32_t *__restrict__ rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < 3; ++k) rows[i * 6 + k] = rows[i * 6 + 3 + k] = (uint32_t)pts[i * 3 + k];
}

// Rows of a SUBSET of the first patches (the patches of one tile), written compactly:
// subset[k] = index i into the x-sorted list, local_off[k] = first local row of i,
// gid_off[i] = first GLOBAL row of i (exclusive scan of all counts); gid receives the global
// row id of every local row.  Same scan and order as pairs_kernel.
__global__ void __launch_bounds__(256)
    pairs_subset_kernel(const int32_t *__restrict__ pts, const int64_t n, const int bz,
                        const int by, const int bx, const int l1max,
                        const int64_t *__restrict__ subset, const int64_t m,
                        const int64_t *__restrict__ local_off, const int64_t *__restrict__ gid_off,
                        uint32_t *__restrict__ rows, long long *__restrict__ gid) {
    const int64_t k = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (k >= m) return;
    const int64_t i = subset[k];
    const int lane = threadIdx.x & 63;
    const int z = pts[i * 3], y = pts[i * 3 + 1], x = pts[i * 3 + 2];
    int64_t out = local_off[k];
    const int64_t shift = gid_off[i] - out;
    for (int64_t j0 = i + 1; j0 < n; j0 += 64) {
        const int64_t j = j0 + lane;
        bool hit = false;
        int jz = 0, jy = 0, jx = 0;
        bool beyond = true;
        if (j < n) {
            jz = pts[j * 3]; jy = pts[j * 3 + 1]; jx = pts[j * 3 + 2];
            beyond = jx - x > bx;
            const int az = abs(jz - z), ay = abs(jy - y), ax = abs(jx - x);
            hit = !beyond && az <= bz && ay <= by && ax <= bx && az + ay + ax <= l1max;
        }
        const unsigned long long mk = __ballot(hit);
        if (hit) {
            const int64_t r = out + __popcll(mk & ((1ull << lane) - 1ull));
            uint32_t *row = rows + r * 6;
            row[0] = (uint32_t)z; row[1] = (uint32_t)y; row[2] = (uint32_t)x;
            row[3] = (uint32_t)jz; row[4] = (uint32_t)jy; row[5] = (uint32_t)jx;
            gid[r] = r + shift;
        }
        out += __popcll(mk);
        if (__ballot(beyond) == ~0ull) break;
    }
}
__global__ void self_pairs_subset_kernel(const int32_t *__restrict__ pts,
                                         const int64_t *__restrict__ subset, const int64_t m,
                                         const int64_t gid_self0, uint32_t *__restrict__ rows,
                                         long long *__restrict__ gid) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const int64_t i = subset[k];
    for (int c = 0; c < 3; ++c) rows[k * 6 + c] = rows[k * 6 + 3 + c] = (uint32_t)pts[i * 3 + c];
    gid[k] = gid_self0 + i;
}

hipError_t launch_pairs_subset(const int32_t *pts, int64_t n, const int *box, int l1max,
                               const int64_t *subset, int64_t m, const int64_t *local_off,
                               const int64_t *gid_off, int64_t n_local_rows, int64_t n_rows_total,
                               int include_single, uint32_t *rows, long long *gid, hipStream_t s) {
    if (n == 0 || m == 0) return hipSuccess;
    PPP_GRID_CHECK((m + 3) / 4, 256);
    pairs_subset_kernel<<<dim3((unsigned)((m + 3) / 4)), dim3(256), 0, s>>>(
        pts, n, box[0], box[1], box[2], l1max, subset, m, local_off, gid_off, rows, gid);
    if (include_single)
        self_pairs_subset_kernel<<<dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s>>>(
            pts, subset, m, n_rows_total, rows + n_local_rows * 6, gid + n_local_rows);
    return hipGetLastError();
}

hipError_t launch_pairs_count(const int32_t *pts, int64_t n, const int *box, int l1max,
                              int64_t *counts, hipStream_t s, const int64_t *subset, int64_t m) {
    const int64_t waves = subset ? m : n;
    if (n == 0 || waves == 0) return hipSuccess;
    PPP_GRID_CHECK((waves + 3) / 4, 256);
    pairs_kernel<false><<<dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s>>>(
        pts, n, box[0], box[1], box[2], l1max, counts, nullptr, nullptr, subset, m);
    return hipGetLastError();
}

hipError_t launch_pairs_fill(const int32_t *pts, int64_t n, const int *box, int l1max,
                             const int64_t *offsets, int64_t n_pair_rows, int include_single,
                             uint32_t *rows, hipStream_t s) {
    if (n == 0) return hipSuccess;
    PPP_GRID_CHECK((n + 3) / 4, 256);
    pairs_kernel<true><<<dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s>>>(
        pts, n, box[0], box[1], box[2], l1max, nullptr, offsets, rows, nullptr, 0);
    if (include_single)
        self_pairs_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(
            pts, n, rows + n_pair_rows * 6);
    return hipGetLastError();
}

// key = ((dz+2pz)*(4py+1) + (dy+2py))*(4px+1) + (dx+2px), then position of A: the sort key
// that groups rows by patch offset for ppp_patch_graph
__global__ void pair_keys_kernel(const uint32_t *__restrict__ rows, const uint64_t n,
                                 int64_t *__restrict__ keys, const Geo G) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *r = rows + i * 6;
    const int64_t dz = (int64_t)r[3] - r[0] + 2 * G.pz, dy = (int64_t)r[4] - r[1] + 2 * G.py,
                  dx = (int64_t)r[5] - r[2] + 2 * G.px;
    const int64_t key = (dz * (4 * G.py + 1) + dy) * (4 * G.px + 1) + dx;
    keys[i] = key * G.V + (((int64_t)r[0] * G.Y + r[1]) * G.X + r[2]);
}

// key for ppp_patch_graph_by_patch: rows of one patch A together; inside a group first the
// rows whose two windows intersect (they alone run the LCG thinning), then by patch offset
__global__ void pair_group_keys_kernel(const uint32_t *__restrict__ rows, const uint64_t n,
                                       int64_t *__restrict__ keys, const Geo G) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *r = rows + i * 6;
    const int dz = (int)r[3] - (int)r[0], dy = (int)r[4] - (int)r[1], dx = (int)r[5] - (int)r[2];
    const int64_t dkey = ((int64_t)(dz + 2 * G.pz) * (4 * G.py + 1) + (dy + 2 * G.py)) * (4 * G.px + 1) +
                         (dx + 2 * G.px);
    const int64_t apart = (abs(dz) >= G.pz || abs(dy) >= G.py || abs(dx) >= G.px) ? 1 : 0;
    const int64_t lin = ((int64_t)r[0] * G.Y + r[1]) * G.X + r[2];
    // A pair whose centres are more than 2(p-1) apart on some axis has no pixel pair within the
    // stored consensus offsets (|q_i| <= p_i - 1 needs |d_i| <= 2(p_i - 1)): its sum stays 0 and
    // the kernel's result is exactly 0.0 (computePatchGraph.cu:88-133 only counts such
    // candidates).  Such rows -- 36 % of the list at max_ps_dist = 2 -- get the largest key: they
    // sort behind every group and are not dispatched (the caller pre-zeroes d_aff).
    const bool far = abs(dz) > 2 * (G.pz - 1) || abs(dy) > 2 * (G.py - 1) || abs(dx) > 2 * (G.px - 1);
    keys[i] = far ? PPP_PAIR_KEY_FAR : ((lin << 18) | (apart << 17) | (dkey & 0x1FFFF));
}

hipError_t launch_pair_group_keys(const uint32_t *rows, uint64_t n, int64_t *keys, const Geo &G,
                                  hipStream_t s) {
    if (n == 0) return hipSuccess;
    PPP_GRID_CHECK((n + 255) / 256, 256);
    pair_group_keys_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(rows, n, keys, G);
    return hipGetLastError();
}

hipError_t launch_pair_keys(const uint32_t *rows, uint64_t n, int64_t *keys, const Geo &G,
                            hipStream_t s) {
    if (n == 0) return hipSuccess;
    PPP_GRID_CHECK((n + 255) / 256, 256);
    pair_keys_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(rows, n, keys, G);
    return hipGetLastError();
}

}  // namespace ppp
