// ppp_sort.hip -- the order-defining host stages that are sorts, on the device (rocPRIM radix
// sort / select inside the library: a caller of the C ABI needs no tensor framework for them).
//
//   rank_order   all_patches + rank_patches_by_score (vote_instances.py:276,286-287,
//                ranked_patches.py:21-30): interior foreground voxels in raster order, stably
//                sorted by score descending.
//   mws_edges    the edge list graph_mws.mws walks (graph_mws.py:17-26): rows with aff != 0 in
//                networkx's edge order, stably sorted by |aff| descending.
//   group_rows   grouping of pair rows by patch A for the per-patch S5 kernel.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "ppp_kernels.hpp"

namespace ppp {

static size_t up256s(size_t v) { return (v + 255) / 256 * 256; }

// ---------------------------------------------------------------------------------------------
// rank order
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    rank_flags_kernel(const uint8_t *__restrict__ fg, uint8_t *__restrict__ flags, const Geo G) {
    const long long v = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (v >= G.V) return;
    const int x = (int)(v % G.X), y = (int)((v / G.X) % G.Y), z = (int)(v / ((long long)G.X * G.Y));
    flags[v] = (fg[v] != 0 && interior(G, z, y, x)) ? 1 : 0;
}
// sort key of a score: the comparison-based sort of the reference treats -0.0 and +0.0 as equal
// (a radix sort would not): adding +0.0 maps both to +0.0 and changes nothing else
__global__ void __launch_bounds__(256)
    rank_keys_kernel(const float *__restrict__ score, const int32_t *__restrict__ idx, long long n,
                     float *__restrict__ keys) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < n) keys[i] = score[idx[i]] + 0.0f;
}
__global__ void __launch_bounds__(256)
    rank_out_kernel(const float *__restrict__ score, const int32_t *__restrict__ idx, long long n,
                    long long *__restrict__ lin, float *__restrict__ out_score) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t v = idx[i];
    lin[i] = v;
    if (out_score) out_score[i] = score[v];
}

struct RankWork {
    uint8_t *flags;       // [V]
    int32_t *idx, *idx2;  // [V]
    float *keys, *keys2;  // [V]
    long long *count;     // [1]
    void *temp;
    size_t temp_bytes;
};
static size_t rank_temp_bytes(long long V) {
    size_t a = 0, b = 0;
    (void)rocprim::select(nullptr, a, rocprim::counting_iterator<int32_t>(0), (uint8_t *)nullptr,
                          (int32_t *)nullptr, (long long *)nullptr, (size_t)V, (hipStream_t)0);
    (void)rocprim::radix_sort_pairs_desc(nullptr, b, (float *)nullptr, (float *)nullptr, (int32_t *)nullptr,
                                         (int32_t *)nullptr, (size_t)V, 0, 32, (hipStream_t)0);
    return up256s(a > b ? a : b);
}
size_t rank_order_workspace_bytes(const Geo &G) {
    return up256s((size_t)G.V) + 4 * up256s((size_t)G.V * 4) + 256 + rank_temp_bytes(G.V);
}
static RankWork carve_rank(void *work, const Geo &G) {
    RankWork W;
    char *p = (char *)work;
    W.flags = (uint8_t *)p;  p += up256s((size_t)G.V);
    W.idx = (int32_t *)p;    p += up256s((size_t)G.V * 4);
    W.idx2 = (int32_t *)p;   p += up256s((size_t)G.V * 4);
    W.keys = (float *)p;     p += up256s((size_t)G.V * 4);
    W.keys2 = (float *)p;    p += up256s((size_t)G.V * 4);
    W.count = (long long *)p; p += 256;
    W.temp = p;
    W.temp_bytes = rank_temp_bytes(G.V);
    return W;
}

// lin i64 [>= count], out_score f32 [>= count] (may be NULL).  Synchronises (the count goes to the host).
hipError_t run_rank_order(const float *score, const uint8_t *fg, long long *lin, float *out_score,
                          long long *n_out, void *work, const Geo &G, hipStream_t s) {
    PPP_GRID_CHECK((G.V + 255) / 256, 256);
    if (G.V >= (1ll << 31)) return hipErrorInvalidValue;   // voxel indices travel as int32
    RankWork W = carve_rank(work, G);
    const dim3 vgrid((unsigned)((G.V + 255) / 256)), block(256);
    rank_flags_kernel<<<vgrid, block, 0, s>>>(fg, W.flags, G);
    hipError_t e;
    size_t tb = W.temp_bytes;
    if ((e = rocprim::select(W.temp, tb, rocprim::counting_iterator<int32_t>(0), W.flags, W.idx, W.count,
                             (size_t)G.V, s)) != hipSuccess) return e;
    long long n = 0;
    if ((e = hipMemcpyAsync(&n, W.count, 8, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
    *n_out = n;
    if (n == 0) return hipSuccess;
    const dim3 ngrid((unsigned)((n + 255) / 256));
    rank_keys_kernel<<<ngrid, block, 0, s>>>(score, W.idx, n, W.keys);
    tb = W.temp_bytes;
    if ((e = rocprim::radix_sort_pairs_desc(W.temp, tb, W.keys, W.keys2, W.idx, W.idx2, (size_t)n, 0, 32, s)) != hipSuccess)
        return e;
    rank_out_kernel<<<ngrid, block, 0, s>>>(score, W.idx2, n, lin, out_score);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// mutex watershed edge list
// ---------------------------------------------------------------------------------------------
// networkx yields an edge at the endpoint that was inserted first, in the order the edges of
// that node were inserted; nodes are inserted in order of first appearance among the rows with
// aff != 0 (setAffgraph, aff_patch_graph.py:31-40).  Hence the edge order is (first appearance
// of the earlier endpoint, row) -- and first appearances are what atomicMin(2*row+side) gives.
__global__ void __launch_bounds__(256)
    mws_nodes_kernel(const uint32_t *__restrict__ nodes, long long n, int32_t *__restrict__ id_vol,
                     const Geo G) {
    const long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (k < n) id_vol[vox(G, (int)nodes[k * 3], (int)nodes[k * 3 + 1], (int)nodes[k * 3 + 2])] = (int32_t)k;
}
__global__ void __launch_bounds__(256)
    mws_first_kernel(const uint32_t *__restrict__ rows, const float *__restrict__ aff, long long n,
                     const int32_t *__restrict__ id_vol, unsigned long long *__restrict__ firstpos,
                     int32_t *__restrict__ eu, int32_t *__restrict__ ev,
                     unsigned long long *__restrict__ counters, const Geo G) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    bool live = false, bad = false;
    if (i < n) {
        const uint32_t *r = rows + i * 6;
        const int32_t a = id_vol[vox(G, (int)r[0], (int)r[1], (int)r[2])];
        const int32_t b = id_vol[vox(G, (int)r[3], (int)r[4], (int)r[5])];
        eu[i] = a;
        ev[i] = b;
        live = aff[i] != 0.0f;          // setAffgraph skips exact zeros
        bad = live && (a < 0 || b < 0);
        if (live && !bad) {
            atomicMin(&firstpos[a], 2ull * (unsigned long long)i);
            atomicMin(&firstpos[b], 2ull * (unsigned long long)i + 1ull);
        }
    }
    const unsigned long long m = __ballot(live), mb = __ballot(bad);
    if ((threadIdx.x & 63) == 0) {
        if (m) atomicAdd(&counters[0], (unsigned long long)__popcll(m));
        if (mb) atomicAdd(&counters[1], (unsigned long long)__popcll(mb));
    }
}
__global__ void __launch_bounds__(256)
    mws_keys_kernel(const float *__restrict__ aff, long long n, const int32_t *__restrict__ eu,
                    const int32_t *__restrict__ ev, const unsigned long long *__restrict__ firstpos,
                    unsigned long long *__restrict__ key2, uint32_t *__restrict__ row_id) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    row_id[i] = (uint32_t)i;
    unsigned long long k = 2ull * (unsigned long long)n;     // rows that are not edges: after every edge
    if (aff[i] != 0.0f && eu[i] >= 0 && ev[i] >= 0) {
        const unsigned long long fa = firstpos[eu[i]], fb = firstpos[ev[i]];
        k = fa < fb ? fa : fb;
    }
    key2[i] = k;
}
__global__ void __launch_bounds__(256)
    mws_wkeys_kernel(const float *__restrict__ aff, const uint32_t *__restrict__ row_id, long long n,
                     uint32_t *__restrict__ wkey) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    // |aff| as an unsigned key: for non-negative floats the bit pattern orders like the value;
    // zeros (rows that are not edges) sort last in descending order
    if (i < n) wkey[i] = __float_as_uint(fabsf(aff[row_id[i]]));
}
__global__ void __launch_bounds__(256)
    mws_emit_kernel(const float *__restrict__ aff, const uint32_t *__restrict__ row_id, long long n_edges,
                    const int32_t *__restrict__ eu, const int32_t *__restrict__ ev,
                    int32_t *__restrict__ out_u, int32_t *__restrict__ out_v) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= n_edges) return;
    const uint32_t r = row_id[i];
    out_u[i] = eu[r];
    // graph_mws.py:20-23: a > 0 attractive, everything else repulsive; flag in bit 31
    out_v[i] = ev[r] | (aff[r] > 0.0f ? (int32_t)0x80000000 : 0);
}

static size_t mws_temp_bytes(long long n) {
    size_t a = 0, b = 0;
    (void)rocprim::radix_sort_pairs(nullptr, a, (unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                    (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n, 0, 64, (hipStream_t)0);
    (void)rocprim::radix_sort_pairs_desc(nullptr, b, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                         (uint32_t *)nullptr, (size_t)n, 0, 32, (hipStream_t)0);
    return up256s(a > b ? a : b);
}
size_t mws_edges_workspace_bytes(long long n_rows, long long n_nodes, const Geo &G) {
    const size_t n = (size_t)(n_rows > 0 ? n_rows : 1);
    return up256s((size_t)G.V * 4) + up256s((size_t)(n_nodes > 0 ? n_nodes : 1) * 8) + 2 * up256s(n * 4) +
           2 * up256s(n * 8) + 4 * up256s(n * 4) + 256 + mws_temp_bytes((long long)n);
}

// out_u / out_v int32 [n_rows] (device): the edges in the order graph_mws.mws visits them, node
// numbers = positions in `nodes`, bit 31 of out_v = attractive.  *n_edges = rows with aff != 0.
// Returns hipErrorInvalidValue when a row with aff != 0 names a voxel that is not in `nodes`.
hipError_t run_mws_edges(const uint32_t *rows, const float *aff, long long n_rows, const uint32_t *nodes,
                         long long n_nodes, int32_t *out_u, int32_t *out_v, long long *n_edges, void *work,
                         const Geo &G, hipStream_t s) {
    *n_edges = 0;
    if (n_rows <= 0) return hipSuccess;
    if (n_rows >= (1ll << 32) - 1 || n_nodes >= (1ll << 31)) return hipErrorInvalidValue;
    PPP_GRID_CHECK((n_rows + 255) / 256, 256);
    const size_t n = (size_t)n_rows;
    char *p = (char *)work;
    int32_t *id_vol = (int32_t *)p;                      p += up256s((size_t)G.V * 4);
    unsigned long long *firstpos = (unsigned long long *)p; p += up256s((size_t)(n_nodes > 0 ? n_nodes : 1) * 8);
    int32_t *eu = (int32_t *)p;                          p += up256s(n * 4);
    int32_t *ev = (int32_t *)p;                          p += up256s(n * 4);
    unsigned long long *key2 = (unsigned long long *)p;  p += up256s(n * 8);
    unsigned long long *key2b = (unsigned long long *)p; p += up256s(n * 8);
    uint32_t *rid = (uint32_t *)p;                       p += up256s(n * 4);
    uint32_t *rid2 = (uint32_t *)p;                      p += up256s(n * 4);
    uint32_t *wk = (uint32_t *)p;                        p += up256s(n * 4);
    uint32_t *wk2 = (uint32_t *)p;                       p += up256s(n * 4);
    unsigned long long *counters = (unsigned long long *)p; p += 256;
    void *temp = p;
    const size_t temp_bytes = mws_temp_bytes(n_rows);
    hipError_t e;
    if ((e = hipMemsetAsync(id_vol, 0xFF, (size_t)G.V * 4, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(firstpos, 0xFF, (size_t)(n_nodes > 0 ? n_nodes : 1) * 8, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(counters, 0, 16, s)) != hipSuccess) return e;
    const dim3 block(256), rgrid((unsigned)((n_rows + 255) / 256));
    if (n_nodes > 0)
        mws_nodes_kernel<<<dim3((unsigned)((n_nodes + 255) / 256)), block, 0, s>>>(nodes, n_nodes, id_vol, G);
    mws_first_kernel<<<rgrid, block, 0, s>>>(rows, aff, n_rows, id_vol, firstpos, eu, ev, counters, G);
    mws_keys_kernel<<<rgrid, block, 0, s>>>(aff, n_rows, eu, ev, firstpos, key2, rid);
    // stable by (first appearance of the earlier endpoint), rows arrive in row order
    unsigned bits = 2;                                       // keys are <= 2 * n_rows
    while (bits < 64 && (2ull * (unsigned long long)n_rows) >> bits) ++bits;
    size_t tb = temp_bytes;
    if ((e = rocprim::radix_sort_pairs(temp, tb, key2, key2b, rid, rid2, n, 0, bits, s)) != hipSuccess)
        return e;
    mws_wkeys_kernel<<<rgrid, block, 0, s>>>(aff, rid2, n_rows, wk);
    tb = temp_bytes;
    if ((e = rocprim::radix_sort_pairs_desc(temp, tb, wk, wk2, rid2, rid, n, 0, 32, s)) != hipSuccess) return e;
    unsigned long long h[2] = {0, 0};
    if ((e = hipMemcpyAsync(h, counters, 16, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
    if (h[1] != 0) return hipErrorInvalidValue;
    *n_edges = (long long)h[0];
    if (h[0])
        mws_emit_kernel<<<dim3((unsigned)((h[0] + 255) / 256)), block, 0, s>>>(aff, rid, (long long)h[0], eu, ev,
                                                                              out_u, out_v);
    return hipGetLastError();
}

}  // namespace ppp
