// ppp_label.hip -- S6: connected components of the positive patch graph (lock-free
// union-find with global atomics) and instance painting.
//
// Reference: aff_patch_graph.py:31-40 (setAffgraph: rows with aff != 0 become edges),
// graph_to_labeling.py:50-54 (components of the aff > 0 sub-graph, enumerated by
// networkx) and :61-86 (paint, later components overwrite earlier ones).
//
// Component ORDER.  networkx enumerates components in the order in which their first
// member was inserted into the positive sub-graph, which itself follows the edge
// iteration of the full graph.  That order equals the ascending order of
//     key(component) = min over members x of firstpos(x),
//     firstpos(x)    = smallest 2*row+side at which x occurs among rows with aff != 0
// (oracle/ppp_oracle.py::connected_components is checked against networkx through the
// golden vectors).  The kernels below compute key(component) for both patches of every
// row; ranking the distinct keys is a tiny sort done by the caller.
//
// Nodes are patch centres, identified by their linear voxel index; the workspace holds
// three uint32 volumes (parent, firstpos, cckey), touched only at node positions.
#include "ppp_kernels.hpp"

namespace ppp {

static constexpr uint32_t NONE = 0xFFFFFFFFu;
// 64-bit order keys (2 * global row id + side); "none" is a value that survives a signed
// MIN all-reduce between ranks
static constexpr unsigned long long NONE64 = PPP_LABEL_NONE_KEY;

__device__ __forceinline__ uint32_t node_of(const Geo &G, const uint32_t *row) {
    return (uint32_t)(((long long)row[0] * G.Y + row[1]) * G.X + row[2]);
}

// Workspace: four volumes touched only at node positions
//   parent u32[V] | haspos u32[V] | firstpos u64[V] | cckey u64[V]
struct LabelWork {
    uint32_t *parent, *haspos;
    unsigned long long *firstpos, *cckey;
};
__host__ __device__ __forceinline__ LabelWork label_carve(void *work, long long V) {
    LabelWork W;
    W.parent = (uint32_t *)work;
    W.haspos = W.parent + V;
    W.firstpos = (unsigned long long *)(W.haspos + V);
    W.cckey = W.firstpos + V;
    return W;
}
size_t label_workspace_bytes(const Geo &G) { return (size_t)G.V * 24; }

__device__ __forceinline__ uint32_t find_root(uint32_t *parent, uint32_t x) {
    // parents only ever decrease, so chasing them terminates
    uint32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        x = p;
        p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return x;
}

__device__ __forceinline__ void unite(uint32_t *parent, uint32_t a, uint32_t b) {
    while (true) {
        a = find_root(parent, a);
        b = find_root(parent, b);
        if (a == b) break;
        const uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
        // hook the larger root under the smaller one; if hi stopped being a root the CAS
        // returns its current parent and the walk continues from there (progress does not
        // depend on how fresh the loads in find_root are)
        const uint32_t seen = atomicCAS(&parent[hi], hi, lo);
        if (seen == hi) break;
        a = seen;
        b = lo;
    }
}

// begin: every node of the caller's list gets a defined state
__global__ void label_begin_kernel(const uint32_t *__restrict__ nodes, uint64_t n_nodes,
                                   LabelWork W, const Geo G) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const uint32_t x = node_of(G, nodes + i * 3);
    W.parent[x] = x; W.haspos[x] = 0u; W.firstpos[x] = NONE64; W.cckey[x] = NONE64;
}

// add: a batch of rows with their GLOBAL row ids (gid == nullptr: ids are gid0 + i).
// Order-free: atomicMin for the first appearance among rows with aff != 0, lock-free unions
// for the rows with aff > 0.
__global__ void label_add_kernel(const uint32_t *__restrict__ pairs, const float *__restrict__ aff,
                                 const long long *__restrict__ gid, const long long gid0,
                                 uint64_t n, LabelWork W, const Geo G) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = aff[i];
    if (a == 0.0f) return;
    const uint32_t u = node_of(G, pairs + i * 6), v = node_of(G, pairs + i * 6 + 3);
    const unsigned long long g = (unsigned long long)(gid ? gid[i] : gid0 + (long long)i);
    atomicMin(&W.firstpos[u], 2ull * g);
    atomicMin(&W.firstpos[v], 2ull * g + 1ull);
    if (a > 0.0f) {
        W.haspos[u] = 1u; W.haspos[v] = 1u;
        unite(W.parent, u, v);
    }
}

// unions given as node pairs (linear voxel indices): merging another rank's forest
__global__ void label_union_edges_kernel(const long long *__restrict__ ea,
                                         const long long *__restrict__ eb, uint64_t n,
                                         LabelWork W) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (ea[i] != eb[i]) unite(W.parent, (uint32_t)ea[i], (uint32_t)eb[i]);
}

// finish: key(component) = min over its members WITH a positive edge of firstpos(member)
__global__ void label_key_kernel(const uint32_t *__restrict__ nodes, uint64_t n_nodes,
                                 LabelWork W, const Geo G) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const uint32_t x = node_of(G, nodes + i * 3);
    if (W.haspos[x]) atomicMin(&W.cckey[find_root(W.parent, x)], W.firstpos[x]);
}
__global__ void label_emit_kernel(const uint32_t *__restrict__ nodes, uint64_t n_nodes,
                                  LabelWork W, long long *__restrict__ out64,
                                  uint32_t *__restrict__ out32, const Geo G) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const uint32_t x = node_of(G, nodes + i * 3);
    // a node without any positive edge is its own root and its key stayed "none"
    const unsigned long long k = W.cckey[find_root(W.parent, x)];
    if (out64) out64[i] = (long long)k;
    if (out32) out32[i] = k == NONE64 ? NONE : (uint32_t)k;
}

hipError_t launch_label_begin(const uint32_t *nodes, uint64_t n_nodes, void *work, const Geo &G,
                              hipStream_t s) {
    if (n_nodes == 0) return hipSuccess;
    PPP_GRID_CHECK((n_nodes + 255) / 256, 256);
    label_begin_kernel<<<dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0, s>>>(
        nodes, n_nodes, label_carve(work, G.V), G);
    return hipGetLastError();
}
hipError_t launch_label_add(const uint32_t *pairs, const float *aff, const long long *gid,
                            long long gid0, uint64_t n, void *work, const Geo &G, hipStream_t s) {
    if (n == 0) return hipSuccess;
    PPP_GRID_CHECK((n + 255) / 256, 256);
    label_add_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(
        pairs, aff, gid, gid0, n, label_carve(work, G.V), G);
    return hipGetLastError();
}
hipError_t launch_label_union_edges(const long long *ea, const long long *eb, uint64_t n,
                                    void *work, const Geo &G, hipStream_t s) {
    if (n == 0) return hipSuccess;
    PPP_GRID_CHECK((n + 255) / 256, 256);
    label_union_edges_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(
        ea, eb, n, label_carve(work, G.V));
    return hipGetLastError();
}
hipError_t launch_label_finish(const uint32_t *nodes, uint64_t n_nodes, long long *key64,
                               uint32_t *key32, void *work, const Geo &G, hipStream_t s) {
    if (n_nodes == 0) return hipSuccess;
    const dim3 gn((unsigned)((n_nodes + 255) / 256)), block(256);
    const LabelWork W = label_carve(work, G.V);
    label_key_kernel<<<gn, block, 0, s>>>(nodes, n_nodes, W, G);
    label_emit_kernel<<<gn, block, 0, s>>>(nodes, n_nodes, W, key64, key32, G);
    return hipGetLastError();
}

// one-shot form (all rows at once, row ids = positions)
hipError_t launch_label(const uint32_t *pairs, const float *aff, uint64_t n,
                        const uint32_t *nodes, uint64_t n_nodes, uint32_t *node_key, void *work,
                        const Geo &G, hipStream_t s) {
    if (n_nodes == 0) return hipSuccess;
    hipError_t e;
    if ((e = launch_label_begin(nodes, n_nodes, work, G, s)) != hipSuccess) return e;
    // rows may mention nodes outside the caller's list: give them a state as well
    if (n) {
        const uint32_t *ends = pairs;
        if ((e = launch_label_begin(ends, 2 * n, work, G, s)) != hipSuccess) return e;
        if ((e = launch_label_add(pairs, aff, nullptr, 0, n, work, G, s)) != hipSuccess) return e;
    }
    return launch_label_finish(nodes, n_nodes, nullptr, node_key, work, G, s);
}

// ---- paint ----------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
    paint_kernel(const T *__restrict__ pred, const uint32_t *__restrict__ nodes,
                 const uint32_t *__restrict__ labels, uint64_t n, uint32_t *inst, float th_f32,
                 const Geo G) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * (uint64_t)G.C) return;
    const uint64_t k = t / G.C;
    const int r = (int)(t % G.C);
    const uint32_t lab = labels[k];
    if (lab == 0) return;
    const int cz = (int)nodes[k * 3], cy = (int)nodes[k * 3 + 1], cx = (int)nodes[k * 3 + 2];
    // NumPy compares the float32 patch with float32(patch_threshold)
    if (!(ldf(pred, (long long)r * G.V + vox(G, cz, cy, cx)) > th_f32)) return;
    const int z = cz + r / (G.py * G.px) - G.rz;
    const int y = cy + (r / G.px) % G.py - G.ry;
    const int x = cx + r % G.px - G.rx;
    if (z < 0 || z >= G.Z || y < 0 || y >= G.Y || x < 0 || x >= G.X) return;
    atomicMax(&inst[vox(G, z, y, x)], lab);
}

// the same painting with the patches given as a TABLE: rows[k][r] = channel r at node k
// (blockwise driver: the patches of the global graph's nodes are gathered chunk by chunk from the
// prediction store; no dense (C, Z, Y, X) block exists)
template <typename T>
__global__ void __launch_bounds__(256)
    paint_rows_kernel(const T *__restrict__ rows, const uint32_t *__restrict__ nodes,
                      const uint32_t *__restrict__ labels, uint64_t n, uint32_t *inst, float th_f32,
                      const Geo G) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * (uint64_t)G.C) return;
    const uint64_t k = t / G.C;
    const int r = (int)(t % G.C);
    const uint32_t lab = labels[k];
    if (lab == 0) return;
    if (!(ldf(rows, (long long)t) > th_f32)) return;
    const int cz = (int)nodes[k * 3], cy = (int)nodes[k * 3 + 1], cx = (int)nodes[k * 3 + 2];
    const int z = cz + r / (G.py * G.px) - G.rz;
    const int y = cy + (r / G.px) % G.py - G.ry;
    const int x = cx + r % G.px - G.rx;
    if (z < 0 || z >= G.Z || y < 0 || y >= G.Y || x < 0 || x >= G.X) return;
    atomicMax(&inst[vox(G, z, y, x)], lab);
}

hipError_t launch_paint_rows(const void *rows, int dtype, const uint32_t *nodes, const uint32_t *labels,
                             uint64_t n, uint32_t *inst, const Geo &G, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint64_t per = ((1ull << 31) / (uint64_t)G.C) & ~255ull;
    for (uint64_t k0 = 0; k0 < n; k0 += per) {
        const uint64_t m = n - k0 < per ? n - k0 : per;
        const dim3 grid((unsigned)((m * (uint64_t)G.C + 255) / 256));
        if (dtype == PPP_F16)
            paint_rows_kernel<__half><<<grid, dim3(256), 0, s>>>((const __half *)rows + k0 * G.C, nodes + k0 * 3, labels + k0, m, inst, G.th_rn, G);
        else
            paint_rows_kernel<float><<<grid, dim3(256), 0, s>>>((const float *)rows + k0 * G.C, nodes + k0 * 3, labels + k0, m, inst, G.th_rn, G);
    }
    return hipGetLastError();
}

hipError_t launch_paint(const void *pred, int dtype, const uint32_t *nodes,
                        const uint32_t *labels, uint64_t n, uint32_t *inst, const Geo &G,
                        hipStream_t s) {
    if (n == 0) return hipSuccess;
    // a thread per (node, patch pixel): chunks of nodes that stay below the 2^32 grid limit
    const uint64_t per = ((1ull << 31) / (uint64_t)G.C) & ~255ull;
    for (uint64_t k0 = 0; k0 < n; k0 += per) {
        const uint64_t m = n - k0 < per ? n - k0 : per;
        const uint64_t total = m * (uint64_t)G.C;
        const dim3 grid((unsigned)((total + 255) / 256));
        if (dtype == PPP_F16)
            paint_kernel<__half><<<grid, dim3(256), 0, s>>>((const __half *)pred, nodes + k0 * 3, labels + k0, m, inst, G.th_rn, G);
        else
            paint_kernel<float><<<grid, dim3(256), 0, s>>>((const float *)pred, nodes + k0 * 3, labels + k0, m, inst, G.th_rn, G);
    }
    return hipGetLastError();
}

}  // namespace ppp
