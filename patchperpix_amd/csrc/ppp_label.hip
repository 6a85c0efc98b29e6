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

__device__ __forceinline__ uint32_t node_of(const Geo &G, const uint32_t *row) {
    return (uint32_t)(((long long)row[0] * G.Y + row[1]) * G.X + row[2]);
}

__global__ void label_init_kernel(const uint32_t *__restrict__ pairs, const float *__restrict__ aff,
                                  uint64_t n, uint32_t *parent, uint32_t *firstpos,
                                  uint32_t *cckey, const Geo G) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t u = node_of(G, pairs + i * 6), v = node_of(G, pairs + i * 6 + 3);
    // every node that occurs in any row gets a defined state (all writers agree)
    parent[u] = u; parent[v] = v;
    firstpos[u] = NONE; firstpos[v] = NONE;
    cckey[u] = NONE; cckey[v] = NONE;
}

__global__ void label_firstpos_kernel(const uint32_t *__restrict__ pairs,
                                      const float *__restrict__ aff, uint64_t n,
                                      uint32_t *firstpos, const Geo G) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || aff[i] == 0.0f) return;
    atomicMin(&firstpos[node_of(G, pairs + i * 6)], (uint32_t)(2 * i));
    atomicMin(&firstpos[node_of(G, pairs + i * 6 + 3)], (uint32_t)(2 * i + 1));
}

__device__ __forceinline__ uint32_t find_root(uint32_t *parent, uint32_t x) {
    // parents only ever decrease, so chasing them terminates
    uint32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        x = p;
        p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return x;
}

__global__ void label_union_kernel(const uint32_t *__restrict__ pairs,
                                   const float *__restrict__ aff, uint64_t n, uint32_t *parent,
                                   const Geo G) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !(aff[i] > 0.0f)) return;
    uint32_t a = node_of(G, pairs + i * 6), b = node_of(G, pairs + i * 6 + 3);
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

__global__ void label_key_kernel(const uint32_t *__restrict__ pairs,
                                 const float *__restrict__ aff, uint64_t n, uint32_t *parent,
                                 const uint32_t *__restrict__ firstpos, uint32_t *cckey,
                                 const Geo G) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !(aff[i] > 0.0f)) return;
    const uint32_t u = node_of(G, pairs + i * 6), v = node_of(G, pairs + i * 6 + 3);
    atomicMin(&cckey[find_root(parent, u)], firstpos[u]);
    atomicMin(&cckey[find_root(parent, v)], firstpos[v]);
}

__global__ void label_emit_kernel(const uint32_t *__restrict__ nodes, uint64_t n_nodes,
                                  uint32_t *parent, const uint32_t *__restrict__ cckey,
                                  const uint32_t *__restrict__ firstpos,
                                  uint32_t *__restrict__ out, const Geo G) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const uint32_t x = node_of(G, nodes + i * 3);
    // a node without any positive edge is its own root and its key stayed NONE
    out[i] = cckey[find_root(parent, x)];
}

__global__ void label_init_nodes_kernel(const uint32_t *__restrict__ nodes, uint64_t n_nodes,
                                        uint32_t *parent, uint32_t *firstpos, uint32_t *cckey,
                                        const Geo G) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const uint32_t x = node_of(G, nodes + i * 3);
    parent[x] = x; firstpos[x] = NONE; cckey[x] = NONE;
}

hipError_t launch_label(const uint32_t *pairs, const float *aff, uint64_t n,
                        const uint32_t *nodes, uint64_t n_nodes, uint32_t *node_key, void *work,
                        const Geo &G, hipStream_t s) {
    if (n_nodes == 0) return hipSuccess;
    uint32_t *parent = (uint32_t *)work;
    uint32_t *firstpos = parent + G.V;
    uint32_t *cckey = firstpos + G.V;
    const dim3 block(256);
    const dim3 gn((unsigned)((n_nodes + 255) / 256));
    // nodes that never occur in a row still need a defined state for the emit pass
    label_init_nodes_kernel<<<gn, block, 0, s>>>(nodes, n_nodes, parent, firstpos, cckey, G);
    if (n) {
        const dim3 grid((unsigned)((n + 255) / 256));
        label_init_kernel<<<grid, block, 0, s>>>(pairs, aff, n, parent, firstpos, cckey, G);
        label_firstpos_kernel<<<grid, block, 0, s>>>(pairs, aff, n, firstpos, G);
        label_union_kernel<<<grid, block, 0, s>>>(pairs, aff, n, parent, G);
        label_key_kernel<<<grid, block, 0, s>>>(pairs, aff, n, parent, firstpos, cckey, G);
    }
    label_emit_kernel<<<gn, block, 0, s>>>(nodes, n_nodes, parent, cckey, firstpos, node_key, G);
    return hipGetLastError();
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

hipError_t launch_paint(const void *pred, int dtype, const uint32_t *nodes,
                        const uint32_t *labels, uint64_t n, uint32_t *inst, const Geo &G,
                        hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint64_t total = n * (uint64_t)G.C;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == PPP_F16)
        paint_kernel<__half><<<grid, dim3(256), 0, s>>>((const __half *)pred, nodes, labels, n, inst, G.th_rn, G);
    else
        paint_kernel<float><<<grid, dim3(256), 0, s>>>((const float *)pred, nodes, labels, n, inst, G.th_rn, G);
    return hipGetLastError();
}

}  // namespace ppp
