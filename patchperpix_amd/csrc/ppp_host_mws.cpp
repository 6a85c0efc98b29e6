// ppp_host_mws.cpp -- mutex watershed on the patch graph, host side like the reference
// (PatchPerPix/vote_instances/graph_mws.py:7-85 on the graph built by setAffgraph,
// aff_patch_graph.py:31-40).
//
// Same decisions as the reference, without its O(E * |mutex|) scans:
//   * nodes are numbered by first appearance among the rows with aff != 0, edges are visited in
//     networkx's order (node-major in insertion order, neighbours in insertion order, each edge
//     once at its first-visited endpoint; a repeated row overwrites the value of its edge) and
//     stably sorted by |aff| descending (graph_mws.py:23-29);
//   * component membership lives in a union-find whose roots carry the reference's component id;
//     a new component gets max(ids in use) + 1, a merge keeps the smaller id (:37-38, :66-72);
//   * "is there a mutex edge between these components" (:46-48, :59-62) is one lookup in a hash
//     set of (root, root) keys that is rewritten, small-to-large, when components merge;
//   * the output order is the creation order of the ids -- a re-issued id keeps its first
//     position and a component emptied by a merge stays in the list (:79-82) -- so the instance
//     label of a node is 1 + position of its component's id in that order.
#include <stdint.h>

#include <algorithm>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/ppp_mi355x.h"

namespace {

struct Edge {
    int32_t a, b;
    float w;        // |aff|
    int8_t attractive;
};

struct Dsu {
    std::vector<int32_t> parent;
    int32_t find(int32_t x) {
        while (parent[x] != x) {
            parent[x] = parent[parent[x]];
            x = parent[x];
        }
        return x;
    }
};

// Open-addressing set of 64-bit keys (linear probing, tombstones), grown by doubling.
struct KeySet {
    static constexpr uint64_t EMPTY = ~0ull, TOMB = ~0ull - 1;
    std::vector<uint64_t> slot;
    size_t used = 0, live = 0;     // used: non-EMPTY slots (incl. tombstones)
    explicit KeySet(size_t cap = 1024) : slot(cap, EMPTY) {}
    static size_t hash(uint64_t k) {
        k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
        return (size_t)k;
    }
    void grow() {
        std::vector<uint64_t> old;
        old.swap(slot);
        slot.assign(live * 4 + 1024 > old.size() ? old.size() * 2 : old.size(), EMPTY);
        used = live = 0;
        for (uint64_t k : old)
            if (k != EMPTY && k != TOMB) insert(k);
    }
    bool contains(uint64_t k) const {
        const size_t m = slot.size() - 1;
        for (size_t i = hash(k) & m;; i = (i + 1) & m) {
            if (slot[i] == k) return true;
            if (slot[i] == EMPTY) return false;
        }
    }
    // true if the key was not there
    bool insert(uint64_t k) {
        if ((used + 1) * 2 > slot.size()) grow();
        const size_t m = slot.size() - 1;
        size_t tomb = (size_t)-1;
        for (size_t i = hash(k) & m;; i = (i + 1) & m) {
            if (slot[i] == k) return false;
            if (slot[i] == TOMB && tomb == (size_t)-1) tomb = i;
            if (slot[i] == EMPTY) {
                if (tomb != (size_t)-1) { slot[tomb] = k; }
                else { slot[i] = k; ++used; }
                ++live;
                return true;
            }
        }
    }
    void erase(uint64_t k) {
        const size_t m = slot.size() - 1;
        for (size_t i = hash(k) & m;; i = (i + 1) & m) {
            if (slot[i] == k) { slot[i] = TOMB; --live; return; }
            if (slot[i] == EMPTY) return;
        }
    }
};

// The loop of graph_mws.mws (:31-77) over edges that are already in visiting order.
//   eu / ev [n_edges]: node numbers, bit 31 of ev = attractive; label [n_nodes] out: 1 + position
//   of the node's component id in creation order, 0 = the node ended in no component.
// Every node is a union-find element from the start (an unassigned node is a singleton without
// a component id).  "Is there a mutex edge between two clusters" (:46-48, :59-62) is ONE lookup:
// the set `mutex` holds a key (smaller root, larger root) for every pair of CURRENT clusters with
// a repulsive edge between them.  It is kept exact under merges by rewriting the keys of the
// absorbed cluster -- each root carries the list of the clusters it has a key with (entries may
// name a cluster by any of its former roots; duplicates are harmless), the cluster with the
// shorter list is the one absorbed, so a key is rewritten O(log n) times.  (Round 3 walked the
// shorter of two endpoint lists with a find per entry: 140 entries per check on a graph with half
// of its edges repulsive -- BASELINE config [4] with a random decoder -- 705 ns per edge; this
// form: see tools/time_mws.py.)  A mutex edge inside one cluster can never matter again (clusters
// only grow) and is dropped.
void watershed(const int32_t *eu, const int32_t *ev, int64_t n_edges, int32_t N, int32_t *label,
               int64_t *issued) {
    Dsu dsu;
    dsu.parent.resize((size_t)N);
    for (int32_t i = 0; i < N; ++i) dsu.parent[i] = i;
    std::vector<int32_t> cc_of_root((size_t)N, 0);      // component id carried by a root (0: none)
    std::vector<std::vector<int32_t>> nbrs((size_t)N);  // clusters this root has a mutex key with
    KeySet mutex(1 << 16);
    // ids currently held by some node: only "the largest one" is ever asked (:37-38), so a flag
    // per id and a maximum that is walked down lazily when its id dies (amortised O(1), no tree)
    std::vector<uint8_t> alive((size_t)N + 2, 0);
    int32_t max_alive = 0;
    std::vector<int32_t> created;                       // ids in creation order (first issue)
    std::vector<uint8_t> ever;                          // id was issued before (ids are <= N)
    ever.assign((size_t)N + 2, 0);
    auto key = [](int32_t a, int32_t b) -> uint64_t {
        const uint32_t lo = (uint32_t)std::min(a, b), hi = (uint32_t)std::max(a, b);
        return ((uint64_t)lo << 32) | hi;
    };
    auto has_mutex = [&](int32_t r0, int32_t r1) { return mutex.contains(key(r0, r1)); };
    // the two clusters become one; returns its root (the one with the longer list)
    std::vector<int32_t> n_nbrs((size_t)N, 0);          // list lengths, next to each other (the
                                                        // vector headers are 24 bytes apiece)
    auto unite = [&](int32_t ra, int32_t rb) -> int32_t {
        int32_t keep = ra, drop = rb;
        if (n_nbrs[keep] < n_nbrs[drop]) std::swap(keep, drop);
        if (n_nbrs[drop]) {
            std::vector<int32_t> &kl = nbrs[keep], &dl = nbrs[drop];
            // (find() of the neighbours BEFORE the link: keep and drop are still both roots)
            for (int32_t x : dl) {
                const int32_t p = dsu.find(x);
                if (p == keep || p == drop) continue;          // becomes internal
                mutex.erase(key(drop, p));
                if (mutex.insert(key(keep, p))) kl.push_back(p);
            }
            mutex.erase(key(keep, drop));
            std::vector<int32_t>().swap(dl);
            n_nbrs[keep] = (int32_t)kl.size();
            n_nbrs[drop] = 0;
        }
        dsu.parent[drop] = keep;
        return keep;
    };
    constexpr int64_t AHEAD = 24;          // the edge list is known: fetch the nodes' entries early
    for (int64_t i = 0; i < n_edges; ++i) {
        if (i + AHEAD < n_edges) {
            const int32_t pa = eu[i + AHEAD], pb = ev[i + AHEAD] & 0x7FFFFFFF;
            __builtin_prefetch(&dsu.parent[pa]);
            __builtin_prefetch(&dsu.parent[pb]);
        }
        const int32_t a = eu[i], b = ev[i] & 0x7FFFFFFF;
        const int32_t r0 = dsu.find(a), r1 = dsu.find(b);
        if (ev[i] < 0) {                                  // attractive
            const int32_t c0 = cc_of_root[r0], c1 = cc_of_root[r1];
            if (c0 == 0 && c1 == 0) {
                // both unassigned (:36-42): a new component, id = max id in use + 1
                while (max_alive > 0 && !alive[(size_t)max_alive]) --max_alive;
                const int32_t id = max_alive + 1;
                const int32_t r = r0 != r1 ? unite(r0, r1) : r0;
                cc_of_root[r] = id;
                alive[(size_t)id] = 1;
                max_alive = id;
                if (!ever[(size_t)id]) { ever[(size_t)id] = 1; created.push_back(id); }
            } else if (c0 == 0 || c1 == 0) {
                // the unassigned node joins unless a mutex edge links it to the component (:44-56)
                if (!has_mutex(r0, r1)) {
                    const int32_t id = c0 == 0 ? c1 : c0;
                    cc_of_root[unite(r0, r1)] = id;
                }
            } else if (c0 != c1) {
                // two components merge into the smaller id unless a mutex edge links them (:57-71)
                if (!has_mutex(r0, r1)) {
                    const int32_t keep = std::min(c0, c1), drop = std::max(c0, c1);
                    cc_of_root[unite(r0, r1)] = keep;
                    alive[(size_t)drop] = 0;
                }
            }
        } else if (r0 != r1) {                            // repulsive (:76-77)
            if (mutex.insert(key(r0, r1))) {
                nbrs[r0].push_back(r1);
                nbrs[r1].push_back(r0);
                ++n_nbrs[r0];
                ++n_nbrs[r1];
            }
        }
    }
    std::vector<int32_t> label_of_id((size_t)N + 2, 0);
    for (size_t i = 0; i < created.size(); ++i) label_of_id[(size_t)created[i]] = (int32_t)i + 1;
    for (int32_t n = 0; n < N; ++n) label[n] = label_of_id[(size_t)cc_of_root[dsu.find(n)]];
    *issued = (int64_t)created.size();
}

}  // namespace

extern "C" {

// ppp_host_mws_sorted: the watershed over an edge list that is already in visiting order (made
// on the device by ppp_mws_edges).  labels int32 [n_nodes] out; returns the number of ids issued.
int64_t ppp_host_mws_sorted(const int32_t *eu, const int32_t *ev, int64_t n_edges, int64_t n_nodes,
                            int32_t *labels) {
    int64_t issued = 0;
    if (n_nodes <= 0) return 0;
    watershed(eu, ev, n_edges, (int32_t)n_nodes, labels, &issued);
    return issued;
}

// pairs u32 [n_rows][6] (A zyx, B zyx), aff f32 [n_rows], vol = volume shape (for node keys).
// out_nodes int32 [cap][3] / out_labels int32 [cap]: every node of the graph (first-appearance
// order) with its label, 0 for nodes the watershed left unassigned.  *n_labels = number of
// labels issued (including emptied components).  Returns the number of nodes, -1 if cap is
// too small, -2 for a coordinate outside vol.
int64_t ppp_host_mws(const uint32_t *pairs, const float *aff, int64_t n_rows, const int32_t *vol,
                     int32_t *out_nodes, int32_t *out_labels, int64_t cap, int64_t *n_labels) {
    const int64_t Y = vol[1], X = vol[2];
    auto lin = [&](const uint32_t *c) { return ((int64_t)c[0] * Y + c[1]) * X + c[2]; };

    // ---- nodes and (deduplicated) edges in insertion order --------------------------------
    // voxel -> node id: a direct table for volumes up to 2^28 voxels, a hash map beyond
    const int64_t V = (int64_t)vol[0] * Y * X;
    const bool direct = V <= (1ll << 28);
    std::vector<int32_t> table(direct ? (size_t)V : 0, -1);
    std::unordered_map<int64_t, int32_t> node_id;
    std::vector<int64_t> node_key;
    struct Row { int32_t u, v; float a; };
    std::vector<Row> rows;
    rows.reserve((size_t)n_rows);
    for (int64_t i = 0; i < n_rows; ++i) {
        if (aff[i] == 0.0f) continue;          // setAffgraph skips exact zeros (NaN is kept)
        int32_t id[2];
        for (int s = 0; s < 2; ++s) {
            const int64_t k = lin(pairs + i * 6 + 3 * s);
            if (direct) {
                if (k < 0 || k >= V) return -2;
                if (table[k] < 0) {
                    table[k] = (int32_t)node_key.size();
                    node_key.push_back(k);
                }
                id[s] = table[k];
            } else {
                auto it = node_id.find(k);
                if (it == node_id.end()) {
                    it = node_id.emplace(k, (int32_t)node_key.size()).first;
                    node_key.push_back(k);
                }
                id[s] = it->second;
            }
        }
        rows.push_back({id[0], id[1], aff[i]});
    }
    std::vector<int32_t>().swap(table);
    const int32_t N = (int32_t)node_key.size();
    if (N > cap) return -1;
    // a repeated (unordered) node pair keeps its first position and takes the last value
    {
        std::vector<uint64_t> key(rows.size());
        for (size_t i = 0; i < rows.size(); ++i) {
            const uint32_t lo = (uint32_t)std::min(rows[i].u, rows[i].v), hi = (uint32_t)std::max(rows[i].u, rows[i].v);
            key[i] = ((uint64_t)lo << 32) | hi;
        }
        std::vector<uint64_t> sorted(key);
        std::sort(sorted.begin(), sorted.end());
        if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) {
            std::unordered_map<uint64_t, size_t> first;
            std::vector<Row> uniq;
            for (size_t i = 0; i < rows.size(); ++i) {
                auto it = first.find(key[i]);
                if (it == first.end()) {
                    first.emplace(key[i], uniq.size());
                    uniq.push_back(rows[i]);
                } else {
                    uniq[it->second].a = rows[i].a;
                }
            }
            rows.swap(uniq);
        }
    }
    // adjacency in insertion order (CSR), then networkx's edge order
    std::vector<int64_t> start((size_t)N + 1, 0);
    for (const Row &r : rows) {
        ++start[r.u + 1];
        if (r.u != r.v) ++start[r.v + 1];
    }
    for (int32_t n = 0; n < N; ++n) start[n + 1] += start[n];
    std::vector<int32_t> nbr((size_t)start[N]);
    std::vector<float> nbr_a((size_t)start[N]);
    {
        std::vector<int64_t> fill(start.begin(), start.end() - 1);
        for (const Row &r : rows) {
            nbr[fill[r.u]] = r.v; nbr_a[fill[r.u]++] = r.a;
            if (r.u != r.v) { nbr[fill[r.v]] = r.u; nbr_a[fill[r.v]++] = r.a; }
        }
    }
    std::vector<Edge> edges;
    edges.reserve(rows.size());
    for (int32_t n = 0; n < N; ++n)
        for (int64_t j = start[n]; j < start[n + 1]; ++j)
            if (nbr[j] >= n) {
                const float a = nbr_a[j];
                // graph_mws.py:17-21: a > 0 attractive, everything else (incl. NaN) repulsive
                edges.push_back(a > 0 ? Edge{n, nbr[j], a, 1} : Edge{n, nbr[j], -a, -1});
            }
    std::vector<int32_t>().swap(nbr);
    std::vector<float>().swap(nbr_a);
    std::vector<Row>().swap(rows);
    std::stable_sort(edges.begin(), edges.end(), [](const Edge &x, const Edge &y) { return x.w > y.w; });

    // ---- the watershed ---------------------------------------------------------------------
    std::vector<int32_t> eu(edges.size()), ev(edges.size());
    for (size_t i = 0; i < edges.size(); ++i) {
        eu[i] = edges[i].a;
        ev[i] = edges[i].b | (edges[i].attractive == 1 ? (int32_t)0x80000000 : 0);
    }
    std::vector<Edge>().swap(edges);
    std::vector<int32_t> label((size_t)N);
    int64_t issued = 0;
    watershed(eu.data(), ev.data(), (int64_t)eu.size(), N, label.data(), &issued);

    // ---- labels ------------------------------------------------------------------------------
    for (int32_t n = 0; n < N; ++n) {
        const int64_t k = node_key[n];
        out_nodes[3 * (int64_t)n + 0] = (int32_t)(k / (Y * X));
        out_nodes[3 * (int64_t)n + 1] = (int32_t)((k / X) % Y);
        out_nodes[3 * (int64_t)n + 2] = (int32_t)(k % X);
        out_labels[n] = label[(size_t)n];
    }
    if (n_labels) *n_labels = issued;
    return N;
}

}  // extern "C"
