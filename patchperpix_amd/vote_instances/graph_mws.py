"""Mutex watershed on the patch graph (reference: PatchPerPix/vote_instances/graph_mws.py:7-85).

Host stage, as in the reference.  ``mws_from_pairs`` works on the (pairs, aff) arrays the
patch-graph kernel produces and reproduces the reference's node numbering, edge order,
component-id re-issue and empty-component quirks; the per-edge scan over the whole mutex
set (O(E*|mutex|) in the reference) is replaced by per-component mutex sets with
small-to-large merging, which answers exactly the same queries.
"""
import numpy as np


def graph_edge_order(pairs, aff):
    """Nodes (first appearance among rows with aff != 0) and the edge iteration order of the
    networkx graph built by setAffgraph (aff_patch_graph.py:31-40): node-major in insertion
    order, neighbours in insertion order, each edge once at its first-visited endpoint."""
    pairs = np.asarray(pairs)
    node_id, nodes, adj, val = {}, [], [], {}
    for i in np.flatnonzero(np.asarray(aff) != 0):
        u = tuple(int(v) for v in pairs[i, :3])
        v = tuple(int(v) for v in pairs[i, 3:6])
        for n in (u, v):
            if n not in node_id:
                node_id[n] = len(nodes)
                nodes.append(n)
                adj.append([])
        iu, iv = node_id[u], node_id[v]
        if (iu, iv) not in val:
            adj[iu].append(iv)
            if iu != iv:
                adj[iv].append(iu)
        val[(iu, iv)] = val[(iv, iu)] = float(aff[i])
    edges = []
    for n in range(len(nodes)):
        for nbr in adj[n]:
            if nbr >= n:  # nbr not visited yet (visit order == id order)
                edges.append((n, nbr, val[(n, nbr)]))
    return nodes, edges


def mws_from_pairs(pairs, aff):
    nodes, edge_iter = graph_edge_order(pairs, aff)
    return _mws_core(nodes, edge_iter)


def _mws_core(nodes, edge_iter):
    """nodes: list of node keys; edge_iter: (node id, node id, aff) in graph edge order."""
    edges = [(e0, e1, a, 1) if a > 0 else (e0, e1, -a, -1) for (e0, e1, a) in edge_iter]
    edges = sorted(edges, key=lambda e: e[2], reverse=True)  # stable, like the reference

    node_cc = [0] * len(nodes)
    members = {}          # cc id -> list of node ids; dict order = creation order, and a
    #                       re-issued id keeps its original position (graph_mws.py:37-38,79-82)
    held = {}             # cc id -> number of nodes currently holding it (for max())
    mutex_pairs = set()   # ordered (e0, e1) tuples, tested verbatim like the reference
    node_mutex = [set() for _ in nodes]   # node -> nodes it shares a mutex edge with
    cc_mutex = {}         # cc id -> set of nodes that have a mutex edge into the cc

    def current_max():
        return max((c for c, k in held.items() if k > 0), default=0)

    for (e0, e1, a, attractive) in edges:
        if attractive == 1 and (e0, e1) not in mutex_pairs:
            c0, c1 = node_cc[e0], node_cc[e1]
            if c0 == 0 and c1 == 0:
                new = current_max() + 1
                mem = [e0] if e0 == e1 else [e0, e1]
                members[new] = mem
                held[new] = len(mem)
                node_cc[e0] = node_cc[e1] = new
                cc_mutex[new] = set(node_mutex[e0]) | set(node_mutex[e1])
            elif c0 == 0 or c1 == 0:
                cc = max(c0, c1)
                ena = e0 if c0 == 0 else e1
                if ena not in cc_mutex[cc]:
                    members[cc].append(ena)
                    held[cc] += 1
                    node_cc[ena] = cc
                    cc_mutex[cc] |= node_mutex[ena]
            elif c0 != c1:
                small, big = (c0, c1) if len(cc_mutex[c0]) <= len(cc_mutex[c1]) else (c1, c0)
                if not any(node_cc[x] == big for x in cc_mutex[small]):
                    keep, drop = min(c0, c1), max(c0, c1)
                    merged = members[c0] + members[c1]
                    for e in members[drop]:
                        node_cc[e] = keep
                    held[keep] += held[drop]
                    held[drop] = 0
                    members[keep] = merged
                    members[drop] = []
                    cc_mutex[keep] = cc_mutex[c0] | cc_mutex[c1]
                    cc_mutex[drop] = set()
        else:
            mutex_pairs.add((e0, e1))
            node_mutex[e0].add(e1)
            node_mutex[e1].add(e0)
            for (x, y) in ((e0, e1), (e1, e0)):
                if node_cc[x] != 0:
                    cc_mutex[node_cc[x]].add(y)
    return [[nodes[i] for i in members[c]] for c in members.keys()]


def mws(affgraph):
    """Reference signature: a networkx-like graph with ``nodes()`` and ``edges.data('aff')``;
    its own node and edge orders are used verbatim."""
    nodes = list(affgraph.nodes())
    node_id = {n: i for i, n in enumerate(nodes)}
    edge_iter = [(node_id[e0], node_id[e1], a) for e0, e1, a in affgraph.edges.data("aff")]
    return _mws_core(nodes, edge_iter)
