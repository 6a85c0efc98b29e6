"""CPU ORACLE at benchmark scale -- TEST INFRASTRUCTURE, not the product.

``ppp_oracle.py`` restates the reference's host stages literally (Python sets, a scan of the whole
mutex set per edge, ...): exact, but quadratic -- at 96^3 / 9^3 the thinning alone would take
hours.  This module holds forms of the SAME functions that finish in minutes at that size; each is
checked against its literal counterpart on every golden and on random cases
(``tests/test_oracle_scale.py``), so a result it produces at 96^3 is the oracle's result.

Only ``tests/`` (and the fixture generator ``tests/golden/gen_scale_fixture.py``) import it.
"""
import numpy as np

from . import ppp_oracle as orc


def thin_cover(sel_coords, mask_to_cover, pred, patchshape, **kw):
    """foreground_cover.py:183-256 (``ppp_oracle.thin_cover``) with the set sizes kept as
    counters: ``len(sets[i])`` = voxels of patch i's foreground that are still in the running
    mask; clearing a voxel lowers the counter of every patch whose window holds it and whose
    prediction there is above fc_threshold.  Same picks in the same order, incl. the degenerate
    end (every counter 0 with voxels left: argmax returns patch 0, the empty index tuple zeroes
    the whole mask).  Returns indices into sel_coords."""
    ps = [int(p) for p in patchshape]
    rad = np.array([p // 2 for p in ps])
    shp = np.array(mask_to_cover.shape)
    radslice = tuple(slice(rad[i], shp[i] - rad[i]) for i in range(3))
    running = mask_to_cover.astype(bool).copy()
    fc = kw["fc_threshold"]
    sel_coords = np.asarray(sel_coords).reshape(-1, 3).astype(np.int64)
    n = len(sel_coords)
    C = int(np.prod(ps))
    inside = np.all(sel_coords - rad >= 0, axis=1) & np.all(sel_coords + rad + 1 <= shp, axis=1)
    # patch index at its centre voxel (-1: no candidate there); a centre that occurs twice keeps
    # every occurrence (a list per voxel is not needed: the cover never selects a voxel twice)
    lin = np.ravel_multi_index(sel_coords.T, shp) if n else np.zeros(0, np.int64)
    assert len(np.unique(lin)) == n, "duplicate centres: use ppp_oracle.thin_cover"
    index_at = np.full(int(np.prod(shp)), -1, dtype=np.int64)
    index_at[lin] = np.arange(n)
    # window offsets in raster order of the patch channel r
    oz, oy, ox = np.meshgrid(np.arange(ps[0]), np.arange(ps[1]), np.arange(ps[2]), indexing="ij")
    off = np.stack([oz.ravel() - rad[0], oy.ravel() - rad[1], ox.ravel() - rad[2]], axis=1)   # [C, 3]
    flat_pred = pred.reshape(C, -1)

    def fg_voxels(c, mask):
        """get_patch_sets.py:32-54: linear indices of the patch's foreground voxels in `mask`"""
        v = c[None, :] + off
        lv = np.ravel_multi_index(v.T, shp)
        m = (flat_pred[np.arange(C), np.ravel_multi_index(c, shp)] > fc) & mask.reshape(-1)[lv]
        return lv[m]

    counts = np.zeros(n, dtype=np.int64)
    for i in range(n):
        if inside[i]:
            counts[i] = len(fg_voxels(sel_coords[i], running))
    selected = np.zeros(n, dtype=bool)
    rflat = running.reshape(-1)
    remaining = int(np.count_nonzero(running[radslice]))
    inner = np.zeros(shp, dtype=bool)
    inner[radslice] = True
    inner = inner.reshape(-1)
    while remaining > 0:
        best = int(np.argmax(counts))
        selected[best] = True
        cleared = fg_voxels(sel_coords[best], running) if inside[best] else np.zeros(0, np.int64)
        if len(cleared) == 0:
            break                      # mask[()] = 0 zeroes everything: the loop ends
        rflat[cleared] = False
        remaining -= int(np.count_nonzero(inner[cleared]))
        # every (cleared voxel v, channel r): the patch centred at v - off[r] loses v if its
        # prediction in channel r is above fc
        vz, vy, vx = np.unravel_index(cleared, shp)
        cz = vz[:, None] - off[None, :, 0]
        cy = vy[:, None] - off[None, :, 1]
        cx = vx[:, None] - off[None, :, 2]
        ok = (cz >= 0) & (cz < shp[0]) & (cy >= 0) & (cy < shp[1]) & (cx >= 0) & (cx < shp[2])
        lc = (np.where(ok, cz, 0) * shp[1] + np.where(ok, cy, 0)) * shp[2] + np.where(ok, cx, 0)
        j = np.where(ok, index_at[lc], -1)
        r = np.broadcast_to(np.arange(C)[None, :], j.shape)
        hit = j >= 0
        jj, rr, ll = j[hit], r[hit], lc[hit]
        keep = inside[jj] & (flat_pred[rr, ll] > fc)
        np.subtract.at(counts, jj[keep], 1)
    return np.nonzero(selected)[0]


def patch_pairs(sel_coords, patchshape, include_single=True, max_ps_dist=2):
    """aff_patch_graph.py:43-110 (``ppp_oracle.patch_pairs``), the candidate filter and the
    canonical order evaluated on arrays."""
    from scipy.spatial import cKDTree
    sel_coords = np.asarray(sel_coords).reshape(-1, 3)
    order = np.argsort(sel_coords[:, 2], kind="stable")
    pts = sel_coords[order].astype(np.uint32)
    n = len(pts)
    ps = np.array([int(p) for p in patchshape])
    rows = np.zeros((0, 2), dtype=np.int64)
    if n > 1:
        raw = cKDTree(pts, leafsize=4).query_pairs(2 * np.sum(ps), p=1, output_type="ndarray")
        if len(raw):
            raw = np.sort(raw.astype(np.int64), axis=1)           # i < j
            d = np.abs(pts[raw[:, 0]].astype(np.float32) - pts[raw[:, 1]].astype(np.float32))
            raw = raw[~np.any(d > max_ps_dist * ps, axis=1)]
            rows = raw[np.lexsort((raw[:, 1], raw[:, 0]))]
    total = len(rows) + (n if include_single else 0)
    if total == 0:
        return pts, None
    arr = np.zeros((total, 6), dtype=np.uint32)
    arr[:len(rows), :3] = pts[rows[:, 0]]
    arr[:len(rows), 3:] = pts[rows[:, 1]]
    if include_single:
        arr[len(rows):, :3] = pts
        arr[len(rows):, 3:] = pts
    return pts, arr


def graph_edges(pairs, aff, shape):
    """``ppp_oracle._graph_edges`` on arrays: node ids in insertion order and the edges in
    ``nx.Graph.edges`` order -- an edge (u, v) is reported at the turn of whichever endpoint was
    inserted first, among that node's edges in the order its neighbours were first linked; a row
    that repeats an edge overwrites its weight (last one wins) and keeps the first position; a
    self loop is one edge."""
    pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 6)
    aff = np.asarray(aff, dtype=np.float32)
    rows = np.nonzero(aff != 0)[0]
    if len(rows) == 0:
        return np.zeros(0, np.int64), np.zeros((0, 2), np.int64), np.zeros(0, np.float32)
    a = np.ravel_multi_index(pairs[rows, :3].T, shape)
    b = np.ravel_multi_index(pairs[rows, 3:].T, shape)
    # insertion order of the nodes: first appearance in the sequence a0, b0, a1, b1, ...
    seq = np.stack([a, b], axis=1).ravel()
    uniq, first = np.unique(seq, return_index=True)
    order = np.argsort(first, kind="stable")
    nodes_lin = uniq[order]
    node_of = np.empty(len(uniq), np.int64)
    node_of[order] = np.arange(len(uniq))
    ia = node_of[np.searchsorted(uniq, a)]
    ib = node_of[np.searchsorted(uniq, b)]
    lo, hi = np.minimum(ia, ib), np.maximum(ia, ib)        # lo was inserted first: its turn reports the edge
    key = lo * len(uniq) + hi
    # unique undirected edges: position = first row, weight = last row
    uk, first_row = np.unique(key, return_index=True)
    last_row = len(key) - 1 - np.unique(key[::-1], return_index=True)[1]
    e_lo, e_hi = uk // len(uniq), uk % len(uniq)
    w = aff[rows][last_row]
    # within node lo's adjacency the neighbours are in the order they were first linked = first row
    ordr = np.lexsort((first_row, e_lo))
    # endpoints as the iteration reports them: (n, nbr) with n the node whose turn it is
    return nodes_lin, np.stack([e_lo[ordr], e_hi[ordr]], axis=1), w[ordr]


def mutex_watershed(pairs, aff, shape):
    """graph_mws.py:7-85 (``ppp_oracle.mutex_watershed``) with per-node mutex partner sets and
    member lists per id instead of scans of the whole mutex set.  Returns (nodes_lin, label per
    node [0 = never assigned], number of ids ever issued): the label of a node is the position of
    its component in the reference's output list + 1 -- ids are first issued in increasing order,
    merged-away ids keep their (empty) slot."""
    import heapq
    nodes_lin, e, w = graph_edges(pairs, aff, shape)
    n = len(nodes_lin)
    if n == 0:
        return nodes_lin, np.zeros(0, np.int64), 0
    order = np.argsort(-np.abs(w), kind="stable")          # sorted(..., key=|a|, reverse=True) is stable
    cc = np.zeros(n, dtype=np.int64)
    members = {}
    partners = [None] * n                                  # mutex partners of a node
    mutex_tuples = set()
    held = []                                              # max-heap (negated) of ids that hold nodes
    n_issued = 0
    e0s, e1s, att = e[order, 0].tolist(), e[order, 1].tolist(), (w[order] > 0).tolist()
    for e0, e1, attractive in zip(e0s, e1s, att):
        if attractive and (e0, e1) not in mutex_tuples:
            c0, c1 = cc[e0], cc[e1]
            if c0 == 0 and c1 == 0:
                while held and not members.get(-held[0]):
                    heapq.heappop(held)
                new = (-held[0] if held else 0) + 1
                members[new] = [e0, e1] if e0 != e1 else [e0]
                cc[e0] = cc[e1] = new
                heapq.heappush(held, -new)
                n_issued = max(n_issued, new)
            elif c0 == 0 or c1 == 0:
                c = max(c0, c1)
                ena = e0 if c0 == 0 else e1
                blocked = partners[ena] is not None and any(cc[p] == c for p in partners[ena])
                if not blocked:
                    members[c].append(ena)
                    cc[ena] = c
            elif c0 != c1:
                small, big = (c0, c1) if len(members[c0]) <= len(members[c1]) else (c1, c0)
                blocked = False
                for m in members[small]:
                    ps_ = partners[m]
                    if ps_ is not None and any(cc[p] == big for p in ps_):
                        blocked = True
                        break
                if not blocked:
                    keep, drop = min(c0, c1), max(c0, c1)
                    for m in members[drop]:
                        cc[m] = keep
                    members[keep] = members[keep] + members[drop]
                    members[drop] = []
        else:
            mutex_tuples.add((e0, e1))
            for x, y in ((e0, e1), (e1, e0)):
                if partners[x] is None:
                    partners[x] = set()
                partners[x].add(y)
    return nodes_lin, cc, n_issued


def paint(nodes_lin, labels, pred, patchshape, shape, th, dtype=np.uint32):
    """graph_to_labeling.py:57-86: components painted in list order (label ascending), later ones
    overwrite earlier ones; within a component every patch writes the same label."""
    ps = [int(p) for p in patchshape]
    rad = np.array([p // 2 for p in ps])
    inst = np.zeros(shape, dtype=dtype)
    order = np.argsort(labels, kind="stable")
    coords = np.stack(np.unravel_index(nodes_lin, shape), axis=1)
    for i in order:
        if labels[i] == 0:
            continue
        c = coords[i]
        patch = pred[(slice(None),) + tuple(int(v) for v in c)].reshape(ps)
        win = tuple(slice(int(c[k] - rad[k]), int(c[k] + rad[k] + 1)) for k in range(3))
        inst[win][patch > th] = labels[i]
    return inst


def components(pairs, aff, shape):
    """graph_to_labeling.py:50-54 (``ppp_oracle.connected_components``): labels of the aff > 0
    sub-graph's components in networkx's enumeration order."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components as cc_
    nodes_lin, e, w = graph_edges(pairs, aff, shape)
    n = len(nodes_lin)
    labels = np.zeros(n, dtype=np.int64)
    pos = w > 0
    if not np.any(pos):
        return nodes_lin, labels, 0
    ep = e[pos]
    _, comp = cc_(coo_matrix((np.ones(len(ep)), (ep[:, 0], ep[:, 1])), shape=(n, n)), directed=False)
    # enumeration order: first appearance of a member in the positive-edge iteration
    seq = ep.ravel()
    in_graph = np.zeros(n, dtype=bool)
    in_graph[seq] = True
    first = np.full(comp.max() + 1, np.iinfo(np.int64).max)
    np.minimum.at(first, comp[seq], np.arange(len(seq)))
    rank = np.argsort(np.argsort(first))
    labels[in_graph] = rank[comp[in_graph]] + 1
    return nodes_lin, labels, int(labels.max())


def to_instance_seg(pred, foreground, mask_to_cover, numinst, patchshape, dtype=np.uint32, **kw):
    """``ppp_oracle.to_instance_seg`` (vote_instances.py:150-452) with the forms above; the three
    kernel stages are the oracle's C loops (S1 in its gather form over offset planes, which
    tests/test_oracle_golden.py holds bit-identical to the serial scatter form)."""
    pred = np.ascontiguousarray(pred, dtype=np.float32)
    ps = [int(p) for p in patchshape]
    shape = tuple(foreground.shape)
    rad = np.array([p // 2 for p in ps])
    radslice = tuple(slice(rad[i], shape[i] - rad[i]) for i in range(3))
    out = {}
    overlap_mask = 1 * (numinst > 1)
    mask_to_cover = mask_to_cover.copy()
    mask_to_cover[overlap_mask > 0] = 0
    out["instances"] = np.zeros(shape, dtype=dtype)
    if np.count_nonzero(mask_to_cover[radslice]) == 0:
        return out
    coords = orc.interior_fg_coords(foreground, rad)
    if len(coords) == 0:
        return out
    cons = orc.consensus_planes(pred, overlap_mask, ps, **kw)
    scores = orc.rank(pred, cons, overlap_mask, ps, **kw)
    out["scores"] = scores
    s = scores[tuple(coords.T)]
    order = np.argsort(-s.astype(np.float64), kind="stable")      # stable, score descending
    ranked_coords, ranked_scores = coords[order], s[order]
    sel = orc.foreground_cover(ranked_coords, ranked_scores, overlap_mask, mask_to_cover, pred, ps,
                               scores_array=scores, **kw)
    sel_coords = sel[0] if isinstance(sel, tuple) else ranked_coords[sel]
    out["cover_coords"] = sel_coords
    if not kw["skipThinCover"] and len(sel_coords) > 0:
        sel_coords = sel_coords[thin_cover(sel_coords, mask_to_cover, pred, ps, **kw)]
        out["thin_coords"] = sel_coords
    pts, pairs = patch_pairs(sel_coords, ps, include_single=kw["includeSinglePatchCCS"],
                             max_ps_dist=kw.get("max_total_patch_distance_in_ps_multiples", 2))
    if pairs is None:
        return out
    out["pairs"] = pairs
    aff = orc.patch_graph(pred, cons, pairs, ps, **kw)
    del cons
    out["aff"] = aff
    if kw.get("mws"):
        nodes_lin, labels, n_ids = mutex_watershed(pairs, aff, shape)
    else:
        nodes_lin, labels, n_ids = components(pairs, aff, shape)
    out["n_ids"] = n_ids
    out["instances"] = paint(nodes_lin, labels, pred, ps, shape, kw["patch_threshold"], dtype=dtype)
    return out
