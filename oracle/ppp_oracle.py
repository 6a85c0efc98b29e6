"""CPU ORACLE for the vote_instances path -- TEST INFRASTRUCTURE, not the product.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product package (``patchperpix_amd``) never does.

It restates, for the host CPU, the algorithm of the reference path
``PatchPerPix/vote_instances`` with ``cuda=True`` semantics:

* kernels (S1 consensus, normalise, S2 rank, S5 patch graph): ``ppp_oracle.c``,
  loaded through ctypes;
* host stages, in NumPy / plain Python: ranking sort (``ranked_patches.py:21-30``),
  greedy foreground cover (``foreground_cover.py:15-180``), set-cover thinning
  (``foreground_cover.py:183-256``), patch-pair enumeration
  (``aff_patch_graph.py:43-110``), graph construction (``aff_patch_graph.py:31-40``),
  connected components / mutex watershed + painting
  (``graph_to_labeling.py:34-155``, ``graph_mws.py:7-85``) and the orchestration with
  its early-outs (``vote_instances.py:150-452``).

PINNING: every function here is checked against golden vectors produced by the
reference itself (``tests/golden/gen_golden.py`` -> ``tests/test_oracle_golden.py``).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BG_INV_TH, BG_HALF_TH, BG_LESS_THAN_TH = 0, 1, 2
VAL_COUNT, VAL_PROB_PRODUCT, VAL_NORM_PROB_PRODUCT = 0, 1, 2


class Params(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("Z", "Y", "X", "pz", "py", "px", "nsz", "nsy", "nsx")] + \
               [("th", ctypes.c_double), ("thi", ctypes.c_double)] + \
               [(n, ctypes.c_int32) for n in
                ("bg_rule", "value_rule", "use_overlap", "norm_rank",
                 "count_pos_neg", "norm_aff", "oz", "oy", "ox")]


def build(force=False):
    """gcc-compile ppp_oracle.c into oracle/_build/libppp_oracle.so."""
    out_dir = os.path.join(_HERE, "_build")
    so = os.path.join(out_dir, "libppp_oracle.so")
    src = os.path.join(_HERE, "ppp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call(["gcc", "-O3", "-fopenmp", "-std=c11", "-ffp-contract=off", "-fPIC",
                               "-shared", src, "-o", so, "-lm"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        for name in ("ppp_oracle_fill_consensus", "ppp_oracle_norm_consensus",
                     "ppp_oracle_rank_patches", "ppp_oracle_patch_graph",
                     "ppp_oracle_consensus_and_rank", "ppp_oracle_fill_consensus_planes"):
            getattr(_LIB, name).restype = None
    return _LIB


# --------------------------------------------------------------------------------------
# parameters: utilVoteInstances.py:340-449 (macro substitution + build flags)
# --------------------------------------------------------------------------------------
def neighshape_of(patchshape):
    """vote_instances.py:249-253."""
    ps = [int(p) for p in patchshape]
    return [2 * p for p in ps] if ps[0] > 1 else [ps[0], 2 * ps[1], 2 * ps[2]]


def make_params(shape_zyx, patchshape, **kw):
    th = float(kw["patch_threshold"])
    P = Params()
    P.Z, P.Y, P.X = [int(s) for s in shape_zyx]
    P.pz, P.py, P.px = [int(p) for p in patchshape]
    P.nsz, P.nsy, P.nsx = neighshape_of(patchshape)
    P.th = th
    P.thi = th if th < 0.5 else 1.0 - th
    if kw.get("vi_bg_use_inv_th", True):
        P.bg_rule = BG_LESS_THAN_TH if th < 0.5 else BG_INV_TH
    elif kw.get("vi_bg_use_half_th", False):
        P.bg_rule = BG_HALF_TH
    elif kw.get("vi_bg_use_less_than_th", False):
        P.bg_rule = BG_LESS_THAN_TH
    else:
        raise RuntimeError("how is bg defined for vote instances?")
    P.use_overlap = 1 if kw.get("overlapping_inst", False) else 0
    if kw.get("consensus_norm_prob_product", True):
        P.value_rule = VAL_NORM_PROB_PRODUCT
    elif kw.get("consensus_prob_product", True):
        P.value_rule = VAL_PROB_PRODUCT
    else:
        assert not kw.get("consensus_norm_aff", True) and \
            not kw.get("consensus_interleaved_cnt", True), \
            "no normalizing for accumulate consensus counter available"
        P.value_rule = VAL_COUNT
    P.norm_rank = 1 if kw.get("rank_norm_patch_score", True) else 0
    P.count_pos_neg = 1 if kw.get("rank_int_counter", False) else 0
    P.norm_aff = 1 if kw.get("patch_graph_norm_aff", True) else 0
    P.oz, P.oy, P.ox = [int(v) for v in kw.get("origin", (0, 0, 0))]
    return P


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _u8(a):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a).astype(np.uint8)
    return a, a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------------------
# kernel stages
# --------------------------------------------------------------------------------------
def consensus(pred, overlap_mask, patchshape, **kw):
    """consensus_array.py:71-206 -> cons f32 [NSZ,NSY,NSX,Z,Y,X] (normalised if
    consensus_norm_aff)."""
    pred, pp = _f32(pred)
    P = make_params(pred.shape[1:], patchshape, **kw)
    ov, op = _u8(overlap_mask if P.use_overlap else None)
    shape = (P.nsz, P.nsy, P.nsx) + tuple(pred.shape[1:])
    cons = np.zeros(shape, dtype=np.float32)
    L = lib()
    if kw.get("consensus_norm_aff", True):
        cnt = np.zeros(shape, dtype=np.float32)
        if kw.get("consensus_interleaved_cnt", True):
            L.ppp_oracle_fill_consensus(pp, op, cons.ctypes.data_as(ctypes.c_void_p),
                                        cnt.ctypes.data_as(ctypes.c_void_p),
                                        ctypes.byref(P))
        else:
            L.ppp_oracle_fill_consensus(pp, op, cons.ctypes.data_as(ctypes.c_void_p),
                                        None, ctypes.byref(P))
            L.ppp_oracle_fill_consensus(pp, op, None,
                                        cnt.ctypes.data_as(ctypes.c_void_p),
                                        ctypes.byref(P))
        L.ppp_oracle_norm_consensus(pp, cons.ctypes.data_as(ctypes.c_void_p),
                                    cnt.ctypes.data_as(ctypes.c_void_p), ctypes.byref(P))
    else:
        assert not kw.get("consensus_interleaved_cnt", True), \
            "consensus aff not normalized so no computation required"
        L.ppp_oracle_fill_consensus(pp, op, cons.ctypes.data_as(ctypes.c_void_p), None,
                                    ctypes.byref(P))
    return cons


def consensus_planes(pred, overlap_mask, patchshape, **kw):
    """The same array as ``consensus`` from the gather form of S1 (one entry at a time, threads
    over offset planes, ppp_oracle_fill_consensus_planes): what bench.py's cpu_baseline times on
    all host cores.  tests/test_oracle_golden.py checks it against ``consensus`` bit for bit."""
    pred, pp = _f32(pred)
    P = make_params(pred.shape[1:], patchshape, **kw)
    ov, op = _u8(overlap_mask if P.use_overlap else None)
    shape = (P.nsz, P.nsy, P.nsx) + tuple(pred.shape[1:])
    cons = np.zeros(shape, dtype=np.float32)
    if not kw.get("consensus_norm_aff", True):
        assert not kw.get("consensus_interleaved_cnt", True), \
            "consensus aff not normalized so no computation required"
    lib().ppp_oracle_fill_consensus_planes(pp, op, cons.ctypes.data_as(ctypes.c_void_p), None,
                                           1 if kw.get("consensus_norm_aff", True) else 0,
                                           ctypes.byref(P))
    return cons


def set_threads(n):
    """Number of OpenMP threads of the oracle's C loops (0 = all cores)."""
    import ctypes.util
    omp = ctypes.CDLL(ctypes.util.find_library("gomp") or "libgomp.so.1")
    omp.omp_set_num_threads(int(n) if n else os.cpu_count())


def rank(pred, cons, overlap_mask, patchshape, **kw):
    """ranked_patches.py:33-74 -> score f32 (Z,Y,X)."""
    pred, pp = _f32(pred)
    cons, cp = _f32(cons)
    P = make_params(pred.shape[1:], patchshape, **kw)
    ov, op = _u8(overlap_mask if P.use_overlap else None)
    score = np.zeros(pred.shape[1:], dtype=np.float32)
    lib().ppp_oracle_rank_patches(pp, cp, op, score.ctypes.data_as(ctypes.c_void_p),
                                  ctypes.byref(P))
    return score


def patch_graph(pred, cons, pairs, patchshape, **kw):
    """aff_patch_graph.py:113-187 -> aff f32 [N]."""
    pred, pp = _f32(pred)
    cons, cp = _f32(cons)
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32)
    P = make_params(pred.shape[1:], patchshape, **kw)
    aff = np.zeros(pairs.shape[0], dtype=np.float32)
    lib().ppp_oracle_patch_graph(pp, cp, pairs.ctypes.data_as(ctypes.c_void_p),
                                 ctypes.c_uint64(pairs.shape[0]),
                                 aff.ctypes.data_as(ctypes.c_void_p), ctypes.byref(P))
    return aff


def positive_planes(cons, patchshape):
    """Compact [NSZ,NSY,NSX,Z,Y,X] to the lexicographically positive offset planes,
    in the order of the compact device layout (linear signed offset L = 1, 2, ...)."""
    pz, py, px = [int(p) for p in patchshape]
    planes = []
    for dz in range(0, pz):
        for dy in range(-(py - 1), py):
            for dx in range(-(px - 1), px):
                if (dz, dy, dx) <= (0, 0, 0):
                    continue
                planes.append(cons[dz + pz - 1, dy + py - 1, dx + px - 1])
    return np.stack(planes, axis=0)


# --------------------------------------------------------------------------------------
# host stages
# --------------------------------------------------------------------------------------
def interior_fg_coords(foreground, rad):
    """vote_instances.py:276,286-287: raster-ordered fg coords inside the rad border."""
    coords = np.transpose(np.where(foreground))
    shp = np.array(foreground.shape)
    keep = np.all(coords >= rad, axis=1) & np.all(coords < shp - rad, axis=1)
    return coords[keep]


def rank_by_score(coords, scores):
    """ranked_patches.py:21-30: stable sort, score descending."""
    s = scores[tuple(coords.T)]
    order = sorted(range(len(coords)), key=lambda i: s[i], reverse=True)
    order = np.array(order, dtype=np.int64)
    return coords[order], s[order]


def _window(c, rad):
    return tuple(slice(int(c[i] - rad[i]), int(c[i] + rad[i] + 1)) for i in range(3))


def _cover_loop(running, radslice, coords, scores, overlap_mask, pred, patchshape, rad, selected, pix_th,
                marked, kw):
    """computeForegroundCoverLoop (foreground_cover.py:111-180), restarting at rank 0."""
    fc = kw["fc_threshold"]
    n = len(coords)
    r = 0
    while np.max(running[radslice]) > 0 and r < n:
        i = r
        r += 1
        if selected[i]:
            continue
        if isinstance(kw.get("score_threshold", False), float) and scores[i] < kw["score_threshold"]:
            break
        c = coords[i]
        if kw.get("mark_close_neighboorhood", False) and marked[tuple(c)]:
            continue
        if overlap_mask[tuple(c)] > 0:
            continue
        patch = pred[(slice(None),) + tuple(int(v) for v in c)].reshape(patchshape)
        win = _window(c, rad)
        if np.count_nonzero(running[win][patch > fc]) > pix_th:
            selected[i] = True
            if kw.get("mark_close_neighboorhood", False):
                m_rad = np.array([0, 3, 3])
                m_start, m_stop = np.asarray(c) - m_rad, np.asarray(c) + m_rad + 1
                # (plain slices: a negative start wraps around, exactly like the reference's)
                marked[tuple(slice(int(m_start[k]), int(m_stop[k])) for k in range(3))] = True
            running[win][patch > fc] = 0


def foreground_cover(ranked_coords, ranked_scores, overlap_mask, mask_to_cover, pred,
                     patchshape, scores_array=None, **kw):
    """foreground_cover.py:15-126 incl. `mark_close_neighboorhood` and
    `select_patches_overlap_neighborhood`.  Returns indices into the ranked list -- or, with
    `select_patches_overlap_neighborhood`, the (coords, scores) the reference rebuilds in raster
    order (:83-85; needs the score volume)."""
    import scipy.ndimage
    patchshape = [int(p) for p in patchshape]
    rad = np.array([p // 2 for p in patchshape])
    radslice = tuple(slice(rad[i], mask_to_cover.shape[i] - rad[i]) for i in range(3))
    running = mask_to_cover.copy()
    n = len(ranked_coords)
    selected = np.zeros(n, dtype=bool)
    marked = np.zeros(running.shape, dtype=bool)
    if kw["select_patches_for_sparse_data"]:
        pix_ths = [0]
    else:
        mid = int(np.prod(patchshape) / 2)
        pix_ths = [t for t in [500, 100, 50, 10, 0] if t < mid]
    for pix_th in pix_ths:
        _cover_loop(running, radslice, ranked_coords, ranked_scores, overlap_mask, pred, patchshape, rad,
                    selected, pix_th, marked, kw)
        if np.sum(running[radslice]) < 1:
            break
    if not kw.get("select_patches_overlap_neighborhood", False):
        return np.nonzero(selected)[0]
    chosen = np.zeros(mask_to_cover.shape, dtype=bool)
    for c in ranked_coords[selected]:
        chosen[tuple(c)] = True
    overlap = overlap_mask.copy()
    overlap_t = scipy.ndimage.binary_dilation(overlap, iterations=2)
    overlap_dil = scipy.ndimage.binary_dilation(overlap, iterations=5)
    fg_dil_mask = np.logical_and(np.logical_and(np.logical_not(overlap_t), overlap_dil), mask_to_cover)
    keep = np.array([(not chosen[tuple(c)]) and bool(fg_dil_mask[tuple(c)]) for c in ranked_coords], dtype=bool)
    sub_c, sub_s = ranked_coords[keep], np.asarray(ranked_scores)[keep]
    sel2 = np.zeros(len(sub_c), dtype=bool)
    _cover_loop(fg_dil_mask, radslice, sub_c, sub_s, overlap_mask, pred, patchshape, rad, sel2, pix_th, marked, kw)
    for c in sub_c[sel2]:
        chosen[tuple(c)] = True
    coords = np.argwhere(chosen)
    return coords, np.asarray(scores_array)[tuple(coords.T)]


def thin_cover(sel_coords, mask_to_cover, pred, patchshape, **kw):
    """foreground_cover.py:183-256 (sample == 1.0).  Returns indices into sel_coords."""
    patchshape = [int(p) for p in patchshape]
    rad = np.array([p // 2 for p in patchshape])
    radslice = tuple(slice(rad[i], mask_to_cover.shape[i] - rad[i]) for i in range(3))
    running = mask_to_cover.copy()
    fc = kw["fc_threshold"]
    n = len(sel_coords)
    shp = np.array(mask_to_cover.shape)

    def fg_set(c, mask):  # get_patch_sets.py:32-54
        if np.all(c - rad >= 0) and np.all(c + rad + 1 <= shp):
            patch = pred[(slice(None),) + tuple(int(v) for v in c)].reshape(patchshape)
            m = (patch > fc) & mask[_window(c, rad)]
            return set(map(tuple, (c - rad) + np.argwhere(m)))
        return set()

    sets = [fg_set(c, mask_to_cover) for c in sel_coords]
    selected = np.zeros(n, dtype=bool)
    while np.max(running[radslice]) > 0:
        best = int(np.argmax([len(s) for s in sets]))
        selected[best] = True
        best_fg = fg_set(sel_coords[best], running)
        running[tuple(zip(*list(best_fg)))] = 0  # empty set -> zeroes the whole mask
        sets = [s - best_fg for s in sets]
    return np.nonzero(selected)[0]


def patch_pairs(sel_coords, patchshape, include_single=True, max_ps_dist=2):
    """aff_patch_graph.py:43-110.  Returns (x-sorted coords, pairs u32[N,6]) in the
    canonical order: (i, j)-sorted pairs of the x-sorted list, then the self pairs."""
    from scipy.spatial import cKDTree
    sel_coords = np.asarray(sel_coords).reshape(-1, 3)
    order = np.argsort(sel_coords[:, 2], kind="stable")
    pts = sel_coords[order].astype(np.uint32)
    n = len(pts)
    ps = np.array([int(p) for p in patchshape])
    rows = []
    if n > 1:
        raw = cKDTree(pts, leafsize=4).query_pairs(2 * np.sum(ps), p=1)
        for (i, j) in raw:
            d = np.abs(pts[i].astype(np.float32) - pts[j].astype(np.float32))
            if not np.any(d > max_ps_dist * ps):
                rows.append((i, j))
        rows.sort()
    total = len(rows) + (n if include_single else 0)
    if total == 0:
        return pts, None
    arr = np.zeros((total, 6), dtype=np.uint32)
    for k, (i, j) in enumerate(rows):
        arr[k, :3] = pts[i]
        arr[k, 3:] = pts[j]
    if include_single:
        arr[len(rows):, :3] = pts
        arr[len(rows):, 3:] = pts
    return pts, arr


def _graph_edges(pairs, aff, keep_zero=False):
    """Edge iteration order of ``nx.Graph.edges`` for the graph built by setAffgraph
    (aff_patch_graph.py:31-40): nodes in insertion order, per node its neighbours in
    insertion order, an edge reported at the turn of its first-visited endpoint.
    keep_zero: the graph of computePatchGraph's NumPy branch (aff_patch_graph.py:264-270), which
    adds EVERY candidate edge, also one whose votes sum to 0 -- such an edge joins nothing but its
    endpoints take their place in the node order (found by the np_c2d_p25_crop golden, round 6)."""
    nodes, adj, val = [], {}, {}
    for i in range(len(aff)):
        if aff[i] == 0 and not keep_zero:
            continue
        u = tuple(int(v) for v in pairs[i, :3])
        v = tuple(int(v) for v in pairs[i, 3:6])
        for n in (u, v):
            if n not in adj:
                adj[n] = []
                nodes.append(n)
        if v not in adj[u]:
            adj[u].append(v)
        if u not in adj[v]:
            adj[v].append(u)
        val[(u, v)] = val[(v, u)] = aff[i]
    seen = set()
    edges = []
    for n in nodes:
        for nbr in adj[n]:
            if nbr not in seen:
                edges.append((n, nbr, val[(n, nbr)]))
        seen.add(n)
    return nodes, edges


def connected_components(pairs, aff, keep_zero=False):
    """graph_to_labeling.py:50-54: CCs of the aff > 0 sub-graph in networkx's
    enumeration order (order of first appearance of a member in the edge iteration)."""
    _, edges = _graph_edges(pairs, aff, keep_zero)
    order, adj = [], {}
    for (u, v, a) in edges:
        if a > 0:
            for n in (u, v):
                if n not in adj:
                    adj[n] = []
                    order.append(n)
            adj[u].append(v)
            adj[v].append(u)
    seen, ccs = set(), []
    for n in order:
        if n in seen:
            continue
        comp, stack = [], [n]
        seen.add(n)
        while stack:
            x = stack.pop()
            comp.append(x)
            for y in adj[x]:
                if y not in seen:
                    seen.add(y)
                    stack.append(y)
        ccs.append(comp)
    return ccs


def mutex_watershed(pairs, aff, keep_zero=False):
    """graph_mws.py:7-85, including its id re-issue and empty-CC quirks."""
    nodes, edge_iter = _graph_edges(pairs, aff, keep_zero)
    node_id = {n: i for i, n in enumerate(nodes)}
    node_cc = {i: 0 for i in range(len(nodes))}
    edges = []
    for (u, v, a) in edge_iter:
        if a > 0:
            edges.append((node_id[u], node_id[v], a, 1))
        else:
            edges.append((node_id[u], node_id[v], -a, -1))
    edges = sorted(edges, key=lambda e: e[2], reverse=True)
    ccs = {0: set(node_id.values())}
    mutex = set()
    for (e0, e1, a, attractive) in edges:
        if attractive == 1 and (e0, e1) not in mutex:
            if node_cc[e0] == 0 and node_cc[e1] == 0:
                new = max(node_cc.values()) + 1
                ccs[new] = {e0, e1}
                ccs[0].discard(e0)
                ccs[0].discard(e1)
                node_cc[e0] = node_cc[e1] = new
            elif node_cc[e0] == 0 or node_cc[e1] == 0:
                cc = max(node_cc[e0], node_cc[e1])
                ena = e0 if node_cc[e0] == 0 else e1
                blocked = any((node_cc[e] == cc and f == ena) or
                              (node_cc[f] == cc and e == ena) for (e, f) in mutex)
                if not blocked:
                    ccs[cc] = ccs[cc] | {e0, e1}
                    ccs[0].discard(e0)
                    ccs[0].discard(e1)
                    node_cc[e0] = node_cc[e1] = cc
            elif node_cc[e0] != node_cc[e1]:
                c0, c1 = node_cc[e0], node_cc[e1]
                blocked = any((node_cc[e] == c0 and node_cc[f] == c1) or
                              (node_cc[f] == c0 and node_cc[e] == c1) for (e, f) in mutex)
                if not blocked:
                    keep, drop = min(c0, c1), max(c0, c1)
                    ccs[keep] = ccs[c0] | ccs[c1]
                    for e in ccs[drop]:
                        node_cc[e] = keep
                    ccs[drop] = set()
        else:
            mutex.add((e0, e1))
    return [[nodes[i] for i in ccs[k]] for k in ccs.keys() if k > 0]


def paint_instances(ccs, pred, patchshape, shape, th, dtype=np.uint16):
    """graph_to_labeling.py:57-86: later components overwrite earlier ones."""
    patchshape = [int(p) for p in patchshape]
    rad = np.array([p // 2 for p in patchshape])
    inst = np.zeros(shape, dtype=dtype)
    for k, cc in enumerate(ccs):
        for c in cc:
            c = np.array(c)
            patch = pred[(slice(None),) + tuple(int(v) for v in c)].reshape(patchshape)
            inst[_window(c, rad)][patch > th] = k + 1
    return inst


def paint_per_channel(ccs, pred, patchshape, shape, th, packed, dtype=np.uint16):
    """graph_to_labeling.py:57-115 with one_instance_per_channel (packed = False: a volume per
    component) or no_overlap_per_channel (packed = True: a component of more than 2000 voxels goes
    into the first channel it does not overlap, else a new one; smaller ones into channel 0)."""
    patchshape = [int(p) for p in patchshape]
    rad = np.array([p // 2 for p in patchshape])
    channels = []
    for k, cc in enumerate(ccs):
        cur = np.zeros(shape, dtype=dtype)
        for c in cc:
            c = np.array(c)
            patch = pred[(slice(None),) + tuple(int(v) for v in c)].reshape(patchshape)
            cur[_window(c, rad)][patch > th] = k + 1
        if not packed:
            channels.append(cur)
        elif not channels:
            channels.append(cur)
        else:
            m = cur > 0
            if np.sum(m) > 2000:
                for ch in channels:
                    if np.all(ch[m] == 0):
                        ch[m] = k + 1
                        break
                else:
                    channels.append(cur)
            else:
                channels[0][m] = k + 1
    return np.stack(channels, axis=0)


def label(pairs, aff, pred, patchshape, shape, keep_zero_edges=False, **kw):
    """keep_zero_edges: the graph came from computePatchGraph's NumPy branch (_graph_edges)."""
    ccs = mutex_watershed(pairs, aff, keep_zero_edges) if kw.get("mws") else \
        connected_components(pairs, aff, keep_zero_edges)
    if kw.get("one_instance_per_channel") or kw.get("no_overlap_per_channel"):
        return paint_per_channel(ccs, pred, patchshape, shape, kw["patch_threshold"],
                                 packed=not kw.get("one_instance_per_channel"))
    return paint_instances(ccs, pred, patchshape, shape, kw["patch_threshold"])


# --------------------------------------------------------------------------------------
# orchestration (vote_instances.py:150-452, cuda=True, no padding / debug / isbi)
# --------------------------------------------------------------------------------------
def to_instance_seg(pred, foreground, mask_to_cover, numinst, patchshape, **kw):
    """Returns a dict of every intermediate; ``instances`` / ``foreground`` are the
    reference's return values."""
    pred = np.ascontiguousarray(pred, dtype=np.float32)
    patchshape = [int(p) for p in patchshape]
    if patchshape[0] == 1 and foreground.shape[0] > 1:
        # pairs across slices make computePatchGraph.cu:98-105 read plane zo = 1 of a consensus
        # array with NSZ = 1: undefined in the reference, nothing to restate
        raise ValueError("2-d patches need 2-d data (Z = 1)")
    rad = np.array([p // 2 for p in patchshape])
    radslice = tuple(slice(rad[i], foreground.shape[i] - rad[i]) for i in range(3))
    out = {}
    overlap_mask = 1 * (numinst > 1)
    mask_to_cover = mask_to_cover.copy()
    mask_to_cover[overlap_mask > 0] = 0
    instances = np.zeros(foreground.shape, dtype=np.uint16)
    out["instances"], out["foreground"] = instances, foreground.astype(np.uint8)
    if np.count_nonzero(mask_to_cover[radslice]) == 0:
        return out
    coords = interior_fg_coords(foreground, rad)
    if len(coords) == 0:
        return out
    cons = consensus(pred, overlap_mask, patchshape, **kw)
    out["cons"] = cons
    if kw.get("selected_patches") is not None:
        # injected by the blockwise driver (vote_instances.py:369-375; it also sets skipRanking)
        sel_coords = np.array(list(kw["selected_patches"]), dtype=np.int32).reshape(-1, 3)
    else:
        scores = rank(pred, cons, overlap_mask, patchshape, **kw)
        out["scores"] = scores
        ranked_coords, ranked_scores = rank_by_score(coords, scores)
        out["ranked_coords"], out["ranked_scores"] = ranked_coords, ranked_scores
        sel = foreground_cover(ranked_coords, ranked_scores, overlap_mask, mask_to_cover,
                               pred, patchshape, scores_array=scores, **kw)
        sel_coords = sel[0] if isinstance(sel, tuple) else ranked_coords[sel]
        out["cover_coords"] = sel_coords
    if not kw["skipThinCover"] and len(sel_coords) > 0:
        keep = thin_cover(sel_coords, mask_to_cover, pred, patchshape, **kw)
        sel_coords = sel_coords[keep]
        out["thin_coords"] = sel_coords
    if kw.get("selected_patch_pairs") is not None:      # vote_instances.py:400-406
        pts = sel_coords
        pairs = np.array(kw["selected_patch_pairs"], dtype=np.uint32).reshape(-1, 6)
        pairs = pairs if len(pairs) else None
    else:
        pts, pairs = patch_pairs(sel_coords, patchshape,
                                 include_single=kw["includeSinglePatchCCS"],
                                 max_ps_dist=kw.get("max_total_patch_distance_in_ps_multiples", 2))
    out["selected_sorted"] = pts
    if pairs is None:
        return out
    out["pairs"] = pairs
    aff = patch_graph(pred, cons, pairs, patchshape, **kw)
    out["aff"] = aff
    if kw.get("return_intermediates", False):
        # vote_instances.py:435-440: (pairs, aff) go back before anything is labelled or painted -- the
        # blockwise driver injects patches of a neighbouring block whose windows may leave this one
        return out
    out["instances"] = label(pairs, aff, pred, patchshape, foreground.shape, **kw)
    return out
