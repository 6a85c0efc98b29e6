"""CPU ORACLE of the reference's NumPy path (``cuda=False``) -- TEST INFRASTRUCTURE, not the product.

Restates, with plain NumPy loops over the centres, the three stages that differ from the kernel
path: the int16 vote array (``consensus_array.py:18-68`` with the key lookup
of ``utilVoteInstances.py:19-56`` and the sets of ``get_patch_sets.py:32-79``), the integer ranking
(``ranked_patches.py:76-105``) and the all-pixel-pairs graph weights
(``aff_patch_graph.py:209-282``); cover, thinning and labelling are the host stages of
``ppp_oracle.py``.  ``removeIntersection=False, sample=1.0`` (the other settings draw from Python's
unseeded ``random`` in the reference).

PINNING: checked against tests/golden/np_*.npz -- outputs of the reference's own functions, made by
tests/golden/gen_golden_numpy_path.py -- in tests/test_numpy_semantics.py.
Only ``tests/`` imports this module.
"""
import numpy as np

from . import ppp_oracle as orc


def _offsets(ps):
    oz, oy, ox = np.meshgrid(np.arange(ps[0]), np.arange(ps[1]), np.arange(ps[2]), indexing="ij")
    rad = [p // 2 for p in ps]
    return np.stack([oz.ravel() - rad[0], oy.ravel() - rad[1], ox.ravel() - rad[2]], axis=1)


def centres(foreground, ps):
    """vote_instances.py:276,286-287"""
    return orc.interior_fg_coords(foreground, np.array([p // 2 for p in ps]))


def patch_sets(pred, mask, c, ps, th):
    """get_foreground_set / get_background_set (get_patch_sets.py:32-79) of centre c against `mask`:
    window coordinates [C, 3] and the membership vectors; float32 compares like NumPy's
    ``patchprob > pthresh`` on a float32 array."""
    off = _offsets(ps)
    v = np.asarray(c)[None, :] + off
    shp = np.array(mask.shape)
    if np.any(v.min(axis=0) < 0) or np.any(v.max(axis=0) >= shp):
        z = np.zeros(len(off), bool)
        return v, z, z
    vals = pred[(slice(None),) + tuple(int(x) for x in c)].astype(np.float32)
    inm = mask[tuple(v.T)].astype(bool)
    return v, (vals > np.float32(th)) & inm, (vals < np.float32(1 - th)) & inm


def key_of(p, q, ps):
    """lookup[p][q - p + patchshape - 1] (utilVoteInstances.py:19-56): (plane, base voxel) with the
    plane = linear signed index of the lexicographically positive offset, 0 for q == p."""
    d = tuple(int(b) - int(a) for a, b in zip(p, q))
    base = p
    if d < (0, 0, 0):
        d = tuple(-x for x in d)
        base = q
    wy, wx = 2 * ps[1] - 1, 2 * ps[2] - 1
    return (d[0] * wy + d[1]) * wx + d[2], tuple(int(x) for x in base)


def n_planes(ps):
    return ((2 * ps[0] - 1) * (2 * ps[1] - 1) * (2 * ps[2] - 1) - 1) // 2 + 1


def consensus(pred, foreground, ps, th):
    """create_consensus_array: int16 votes [planes, Z, Y, X]."""
    ps = [int(p) for p in ps]
    votes = np.zeros((n_planes(ps),) + tuple(foreground.shape), dtype=np.int16)
    for c in centres(foreground, ps):
        v, pf, pb = patch_sets(pred, foreground, c, ps, th)
        f = [tuple(x) for x in v[pf]]
        b = [tuple(x) for x in v[pb]]
        if len(f) > 1:
            for i, p in enumerate(f[:-1]):
                for q in f[i + 1:]:
                    k, base = key_of(p, q, ps)
                    votes[(k,) + base] += 1
        if f and b:
            keys = {key_of(p, q, ps) for p in f for q in b}       # `a[idx] -= 1`: once per distinct key
            for k, base in keys:
                votes[(k,) + base] -= 1
    return votes


def rank(pred, foreground, votes, ps, th):
    """rank_patches: int64 score per centre, in the centres' (raster) order."""
    ps = [int(p) for p in ps]
    cs = centres(foreground, ps)
    scores = np.zeros(len(cs), dtype=np.int64)
    for i, c in enumerate(cs):
        v, pf, pb = patch_sets(pred, foreground, c, ps, th)
        f = [tuple(x) for x in v[pf]]
        b = [tuple(x) for x in v[pb]]
        s = 0
        for j, p in enumerate(f[:-1] if len(f) > 1 else []):
            for q in f[j + 1:]:
                k, base = key_of(p, q, ps)
                s += 1 if votes[(k,) + base] > 0 else -1
        if f and b:
            for p in f:
                for q in b:
                    k, base = key_of(p, q, ps)
                    s += 1 if votes[(k,) + base] < 0 else -1
        scores[i] = s
    return cs, scores


def ranked(cs, scores):
    """sorted(..., key=score, reverse=True): stable"""
    order = np.argsort(-scores, kind="stable")
    return cs[order], scores[order]


def patch_graph(pred, mask_to_cover, overlap_mask, votes, selected_sorted, ps, th, include_single=True):
    """computePatchGraph, NumPy branch: edges in the order of the (r1, r2) loops as
    (rows int32 [n, 6], weight int64 [n])."""
    ps = [int(p) for p in ps]
    sel = np.asarray(selected_sorted).reshape(-1, 3)
    sets = []
    for c in sel:
        v, pf, _ = patch_sets(pred, mask_to_cover, c, ps, th)
        sets.append([tuple(x) for x in v[pf]])
    rows, weights = [], []
    for r1 in range(len(sel)):
        for r2 in range(r1 if include_single else r1 + 1, len(sel)):
            a, b = sel[r1], sel[r2]
            if overlap_mask[tuple(a)] and overlap_mask[tuple(b)]:
                continue
            if np.any(np.abs(b - a) > np.array(ps)):
                continue
            if not sets[r1] or not sets[r2]:
                continue
            w, n = 0, 0
            for p in sets[r1]:
                for q in sets[r2]:
                    d = [abs(x - y) for x, y in zip(p, q)]
                    if all(x < s for x, s in zip(d, ps)) and any(d):
                        k, base = key_of(p, q, ps)
                        w += int(votes[(k,) + base])
                        n += 1
            if n:
                rows.append(list(a) + list(b))
                weights.append(w)
    return np.array(rows, dtype=np.int32).reshape(-1, 6), np.array(weights, dtype=np.int64)
