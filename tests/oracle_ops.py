"""Oracle-backed stand-in for patchperpix_amd.tiling.DeviceOps (CPU, torch CPU tensors).

Lets the slab decomposition / collectives of patchperpix_amd.tiling be exercised without a GPU:
the voxel-local stages are evaluated by the CPU oracle on exactly the local sub-volumes, boxes
and coordinate offsets a rank would hand to the HIP kernels.  TEST INFRASTRUCTURE ONLY."""
import numpy as np
import torch

from oracle import ppp_oracle as orc
from patchperpix_amd import backend


class OracleOps:
    device = torch.device("cpu")

    def __init__(self, **flags):
        self.kw = flags

    # -- helpers
    def _ps(self, P):
        return [P.pz, P.py, P.px]

    def _ref_layout(self, cons_box_t, P):
        ps = self._ps(P)
        ns = orc.neighshape_of(ps)
        full = np.zeros(tuple(ns) + (P.Z, P.Y, P.X), dtype=np.float32)
        b = P.cons_box
        c = cons_box_t.numpy()
        k = 0
        for dz in range(0, ps[0]):
            for dy in range(-(ps[1] - 1), ps[1]):
                for dx in range(-(ps[2] - 1), ps[2]):
                    if (dz, dy, dx) <= (0, 0, 0):
                        continue
                    full[dz + ps[0] - 1, dy + ps[1] - 1, dx + ps[2] - 1,
                         b.z0:b.z1, b.y0:b.y1, b.x0:b.x1] = c[k]
                    k += 1
        return full

    # -- the DeviceOps API
    def consensus(self, pred, ov, P):
        ps = self._ps(P)
        full = orc.consensus(pred.numpy().astype(np.float32), ov.numpy(), ps, **self.kw)
        b = P.cons_box
        pos = orc.positive_planes(full, ps)[:, b.z0:b.z1, b.y0:b.y1, b.x0:b.x1]
        return torch.from_numpy(np.ascontiguousarray(pos))

    # -- consensus cache (same contract as DeviceOps: COMPACT planes over the cache's box, filled
    #    part by part; a tile's share handed out in the layout rank_patches / patch_graph read)
    def cons_cache_alloc(self, P):
        n_planes = ((2 * P.pz - 1) * (2 * P.py - 1) * (2 * P.px - 1) - 1) // 2
        return torch.full((n_planes,) + P.cons_box.shape(), float("nan"), dtype=torch.float32)

    def cons_cache_fill(self, pred, ov, P, part, cache):
        z0, y0, x0, z1, y1, x1 = [int(v) for v in part]
        b = P.cons_box
        assert b.z0 <= z0 < z1 <= b.z1 and b.y0 <= y0 < y1 <= b.y1 and b.x0 <= x0 < x1 <= b.x1
        key = (pred.data_ptr(), tuple(pred.shape))
        if getattr(self, "_cache_full_key", None) != key:
            full = orc.consensus(pred.numpy().astype(np.float32), ov.numpy(), self._ps(P), **self.kw)
            self._cache_full = orc.positive_planes(full, self._ps(P))
            self._cache_full_key = key
        piece = self._cache_full[:, z0:z1, y0:y1, x0:x1]
        view = cache[:, z0 - b.z0:z1 - b.z0, y0 - b.y0:y1 - b.y0, x0 - b.x0:x1 - b.x0]
        assert bool(torch.isnan(view).all()), "a base voxel of the cache is filled twice"
        view.copy_(torch.from_numpy(np.ascontiguousarray(piece)))

    def cons_from_cache(self, cache, cache_box, P, out=None):
        cz0, cy0, cx0 = [int(v) for v in cache_box[:3]]
        b = P.cons_box
        piece = cache[:, b.z0 - cz0:b.z1 - cz0, b.y0 - cy0:b.y1 - cy0, b.x0 - cx0:b.x1 - cx0]
        assert not bool(torch.isnan(piece).any()), "a base voxel of the cache was never filled"
        return piece.contiguous(), P

    def rank_patches(self, pred, cons, ov, P, score_box):
        full = self._ref_layout(cons, P)
        s = orc.rank(pred.numpy().astype(np.float32), full, ov.numpy(), self._ps(P), **self.kw)
        return torch.from_numpy(s)

    def patch_bits(self, pred, centres, thresh, P, scratch=None):
        p = pred.numpy().astype(np.float32)
        c = centres.numpy()
        C = p.shape[0]
        words = (C + 31) // 32
        vals = p[(slice(None),) + tuple(c.T)].T > np.float32(thresh)
        out = np.zeros((len(c), words), dtype=np.uint32)
        for r in range(C):
            out[:, r // 32] |= vals[:, r].astype(np.uint32) << np.uint32(r % 32)
        return torch.from_numpy(out.view(np.int32))

    def patch_pairs(self, sorted_zyx, P, max_ps_dist, include_single):
        _, rows = orc.patch_pairs(sorted_zyx.numpy(), self._ps(P), include_single=include_single,
                                  max_ps_dist=max_ps_dist)
        return None if rows is None else torch.from_numpy(rows.view(np.int32).copy())

    # -- streaming pairs / labels (same contracts as backend.pair_counts_subset, pairs_subset,
    #    LabelState; plain Python over the oracle's canonical pair list)
    def _all_rows(self, sorted_zyx, P, max_ps_dist):
        pts = sorted_zyx.numpy()
        memo = getattr(self, "_rows_memo", None)
        if memo is not None and memo[0] == (pts.tobytes(), max_ps_dist):
            return memo[1]
        out = self._all_rows_uncached(pts, P, max_ps_dist)
        self._rows_memo = ((pts.tobytes(), max_ps_dist), out)
        return out

    def _all_rows_uncached(self, pts, P, max_ps_dist):
        _, rows = orc.patch_pairs(pts, self._ps(P), include_single=False, max_ps_dist=max_ps_dist)
        rows = np.zeros((0, 6), np.uint32) if rows is None else rows
        index = {tuple(int(v) for v in c): i for i, c in enumerate(pts)}
        owner = np.array([index[tuple(int(v) for v in r[:3])] for r in rows], dtype=np.int64)
        return pts, rows, owner

    def pair_counts(self, sorted_zyx, subset, P, max_ps_dist):
        pts, rows, owner = self._all_rows(sorted_zyx, P, max_ps_dist)
        counts = np.bincount(owner, minlength=len(pts)).astype(np.int64)
        keep = np.zeros(len(pts), dtype=bool)
        keep[subset.numpy()] = True
        counts[~keep] = 0
        return torch.from_numpy(counts)

    def pairs_subset(self, sorted_zyx, subset, counts, goffsets, n_rows_total, P, max_ps_dist,
                     include_single):
        pts, rows, owner = self._all_rows(sorted_zyx, P, max_ps_dist)
        sub = subset.numpy()
        if len(sub) == 0:
            return None, None
        assert np.all(np.diff(sub) > 0)
        keep = np.zeros(len(pts), dtype=bool)
        keep[sub] = True
        sel = np.flatnonzero(keep[owner])          # canonical order is kept
        # the global ids must agree with the scan of the all-reduced counts
        g = goffsets.numpy()
        c = counts.numpy()
        gid = np.concatenate([g[i] + np.arange(c[i]) for i in sub]) if len(sub) else np.zeros(0, np.int64)
        assert np.array_equal(gid, sel)
        out_rows, out_gid = rows[sel], gid.astype(np.int64)
        if include_single:
            selfs = np.concatenate([pts[sub], pts[sub]], axis=1).astype(np.uint32)
            out_rows = np.concatenate([out_rows, selfs], axis=0)
            out_gid = np.concatenate([out_gid, int(n_rows_total) + sub.astype(np.int64)])
        if len(out_rows) == 0:
            return None, None
        return torch.from_numpy(np.ascontiguousarray(out_rows).view(np.int32).copy()), \
            torch.from_numpy(out_gid)

    def label_state(self, nodes, P):
        return OracleLabelState(nodes.numpy(), (P.Z, P.Y, P.X))

    def thin_shard(self, mask_local, lin_local, index_global, bits, P_local, global_z):
        return OracleThinShard(mask_local, lin_local, index_global, bits, P_local, global_z)

    def cover_shard(self, mask_local, lin_local, rank_id, bits, P_local, global_z):
        return OracleCoverShard(mask_local, lin_local, rank_id, bits, P_local, global_z)

    def patch_graph(self, pred, cons, rows, P):
        full = self._ref_layout(cons, P)
        kw = dict(self.kw, origin=(P.origin_z, P.origin_y, P.origin_x))
        aff = orc.patch_graph(pred.numpy().astype(np.float32), full,
                              rows.numpy().view(np.uint32), self._ps(P), **kw)
        return torch.from_numpy(aff)

    def rank_order(self, score_dev, foreground, ps):
        score_host = score_dev.numpy()
        fg = foreground.numpy() if torch.is_tensor(foreground) else foreground
        lin = backend.host_rank_order(score_host, fg, ps)
        return torch.from_numpy(lin), torch.from_numpy(np.ascontiguousarray(score_host.reshape(-1)[lin]))

    def label_components(self, rows, aff, nodes, P):
        ccs = orc.connected_components(rows.numpy().view(np.uint32), aff.numpy())
        key = {}
        for k, cc in enumerate(ccs):
            for n in cc:
                key[n] = k
        out = np.array([key.get(tuple(int(v) for v in n), backend.NONE_KEY)
                        for n in nodes.numpy()], dtype=np.int64)
        return torch.from_numpy(out)

    def paint(self, pred, nodes, labels, inst, P):
        p = pred.numpy().astype(np.float32)
        ps = self._ps(P)
        rad = [q // 2 for q in ps]
        th = np.float32(self.kw["patch_threshold"])
        a = inst.numpy()
        for c, lab in zip(nodes.numpy(), labels.numpy()):
            patch = p[:, c[0], c[1], c[2]].reshape(ps) > th
            for r in np.argwhere(patch):
                z, y, x = c[0] + r[0] - rad[0], c[1] + r[1] - rad[1], c[2] + r[2] - rad[2]
                if 0 <= z < a.shape[0] and 0 <= y < a.shape[1] and 0 <= x < a.shape[2]:
                    a[z, y, x] = max(a[z, y, x], lab)
        return inst

    def greedy_cover(self, mask_to_cover, bits, lin, scores, never, pix_ths, radslice, P, kw):
        """The sequential loop (native host code of the library, pinned to the reference's
        goldens by tests/test_abi_and_host.py)."""
        lin, scores, never = lin.numpy(), scores.numpy(), never.numpy()
        if torch.is_tensor(mask_to_cover):
            mask_to_cover = mask_to_cover.numpy()
        running, _owner = backend.padded_mask(mask_to_cover)
        selected = np.zeros(len(lin), dtype=np.uint8)
        b = bits.numpy().view(np.uint32)
        # `never` as the loop sees it: overlap centres are skipped, the score threshold breaks
        ov = np.zeros(running.size, dtype=np.uint8)
        thr = kw.get("score_threshold", False)
        thr = thr if isinstance(thr, float) else None
        cut = len(lin)
        if thr is not None:
            below = np.flatnonzero(np.asarray(scores, dtype=np.float64) < thr)
            cut = int(below[0]) if len(below) else cut
        ov[lin[never[:cut].nonzero()[0]]] = 1
        remaining = int(np.count_nonzero(running[radslice]))
        for pix_th in pix_ths:
            if remaining <= 0:
                break
            remaining, stopped = backend.host_cover_pass(
                running, ov.reshape(running.shape), [P.pz, P.py, P.px], lin, scores, b, pix_th,
                thr, selected, remaining)
            if remaining < 1:
                break
        return torch.from_numpy(selected.astype(bool))


class OracleLabelState:
    """backend.LabelState in NumPy: union-find over node indices, first appearances among rows
    with aff != 0, "has a positive edge"; keys as in oracle.connected_components."""
    NONE = backend.NONE_KEY64

    def __init__(self, nodes, shape):
        self.nodes = nodes
        Z, Y, X = shape
        self.lin = (nodes[:, 0].astype(np.int64) * Y + nodes[:, 1]) * X + nodes[:, 2]
        self.of_lin = {int(l): i for i, l in enumerate(self.lin)}
        n = len(nodes)
        self.parent = np.arange(n)
        self.firstpos = np.full(n, self.NONE, dtype=np.int64)
        self.haspos = np.zeros(n, dtype=np.int32)
        self.YX = (Y, X)

    def _find(self, x):
        while self.parent[x] != x:
            self.parent[x] = self.parent[self.parent[x]]
            x = self.parent[x]
        return x

    def _union(self, a, b):
        a, b = self._find(a), self._find(b)
        if a != b:
            self.parent[max(a, b)] = min(a, b)

    def add(self, rows, aff, gid=None, first_id=0):
        r = rows.numpy().view(np.uint32).astype(np.int64)
        a = aff.numpy()
        g = gid.numpy() if gid is not None else first_id + np.arange(len(r))
        Y, X = self.YX
        for i in range(len(r)):
            if a[i] == 0:
                continue
            u = self.of_lin[int((r[i, 0] * Y + r[i, 1]) * X + r[i, 2])]
            v = self.of_lin[int((r[i, 3] * Y + r[i, 4]) * X + r[i, 5])]
            self.firstpos[u] = min(self.firstpos[u], 2 * int(g[i]))
            self.firstpos[v] = min(self.firstpos[v], 2 * int(g[i]) + 1)
            if a[i] > 0:
                self.haspos[u] = self.haspos[v] = 1
                self._union(u, v)

    def export(self):
        par = np.array([self.lin[self._find(i)] for i in range(len(self.nodes))], dtype=np.int64)
        return torch.from_numpy(par), torch.from_numpy(self.firstpos.copy()), \
            torch.from_numpy(self.haspos.copy())

    def merge(self, parents, firstpos, haspos):
        for row in parents.numpy().reshape(-1, len(self.nodes)):
            for i, l in enumerate(row):
                self._union(i, self.of_lin[int(l)])
        self.firstpos = firstpos.numpy().copy()
        self.haspos = haspos.numpy().astype(np.int32).copy()

    def finish(self):
        n = len(self.nodes)
        key = np.full(n, self.NONE, dtype=np.int64)
        for i in range(n):
            if self.haspos[i]:
                r = self._find(i)
                key[r] = min(key[r], self.firstpos[i])
        return torch.from_numpy(np.array([key[self._find(i)] for i in range(n)], dtype=np.int64))


class OracleCoverShard:
    """backend.CoverShard in NumPy: the count / filter / select steps of csrc/ppp_cover.hip on a
    rank's local buffer, and the zone export / import used to exchange halos."""
    COUNT, FILTER, SELECT = 0, 1, 2
    NONE = 0x7F7F7F7F
    IMAX = 0x7FFFFFFF

    def __init__(self, mask, lin_local, rank_id, bits, P, global_z):
        self.mask_t = mask
        self.shape = (P.Z, P.Y, P.X)
        self.ps = (P.pz, P.py, P.px)
        self.oz, self.gz = P.origin_z, int(global_z)
        self.lin = lin_local.numpy().astype(np.int64)
        self.rank_id = rank_id.numpy().astype(np.int64)
        self.n = len(self.lin)
        C = int(np.prod(self.ps))
        b = bits.numpy().view(np.uint32).reshape(self.n, -1) if self.n else np.zeros((0, (C + 31) // 32), np.uint32)
        self.pbits = np.zeros((self.n, C), dtype=bool)
        for r in range(C):
            self.pbits[:, r] = (b[:, r // 32] >> np.uint32(r % 32)) & 1
        self.pbits = self.pbits.reshape((self.n,) + self.ps)
        self.state = self.cleared = None
        self._alive = False

    def _win(self, c):
        rad = [p // 2 for p in self.ps]
        return tuple(slice(c[i] - rad[i], c[i] + rad[i] + 1) for i in range(3))

    def open(self, state):
        self.state = state.clone()
        self.cleared = torch.zeros(max(self.n, 1), dtype=torch.int32)
        self.run = self.mask_t.numpy() != 0
        self.rank_vol = np.full(self.shape, self.NONE, dtype=np.int64)
        self.loc_vol = np.full(self.shape, -1, dtype=np.int64)
        self.dirty = np.ones(self.shape, dtype=bool)
        st = self.state.numpy()
        self.centres = np.stack(np.unravel_index(self.lin, self.shape), axis=1) if self.n else np.zeros((0, 3), int)
        for i in range(self.n):
            c = tuple(self.centres[i])
            self.loc_vol[c] = i
            if st[i] == 0:
                self.rank_vol[c] = self.rank_id[i]

    def step(self, what, pix_th=0):
        st = self.state.numpy()
        if what == self.COUNT:
            alive = False
            for i in range(self.n):
                c = tuple(self.centres[i])
                if self.rank_vol[c] == self.NONE:
                    continue
                if self.dirty[c]:
                    hits = int(np.count_nonzero(self.run[self._win(c)] & self.pbits[i]))
                    if hits <= pix_th:
                        st[i] = 2
                        self.rank_vol[c] = self.NONE
                        continue
                alive = True
            self.dirty[:] = False
            self._alive = alive
        elif what == self.FILTER:
            from scipy import ndimage
            self.nbr_min = ndimage.minimum_filter(self.rank_vol, size=[2 * p - 1 for p in self.ps],
                                                  mode="constant", cval=self.NONE)
        else:
            cl = self.cleared.numpy()
            rad = [p // 2 for p in self.ps]
            ready = [i for i in range(self.n)
                     if self.rank_vol[tuple(self.centres[i])] != self.NONE and
                     self.nbr_min[tuple(self.centres[i])] == self.rank_vol[tuple(self.centres[i])]]
            for i in ready:
                c = self.centres[i]
                w = self._win(c)
                hit = self.run[w] & self.pbits[i]
                zz, yy, xx = np.nonzero(hit)
                gz_ = zz + c[0] - rad[0] + self.oz
                gy, gx = yy + c[1] - rad[1], xx + c[2] - rad[2]
                inside = (gz_ >= rad[0]) & (gz_ < self.gz - rad[0]) & (gy >= rad[1]) & \
                    (gy < self.shape[1] - rad[1]) & (gx >= rad[2]) & (gx < self.shape[2] - rad[2])
                cl[i] = int(np.count_nonzero(inside))
                self.run[w] &= ~hit
                box = tuple(slice(max(c[k] - (self.ps[k] - 1), 0), min(c[k] + self.ps[k], self.shape[k]))
                            for k in range(3))
                self.dirty[box] = True
                st[i] = 1
                self.rank_vol[tuple(c)] = self.NONE

    def alive(self):
        return bool(self._alive)

    def zone(self, imp, z_lo, z_hi, own, rank=None, mask=None, clean=None):
        n = (z_hi - z_lo) * self.shape[1] * self.shape[2]
        if n <= 0:
            return
        if not imp:
            if rank is not None:
                r = np.full((z_hi - z_lo,) + self.shape[1:], self.IMAX, dtype=np.int64)
                for z in range(z_lo, z_hi):
                    if own[0] <= z < own[1]:
                        r[z - z_lo] = self.rank_vol[z]
                rank.numpy()[:n] = r.reshape(-1)
            if mask is not None:
                mask.numpy()[:n] = self.run[z_lo:z_hi].reshape(-1)
                clean.numpy()[:n] = ~self.dirty[z_lo:z_hi].reshape(-1)
        else:
            if rank is not None:
                r = rank.numpy()[:n].astype(np.int64).reshape((z_hi - z_lo,) + self.shape[1:])
                self.rank_vol[z_lo:z_hi] = np.where(r == self.IMAX, self.NONE, r)
            if mask is not None:
                m = mask.numpy()[:n].reshape((z_hi - z_lo,) + self.shape[1:]) != 0
                cln = clean.numpy()[:n].reshape((z_hi - z_lo,) + self.shape[1:]) != 0
                self.run[z_lo:z_hi] &= m
                self.dirty[z_lo:z_hi] |= ~cln

    def close(self):
        self.mask_t.numpy()[~self.run] = 0


class OracleThinShard:
    """backend.ThinShard in NumPy: the count / filter / select steps of the sharded set-cover thinning
    (csrc/ppp_cover.hip, thin_* kernels) on a rank's local buffer, with the zone export / import."""
    COUNT, FILTER, SELECT = 0, 1, 2
    MAXC = 1 << 20
    NONE = 0x7F7F7F7F7F7F7F7F
    ZNONE = 0x7FFFFFFFFFFFFFFF

    def __init__(self, mask, lin_local, index_global, bits, P, global_z):
        self.mask_t = mask
        self.shape = (P.Z, P.Y, P.X)
        self.ps = (P.pz, P.py, P.px)
        self.oz, self.gz = P.origin_z, int(global_z)
        self.lin = lin_local.numpy().astype(np.int64)
        self.index = index_global.numpy().astype(np.int64)
        self.n = len(self.lin)
        C = int(np.prod(self.ps))
        b = bits.numpy().view(np.uint32).reshape(self.n, -1) if self.n else np.zeros((0, (C + 31) // 32), np.uint32)
        self.pbits = np.zeros((self.n, C), dtype=bool)
        for r in range(C):
            self.pbits[:, r] = (b[:, r // 32] >> np.uint32(r % 32)) & 1
        self.pbits = self.pbits.reshape((self.n,) + self.ps)
        self.state = torch.zeros(max(self.n, 1), dtype=torch.int32)
        self.count = torch.zeros(max(self.n, 1), dtype=torch.int32)
        self.cleared = torch.zeros(max(self.n, 1), dtype=torch.int32)
        self.run = self.mask_t.numpy() != 0
        self.key_vol = np.full(self.shape, self.NONE, dtype=np.int64)
        self.loc_vol = np.full(self.shape, -1, dtype=np.int64)
        self.dirty = np.ones(self.shape, dtype=bool)
        self.centres = np.stack(np.unravel_index(self.lin, self.shape), axis=1) if self.n else np.zeros((0, 3), int)
        for i in range(self.n):
            c = tuple(self.centres[i])
            self.loc_vol[c] = i
            self.key_vol[c] = (self.MAXC << 32) | int(self.index[i])
        self._alive = False

    def _win(self, c):
        rad = [p // 2 for p in self.ps]
        return tuple(slice(c[i] - rad[i], c[i] + rad[i] + 1) for i in range(3))

    def step(self, what):
        st, cnt, cl = self.state.numpy(), self.count.numpy(), self.cleared.numpy()
        if what == self.COUNT:
            alive = False
            for i in range(self.n):
                c = tuple(self.centres[i])
                if self.key_vol[c] == self.NONE:
                    continue
                if self.dirty[c]:
                    hits = int(np.count_nonzero(self.run[self._win(c)] & self.pbits[i]))
                    if hits == 0:
                        st[i] = 2
                        self.key_vol[c] = self.NONE
                        continue
                    self.key_vol[c] = ((self.MAXC - hits) << 32) | int(self.index[i])
                alive = True
            self.dirty[:] = False
            self._alive = alive
        elif what == self.FILTER:
            from scipy import ndimage
            self.nbr_min = ndimage.minimum_filter(self.key_vol, size=[2 * p - 1 for p in self.ps],
                                                  mode="constant", cval=self.NONE)
        else:
            rad = [p // 2 for p in self.ps]
            ready = [i for i in range(self.n)
                     if self.key_vol[tuple(self.centres[i])] != self.NONE and
                     self.nbr_min[tuple(self.centres[i])] == self.key_vol[tuple(self.centres[i])]]
            for i in ready:
                c = self.centres[i]
                w = self._win(c)
                hit = self.run[w] & self.pbits[i]
                zz, yy, xx = np.nonzero(hit)
                gz_ = zz + c[0] - rad[0] + self.oz
                gy, gx = yy + c[1] - rad[1], xx + c[2] - rad[2]
                inside = (gz_ >= rad[0]) & (gz_ < self.gz - rad[0]) & (gy >= rad[1]) & \
                    (gy < self.shape[1] - rad[1]) & (gx >= rad[2]) & (gx < self.shape[2] - rad[2])
                cl[i] = int(np.count_nonzero(inside))
                cnt[i] = self.MAXC - int(self.key_vol[tuple(c)] >> 32)
                self.run[w] &= ~hit
                box = tuple(slice(max(c[k] - (self.ps[k] - 1), 0), min(c[k] + self.ps[k], self.shape[k]))
                            for k in range(3))
                self.dirty[box] = True
                st[i] = 1
                self.key_vol[tuple(c)] = self.NONE

    def alive(self):
        return bool(self._alive)

    def zone(self, imp, z_lo, z_hi, own, key=None, mask=None, clean=None):
        n = (z_hi - z_lo) * self.shape[1] * self.shape[2]
        if n <= 0:
            return
        if not imp:
            if key is not None:
                r = np.full((z_hi - z_lo,) + self.shape[1:], self.ZNONE, dtype=np.int64)
                for z in range(z_lo, z_hi):
                    if own[0] <= z < own[1]:
                        r[z - z_lo] = self.key_vol[z]
                key.numpy()[:n] = r.reshape(-1)
            if mask is not None:
                mask.numpy()[:n] = self.run[z_lo:z_hi].reshape(-1)
                clean.numpy()[:n] = ~self.dirty[z_lo:z_hi].reshape(-1)
        else:
            if key is not None:
                r = key.numpy()[:n].astype(np.int64).reshape((z_hi - z_lo,) + self.shape[1:])
                self.key_vol[z_lo:z_hi] = np.where(r == self.ZNONE, self.NONE, r)
            if mask is not None:
                m = mask.numpy()[:n].reshape((z_hi - z_lo,) + self.shape[1:]) != 0
                cln = clean.numpy()[:n].reshape((z_hi - z_lo,) + self.shape[1:]) != 0
                self.run[z_lo:z_hi] &= m
                self.dirty[z_lo:z_hi] |= ~cln

    def close(self):
        self.mask_t.numpy()[~self.run] = 0
