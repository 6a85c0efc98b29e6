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

    def rank_patches(self, pred, cons, ov, P, score_box):
        full = self._ref_layout(cons, P)
        s = orc.rank(pred.numpy().astype(np.float32), full, ov.numpy(), self._ps(P), **self.kw)
        return torch.from_numpy(s)

    def patch_bits(self, pred, centres, thresh, P):
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

    def patch_graph(self, pred, cons, rows, P):
        full = self._ref_layout(cons, P)
        kw = dict(self.kw, origin=(P.origin_z, P.origin_y, P.origin_x))
        aff = orc.patch_graph(pred.numpy().astype(np.float32), full,
                              rows.numpy().view(np.uint32), self._ps(P), **kw)
        return torch.from_numpy(aff)

    def rank_order(self, score_dev, foreground, ps):
        score_host = score_dev.numpy()
        lin = backend.host_rank_order(score_host, foreground, ps)
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
