"""More than 65 535 instance ids: the uint32 id path of the blockwise / stitched entry.

Reference: the whole-volume entry paints into a uint16 map (vote_instances.py:230, with
np.seterr(over='raise') at :37), the blockwise driver into a uint32 one
(stitch_patch_graph.py:120) that is optionally compacted (remove_small_components + relabel,
:831-834) before every dataset is written as uint16 (:852-870).  With the shipped
``mws = true`` + ``includeSinglePatchCCS = true`` every selected patch is issued an id
(graph_mws.py:34-41), so large volumes need the uint32 path.

The case: > 65 536 isolated two-pixel instances, so far apart that no two selected patches form a
pair -- every instance is one self-pair component.  The labelling of the pair list is checked
against the oracle's (connected components / host mutex watershed + in-order painting)."""
import numpy as np
import pytest

from patchperpix_amd import synth, tiling
from patchperpix_amd.flags import FLYLIGHT, FLYLIGHT_CC


def isolated_blobs(n_side, ps=(1, 3, 3), spacing=7, seed=5):
    """n_side x n_side instances of 1 x 1 x 2 voxels on a grid: selected patches of different
    instances are more than 2 p apart on an axis (no pair rows between them)."""
    Y = X = n_side * spacing + 4
    lab = np.zeros((1, Y, X), dtype=np.int64)
    yy, xx = np.meshgrid(np.arange(n_side), np.arange(n_side), indexing="ij")
    ids = (yy * n_side + xx + 1).astype(np.int64)
    for dx in range(2):
        lab[0, 3 + yy * spacing, 3 + xx * spacing + dx] = ids
    pred = synth.pred_from_labels(lab, ps, seed=seed)
    fg = lab != 0
    return dict(pred=pred, foreground=fg, numinst=fg.astype(np.uint8), labels=lab)


def expected_map(pairs, aff, pred, ps, shape, kw):
    """Oracle labelling of a pair list into a uint32 map (components in networkx's order, or the
    host mutex watershed; painted in id order, later ids overwrite)."""
    from oracle import ppp_oracle as orc
    from patchperpix_amd import backend
    if kw["mws"]:
        # (the oracle's pure-Python watershed is quadratic in the number of ids; the C++ host
        # version is pinned to the reference's goldens in tests/test_abi_and_host.py)
        nodes, labels, _ = backend.host_mws(pairs, aff, shape)
        order = np.argsort(labels, kind="stable")
        ccs_iter = [(int(labels[i]), [tuple(int(v) for v in nodes[i])]) for i in order]
    else:
        ccs = orc.connected_components(pairs, aff)
        ccs_iter = [(k + 1, cc) for k, cc in enumerate(ccs)]
    rad = np.array([p // 2 for p in ps])
    inst = np.zeros(shape, dtype=np.uint32)
    th = np.float32(kw["patch_threshold"])
    for lab, cc in ccs_iter:
        for c in cc:
            patch = pred[(slice(None),) + tuple(c)].reshape(ps) > th
            win = tuple(slice(int(c[i] - rad[i]), int(c[i] + rad[i] + 1)) for i in range(3))
            inst[win][patch] = lab
    return inst


@pytest.mark.parametrize("flags", [FLYLIGHT_CC, FLYLIGHT], ids=["cc", "mws"])
def test_more_than_65535_ids_cpu(flags):
    """assemble() with the oracle standing in for the kernels: uint16 refuses, uint32 carries."""
    import torch
    from oracle_ops import OracleOps
    ps = [1, 3, 3]
    c = isolated_blobs(257)
    kw = dict(flags, skipThinCover=True)
    shape = c["foreground"].shape

    class CachedOps(OracleOps):          # three assemblies of the same volume: S1 / S2 once
        memo = {}

        def consensus(self, pred, ov, P):
            if "c" not in self.memo:
                self.memo["c"] = OracleOps.consensus(self, pred, ov, P)
            return self.memo["c"]

        def rank_patches(self, pred, cons, ov, P, score_box):
            if "r" not in self.memo:
                self.memo["r"] = OracleOps.rank_patches(self, pred, cons, ov, P, score_box)
            return self.memo["r"]

        def patch_graph(self, pred, cons, rows, P):
            if "g" not in self.memo:
                self.memo["g"] = OracleOps.patch_graph(self, pred, cons, rows, P)
            return self.memo["g"]

    ops = CachedOps(**kw)
    slabs = tiling.plan_slabs(1, 1)
    args = lambda: (torch.from_numpy(c["pred"]), 0, shape, c["foreground"].copy(),
                    c["foreground"].copy(), c["numinst"], ps, slabs)
    pairs, aff = tiling.assemble(*args(), ops=ops, return_intermediates=True, **kw)
    assert len(pairs) > 65535 and np.all(pairs[:, :3] == pairs[:, 3:])     # self-pairs only
    want = expected_map(pairs, aff, c["pred"], ps, shape, kw)
    assert len(np.unique(want)) - 1 == 257 * 257
    with pytest.raises(OverflowError):
        tiling.assemble(*args(), ops=ops, **kw)
    got, fg = tiling.assemble(*args(), ops=ops, _instances_dtype=np.uint32, **kw)
    assert got.dtype == np.uint32 and np.array_equal(got, want)
    assert int(got.max()) == 257 * 257


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [FLYLIGHT_CC, FLYLIGHT], ids=["cc", "mws"])
@pytest.mark.parametrize("tiles", [None, (2, 2)])
def test_more_than_65535_ids_gpu(flags, tiles):
    """The same through the drop-in entry point with the real kernels, fused and y/x-tiled."""
    from patchperpix_amd.vote_instances import vote_instances as vi
    ps = [1, 3, 3]
    c = isolated_blobs(257)
    kw = dict(flags, skipThinCover=True)
    if tiles:
        kw.update(_n_slabs=1, _yx_tiles=tiles)
    shape = c["foreground"].shape
    call = lambda **extra: vi.to_instance_seg(c["pred"].copy(), c["foreground"].copy(),
                                              c["foreground"].copy(), c["numinst"].copy(), ps,
                                              **dict(kw, **extra))
    pairs, aff = call(return_intermediates=True)
    assert len(pairs) > 65535
    want = expected_map(pairs, aff, c["pred"], ps, shape, kw)
    with pytest.raises(OverflowError):
        call()
    got, fg = call(_instances_dtype=np.uint32)
    assert got.dtype == np.uint32 and np.array_equal(got, want)
    assert len(np.unique(got)) - 1 == 257 * 257


def test_stitched_entry_compacts_uint32_ids(tmp_path, monkeypatch):
    """stitch_main: uint32 map -> remove_small_components -> relabel -> uint16 datasets
    (stitch_patch_graph.py:120, 831-870).  The assembly itself is replaced by a stub that returns
    ids far above 65 535."""
    from patchperpix_amd import postprocess
    shape = (4, 12, 12)
    inst32 = np.zeros(shape, dtype=np.uint32)
    inst32[1, 2:6, 2:6] = 70001
    inst32[2, 6:10, 6:10] = 400123
    inst32[3, 1, 1] = 99999                     # a one-voxel instance: removed
    fg = inst32 > 0

    def fake_tiled(pred, fg_, mask, numinst, ps, n_slabs, **kw):
        assert np.dtype(kw["_instances_dtype"]) == np.uint32
        return inst32.copy(), fg_.astype(np.uint8)
    monkeypatch.setattr(tiling, "to_instance_seg_tiled", fake_tiled)
    written = {}
    from patchperpix_amd.vote_instances import vote_instances as vi
    monkeypatch.setattr(vi, "write_result", lambda fn, ds: written.update(ds))
    pred = np.zeros((27,) + shape, dtype=np.float32)
    pred[13] = fg
    np.save(tmp_path / "p.npy", pred)
    out = tiling.stitch_main(str(tmp_path / "p.npy"), result_folder=str(tmp_path),
                             patchshape=[3, 3, 3], patch_threshold=0.5, remove_small_comps=2,
                             cuda=True)
    assert set(np.unique(out)) == {0, 1, 2}
    assert written["vote_instances"].dtype == np.uint16
    assert np.array_equal(written["vote_instances"] > 0, postprocess.remove_small_components(inst32, 2) > 0)
    assert written["vote_instances"][1, 3, 3] == 1 and written["vote_instances"][2, 7, 7] == 2
