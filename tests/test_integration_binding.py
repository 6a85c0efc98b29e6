"""INTEGRATION.md path B, executed: the reference-side ctypes binding
(integration/ppp_reference_binding.py -- cuda_code.py replacement + the three launchers with the
reference's argument lists and consensus layout) against the goldens."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import REPO, Golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["c3d_p3_cells", "c2d_p5_th09_inv", "c3d_p5_cells", "c3d_p3_overlap"])
def test_reference_side_binding_matches_goldens(name):
    import torch
    spec = importlib.util.spec_from_file_location(
        "ppp_reference_binding", os.path.join(REPO, "integration", "ppp_reference_binding.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    g = Golden(name)
    ctx = b.init_cuda()
    ps = g.patchshape
    neigh = [2 * p for p in ps] if ps[0] > 1 else [1, 2 * ps[1], 2 * ps[2]]    # vote_instances.py:249-253
    pred = torch.from_numpy(g.pred.astype(np.float32)).cuda()
    cons = b.create_consensus_array_cuda(pred, g.overlap_mask, ps, neigh, **g.kw)
    assert tuple(cons.shape) == tuple(neigh) + g.pred.shape[1:]
    scores = b.rank_patches_cuda(pred, cons, ps, neigh, g.overlap_mask, **g.kw)
    b.sync(ctx)
    assert np.array_equal(scores.cpu().numpy().view(np.uint32), g["scores"].astype(np.float32).view(np.uint32))
    if g.has("cons_pos"):
        from oracle import ppp_oracle as orc
        got = orc.positive_planes(cons.cpu().numpy(), ps)
        assert np.array_equal(got.view(np.uint32), g["cons_pos"].view(np.uint32))
    aff = b.computePatchGraph_cuda(pred, cons, g["pairs"], ps, neigh, **g.kw)
    assert np.array_equal(aff.cpu().numpy().view(np.uint32), g["aff"].astype(np.float32).view(np.uint32))
    b.delete_cuda(ctx)


def test_reference_side_binding_of_the_numpy_stages():
    """the cuda=False launchers of the binding (create_consensus_array / rank_patches with the
    reference's array layout and list format) against a golden of the reference's own functions"""
    import json
    import torch
    from conftest import GOLDEN_DIR
    spec = importlib.util.spec_from_file_location(
        "ppp_reference_binding", os.path.join(REPO, "integration", "ppp_reference_binding.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    z = np.load(os.path.join(GOLDEN_DIR, "np_c2d_p5_th09.npz"))
    kw = json.loads(str(z["flags"]))
    ps = [int(p) for p in z["patchshape"]]
    neigh = [2 * p for p in ps] if ps[0] > 1 else [1, 2 * ps[1], 2 * ps[2]]
    fg = z["foreground"].astype(bool)
    pred = torch.from_numpy(z["pred_f16"].astype(np.float32)).cuda()
    b.init_cuda()
    b.create_consensus_array.patch_threshold = kw["patch_threshold"]
    full, votes = b.create_consensus_array(pred, fg, fg.shape, ps, neigh)
    want = np.zeros(tuple(int(v) for v in z["cons_shape"]), dtype=np.int16)
    idx = z["cons_index"]
    want[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]] = z["cons_value"]
    assert full.dtype == np.int16 and np.array_equal(full, want)
    rad = [p // 2 for p in ps]
    every = np.transpose(np.where(fg))
    all_patches = [p for p in every if np.all(p >= rad) and np.all(p < np.array(fg.shape) - rad)]
    ranked = b.rank_patches(pred, fg, votes, all_patches, ps)
    assert [list(c) for c, _ in ranked] == z["ranked_coords"].tolist()
    assert [s for _, s in ranked] == z["ranked_scores"].tolist()
