"""The reference's BLOCKWISE semantics (patchperpix_amd.blockwise) against goldens produced by
running the reference's own blockwise driver (tests/golden/gen_golden_blockwise.py): every
block's stored pair rows and affinities (bit patterns), the inter-block rows, the on-disk layout
of the block graph, resume from existing blocks, and the final instance map.

CPU: the per-block assembly (`_do_block`) is served by the CPU oracle; GPU (-m gpu): by the real
kernels through ``to_instance_seg``."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from patchperpix_amd import blockwise, minizarr


def bw_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "bw_*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    kw = json.loads(str(z["flags"]))
    kw.setdefault("max_total_patch_distance_in_ps_multiples", 2)
    return z, kw


def oracle_do_block(block, foreground, numinst, patchshape, kw, **extra):
    from oracle import ppp_oracle as orc
    k = dict(kw, **extra)
    k.pop("patchshape", None)
    res = orc.to_instance_seg(block, foreground, foreground.copy(), numinst, [int(p) for p in patchshape], **k)
    if "pairs" not in res:
        return None, None
    return res["pairs"], res["aff"]


def run(name, tmp_path, do_block=None, monkeypatch=None):
    z, kw = load(name)
    pred_file = str(tmp_path / "sample.zarr")
    g = minizarr.open(pred_file, "w")
    pred16 = z["pred_f16"]
    g.create_dataset("volumes/pred_affs", data=pred16, chunks=(pred16.shape[0], 8, 8, 8))
    out_dir = str(tmp_path / "out")
    written = {}
    from patchperpix_amd.vote_instances import vote_instances as vi
    if monkeypatch is not None:
        monkeypatch.setattr(vi, "write_result", lambda fn, ds: written.update(ds))
        if do_block is not None:
            monkeypatch.setattr(blockwise, "_do_block", do_block)
    inst = blockwise.main(pred_file, result_folder=out_dir, **kw)
    return z, kw, inst, written, out_dir


def check_blocks(z, out_dir):
    res = minizarr.open(os.path.join(out_dir, "sample.zarr"), "r")
    for key in [str(k) for k in z["block_keys"]]:
        path = "volumes/blocks/" + key
        tag = key.replace("/", "__")
        pairs = np.asarray(res[path + "/patch_pairs"][...])
        aff = np.asarray(res[path + "/aff_graph_mat"][...])
        assert pairs.dtype == np.uint32 and aff.dtype == np.float32
        assert np.array_equal(pairs, z["pairs__" + tag]), key
        assert np.array_equal(aff.view(np.uint32), z["aff__" + tag].view(np.uint32)), key
        assert "block_shape" in res[path + "/patch_pairs"].attrs
    # nothing the reference did not write
    blocks = res["volumes/blocks"]
    have = []
    for bk in sorted(blocks.keys()):
        names = sorted(blocks[bk].keys())
        if "patch_pairs" in names:
            have.append(bk)
        have += [bk + "/" + n for n in names if n not in ("patch_pairs", "aff_graph_mat")]
    assert sorted(have) == sorted(str(k) for k in z["block_keys"])


@pytest.mark.parametrize("name", bw_names())
def test_blockwise_matches_reference_cpu(name, tmp_path, monkeypatch):
    monkeypatch.setattr(blockwise, "label_graph", cpu_label_graph)
    z, kw, inst, written, out_dir = run(name, tmp_path, oracle_do_block, monkeypatch)
    check_blocks(z, out_dir)
    assert inst.dtype == np.uint32
    assert np.array_equal(written["vote_instances"], z["instances_u16"])
    assert np.array_equal(written["vote_foreground"] != 0, z["vote_foreground"] != 0)


def cpu_label_graph(vol, rows, aff, shape, kwargs):
    """label_graph without a device: networkx-order components / host mutex watershed + painting in
    NumPy (the oracle's), on the de-duplicated edge list."""
    from oracle import ppp_oracle as orc
    from patchperpix_amd import backend
    rows, aff = blockwise.dedupe_edges(rows, aff, shape)
    ps = [int(p) for p in kwargs["patchshape"]]
    pred = np.asarray(vol.affs[...]).astype(np.float32)
    if kwargs.get("mws"):
        ccs = orc.mutex_watershed(rows, aff)
    else:
        ccs = orc.connected_components(rows, aff)
    return orc.paint_instances(ccs, pred, ps, shape, kwargs["patch_threshold"], dtype=np.uint32)


def test_dedupe_edges_is_networkx_add_edge():
    rows = np.array([[1, 1, 1, 2, 2, 2], [3, 3, 3, 1, 1, 1], [2, 2, 2, 1, 1, 1], [4, 4, 4, 4, 4, 4],
                     [1, 1, 1, 3, 3, 3], [5, 5, 5, 1, 1, 1]], dtype=np.uint32)
    aff = np.array([0.5, -0.25, 0.75, 0.0, 0.125, 1.0], dtype=np.float32)
    r, a = blockwise.dedupe_edges(rows, aff, (8, 8, 8))
    # (1,2): first at 0, last value 0.75; (3,1): first at 1, last value 0.125; the zero row is dropped
    assert r.tolist() == [[1, 1, 1, 2, 2, 2], [3, 3, 3, 1, 1, 1], [5, 5, 5, 1, 1, 1]]
    assert a.tolist() == [0.75, 0.125, 1.0]


def test_resume_skips_existing_blocks(tmp_path, monkeypatch):
    """stitch_patch_graph.py:584-587, 194-201: blocks and inter-block rows already in the result
    zarr are not computed again."""
    name = "bw_p3_cc_overlap"
    monkeypatch.setattr(blockwise, "label_graph", cpu_label_graph)
    calls = []

    def counting(*a, **k):
        calls.append(1)
        return oracle_do_block(*a, **k)
    z, kw, inst, written, out_dir = run(name, tmp_path, counting, monkeypatch)
    n_first = len(calls)
    assert n_first > 0
    del calls[:]
    inst2 = blockwise.main(str(tmp_path / "sample.zarr"), result_folder=out_dir, **kw)
    assert len(calls) == 0 and np.array_equal(inst, inst2)


@pytest.mark.gpu
@pytest.mark.parametrize("name", bw_names())
def test_blockwise_matches_reference_gpu(name, tmp_path, monkeypatch):
    """The same with the real kernels: per-block to_instance_seg(return_intermediates), the
    inter-block S1 + S5 on injected patches and pairs, device labelling and painting."""
    z, kw, inst, written, out_dir = run(name, tmp_path, None, monkeypatch)
    check_blocks(z, out_dir)
    assert np.array_equal(written["vote_instances"], z["instances_u16"])
    # through the package's drop-in entry point as well
    from patchperpix_amd.vote_instances import stitch_patch_graph as spg
    out2 = str(tmp_path / "out2")
    inst2 = spg.main(str(tmp_path / "sample.zarr"), result_folder=out2, blockwise_semantics="reference", **kw)
    assert np.array_equal(inst2, inst)


def test_patch_rows_are_gathered_chunk_by_chunk():
    """label_graph's patch table: row k = affs[:, node k], every touched chunk of the store read
    once (and only the bounding box of its nodes) -- no dense (C, Z, Y, X) scratch."""
    from patchperpix_amd import blockwise as bw
    rng = np.random.default_rng(3)
    data = rng.random((27, 20, 33, 41)).astype(np.float16)

    class Chunked:
        shape, chunks, dtype = data.shape, (27, 8, 16, 16), data.dtype
        reads = []

        def __getitem__(self, sel):
            self.reads.append(sel)
            return data[sel]
    arr = Chunked()
    lin = rng.choice(20 * 33 * 41, size=500, replace=False)
    nodes = np.stack(np.unravel_index(lin, (20, 33, 41)), axis=1).astype(np.int64)
    rows = bw.gather_patch_rows(arr, nodes)
    assert rows.dtype == np.float32 and np.array_equal(rows, data[:, nodes[:, 0], nodes[:, 1], nodes[:, 2]].T.astype(np.float32))
    n_chunks = len({(z // 8, y // 16, x // 16) for z, y, x in nodes.tolist()})
    assert len(arr.reads) == n_chunks
    order = bw._chunk_order(arr, nodes)
    assert sorted(order.tolist()) == list(range(500))
    cid = [(z // 8, y // 16, x // 16) for z, y, x in nodes[order].tolist()]
    assert cid == sorted(cid)                                      # nodes of a chunk are consecutive
    # an ndarray is one chunk
    assert np.array_equal(bw.gather_patch_rows(data, nodes[:7]), rows[:7])
