"""CPU: the oracle (oracle/) against the golden vectors produced by the reference.

Bit-exact on every float stage (consensus, scores, patch affinities) and on every
integer stage (ranking order, cover, thinning, pair set, instance map)."""
import hashlib

import numpy as np
import pytest

from conftest import same_partition
from oracle import ppp_oracle as orc


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_stages_match_reference(golden):
    g = golden
    out = orc.to_instance_seg(g.pred, g.foreground, g.foreground.copy(), g.numinst,
                              g.patchshape, **g.kw)
    early = int(g["early_out"])
    if early in (1, 2):
        assert "cons" not in out
        assert not out["instances"].any()
        return
    # S1
    cons_pos = orc.positive_planes(out["cons"], g.patchshape)
    used = np.zeros(out["cons"].shape[:3], dtype=bool)
    pz, py, px = g.patchshape
    for dz in range(0, pz):
        for dy in range(-(py - 1), py):
            for dx in range(-(px - 1), px):
                if (dz, dy, dx) > (0, 0, 0):
                    used[dz + pz - 1, dy + py - 1, dx + px - 1] = True
    assert not out["cons"][~used].any()
    if g.has("cons_pos"):
        assert np.array_equal(_bits(cons_pos), _bits(g["cons_pos"]))
    else:
        assert hashlib.sha256(np.ascontiguousarray(cons_pos).tobytes()).hexdigest() == \
            str(g["cons_pos_sha256"])
    # S2 + sort
    assert np.array_equal(_bits(out["scores"]), _bits(g["scores"]))
    assert np.array_equal(out["ranked_coords"], g["ranked_coords"])
    assert np.array_equal(_bits(out["ranked_scores"]), _bits(g["ranked_scores"]))
    # S3 / S4
    assert np.array_equal(out["cover_coords"], g["cover_coords"])
    if g.has("thin_coords"):
        assert np.array_equal(out["thin_coords"], g["thin_coords"])
    assert np.array_equal(out["selected_sorted"], g["selected_sorted"])
    if early == 3:
        assert "pairs" not in out
        return
    # pairs: canonical order equal; the reference's own set order equal as a set
    assert np.array_equal(out["pairs"], g["pairs"])
    ref = g["pairs_ref_order"]
    assert sorted(map(tuple, ref.tolist())) == sorted(map(tuple, out["pairs"].tolist()))
    # S5
    assert np.array_equal(_bits(out["aff"]), _bits(g["aff"]))
    # S6: identical ids (same canonical pair order), hence identical partition
    assert np.array_equal(out["instances"], g["instances"])
    assert same_partition(out["instances"], g["instances"])


def test_reference_set_order_gives_same_partition_or_paint(golden):
    """The reference's own pair order (a Python set) only changes the paint order of
    components; where no two components overlap the maps agree up to permutation."""
    g = golden
    if int(g["early_out"]) != 0:
        pytest.skip("early-out case")
    ref_order = g["pairs_ref_order"]
    aff = orc.patch_graph(g.pred, _cons_ref_layout(g), ref_order, g.patchshape, **g.kw) \
        if g.has("cons_pos") else None
    if aff is None:
        pytest.skip("consensus stored as hash only")
    inst = orc.label(ref_order, aff, g.pred, g.patchshape, g.foreground.shape, **g.kw)
    assert np.array_equal(inst, g["e2e_instances_ref_order"])


def _cons_ref_layout(g):
    pz, py, px = g.patchshape
    ns = orc.neighshape_of(g.patchshape)
    cons = np.zeros(tuple(ns) + g.foreground.shape, dtype=np.float32)
    k = 0
    for dz in range(0, pz):
        for dy in range(-(py - 1), py):
            for dx in range(-(px - 1), px):
                if (dz, dy, dx) <= (0, 0, 0):
                    continue
                cons[dz + pz - 1, dy + py - 1, dx + px - 1] = g["cons_pos"][k]
                k += 1
    return cons


@pytest.mark.parametrize("name", ["c2d_p5_blobs", "c3d_p3_blobs"])
def test_reference_numpy_path_agrees_on_separated_instances(name):
    """The reference's NumPy path (cuda=False: int16 +-1 votes, integer ranks -- different
    arithmetic from the kernels) and the kernel semantics give the same
    partition of two well separated instances wherever both label a voxel.  This is the
    'plumbing' relation of BASELINE config [0]; it is not a bit-parity claim."""
    from conftest import Golden
    g = Golden(name)
    a, b = g["instances_cpu_path"], g["instances"]
    both = (a > 0) & (b > 0)
    assert both.sum() > 0.9 * max((a > 0).sum(), (b > 0).sum())
    assert len(np.unique(a[a > 0])) == len(np.unique(b[b > 0])) == 2
    assert same_partition(np.where(both, a, 0), np.where(both, b, 0))


def test_gather_form_consensus_equals_scatter_form(golden):
    """ppp_oracle_fill_consensus_planes (S1 as a gather over offset planes, the form the CPU
    baseline runs on all cores) == the serial scatter restatement, bit for bit, on every golden
    (all background rules, value rules, overlap on / off, 2-d and 3-d)."""
    from oracle import ppp_oracle as orc
    g = golden
    if g.has("early_out") and int(g["early_out"]):
        pytest.skip("early-out case")
    if int(np.prod(g.patchshape)) > 400:
        pytest.skip("serial scatter form too slow for this patch size here")
    a = orc.consensus(g.pred, g.overlap_mask, g.patchshape, **g.kw)
    b = orc.consensus_planes(g.pred, g.overlap_mask, g.patchshape, **g.kw)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
