"""GPU (MI355X): the HIP kernels, called through the C ABI, against
 (a) the golden vectors produced by the reference itself, and
 (b) the CPU oracle on fresh seeded inputs,
bit-exact on every float stage (compared as uint32 bit patterns; tolerance 0 ulp) and on
every integer stage."""
import hashlib
import os

import numpy as np
import pytest

from conftest import same_partition

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    from patchperpix_amd import backend
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    assert backend.device_count() >= 1
    return torch


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _stage_outputs(torch, pred_host, overlap_mask, patchshape, kw, pairs=None, f16=False):
    from patchperpix_amd import backend
    P = backend.make_params(pred_host.shape[1:], patchshape, **kw)
    pred = _dev(torch, pred_host.astype(np.float16) if f16 else pred_host.astype(np.float32))
    ov = _dev(torch, (overlap_mask > 0).astype(np.uint8)) if P.use_overlap else None
    cons = backend.consensus(pred, ov, P)
    s1_kernel = backend.lib().ppp_consensus_kernel_name().decode()
    score = backend.rank_patches(pred, cons, ov, P)
    out = dict(cons=cons.cpu().numpy(), score=score.cpu().numpy(), P=P, pred=pred, cons_dev=cons,
               s1_kernel=s1_kernel)
    if backend.rank_vm_available(P):
        # the row-stationary ranking kernel on the voxel-major layout: same bits as the gather kernel
        vm0, Pv0 = backend.cons_to_voxel_major(cons, P)
        score_vm = backend.rank_patches(pred, vm0, ov, Pv0).cpu().numpy()
        assert np.array_equal(out["score"].view(np.uint32), score_vm.view(np.uint32))
        del vm0
    if pairs is not None and len(pairs):
        pd = _dev(torch, np.ascontiguousarray(pairs, dtype=np.uint32).view(np.int32))
        out["aff"] = backend.patch_graph(pred, cons, pd, P).cpu().numpy()          # row order
        order = backend.pair_order(pd, P)                                         # grouped
        aff2 = backend.patch_graph(pred, cons, pd, P, order=order).cpu().numpy()
        assert np.array_equal(out["aff"].view(np.uint32), aff2.view(np.uint32))
        vm, Pv = backend.cons_to_voxel_major(cons, P)                             # re-layout
        aff3 = backend.patch_graph(pred, vm, pd, Pv, order=order).cpu().numpy()
        assert np.array_equal(out["aff"].view(np.uint32), aff3.view(np.uint32))
        if P.px in (3, 5, 7, 9) or (P.px == 25 and P.pz == 1):                    # per-patch kernel
            aff4 = backend.patch_graph_by_patch(pred, vm, pd, Pv).cpu().numpy()
            assert np.array_equal(out["aff"].view(np.uint32), aff4.view(np.uint32))
            # ... with the thinning decisions made inside the kernel (no masks beforehand), and
            # with mask budgets that cut the groups into several batches
            for budget in ("0", "65536", "4194304"):
                os.environ["PPP_PA_LCG_BYTES"] = budget
                try:
                    aff5 = backend.patch_graph_by_patch(pred, vm, pd, Pv).cpu().numpy()
                finally:
                    del os.environ["PPP_PA_LCG_BYTES"]
                assert np.array_equal(out["aff"].view(np.uint32), aff5.view(np.uint32)), budget
    return out


def test_kernels_match_reference_goldens(golden, torch_cuda):
    g = golden
    if int(g["early_out"]) in (1, 2):
        pytest.skip("early-out case (covered by the end-to-end test)")
    pairs = g["pairs"] if g.has("pairs") else None
    for f16 in (False, True):   # the goldens' inputs are float16-representable
        o = _stage_outputs(torch_cuda, g.pred, g.overlap_mask, g.patchshape, g.kw, pairs, f16)
        if g.has("cons_pos"):
            assert np.array_equal(_bits(o["cons"]), _bits(g["cons_pos"]))
        else:
            assert hashlib.sha256(np.ascontiguousarray(o["cons"]).tobytes()).hexdigest() == \
                str(g["cons_pos_sha256"])
        assert np.array_equal(_bits(o["score"]), _bits(g["scores"]))
        if pairs is not None:
            assert np.array_equal(_bits(o["aff"]), _bits(g["aff"]))


@pytest.mark.parametrize("pipeline", ["fused", "stages"])
def test_end_to_end_matches_reference(golden, torch_cuda, monkeypatch, pipeline):
    """to_instance_seg through the drop-in entry point: identical instance ids (the pair order
    is the canonical one the golden was generated with), hence identical partition.  Both the
    device-resident pipeline and the one that goes through the reference's stage functions."""
    from patchperpix_amd.vote_instances import vote_instances as vi
    monkeypatch.setenv("PPP_PIPELINE", pipeline)
    g = golden
    kw = dict(g.kw, debug=False, isbiHack=False, save_no_intermediates=True, sample=1.0,
              result_folder="/tmp", affinities="x.zarr")
    inst, fg = vi.to_instance_seg(g.pred.copy(), g.foreground.copy(), g.foreground.copy(),
                                  g.numinst.copy(), g.patchshape, **kw)
    if g.has("instances"):
        assert inst.dtype == np.uint16
        assert np.array_equal(inst, g["instances"])
        assert same_partition(inst, g["instances"])
        assert np.array_equal(fg, g["foreground_out"])
    else:
        assert not inst.any()
    # return_intermediates hands back (pairs, aff) like the reference
    res = vi.to_instance_seg(g.pred.copy(), g.foreground.copy(), g.foreground.copy(),
                             g.numinst.copy(), g.patchshape,
                             **dict(kw, return_intermediates=True))
    if g.has("aff"):
        assert np.array_equal(res[0], g["pairs"])
        assert np.array_equal(_bits(res[1]), _bits(g["aff"]))
    else:
        assert res == (None, None)


CASES = [
    # shape, patchshape, synth kwargs, flag overrides
    ((20, 22, 24), (5, 5, 5), dict(seed=21, cell=[8, 8, 8], overlap_frac=0.02), {}),
    ((9, 40, 70), (3, 7, 5), dict(seed=22, cell=[4, 10, 9], noise=0.3),
     dict(patch_threshold=0.8, vi_bg_use_inv_th=True, vi_bg_use_less_than_th=False)),
    ((1, 64, 67), (1, 9, 9), dict(seed=23, cell=[1, 14, 14], noise=0.2),
     dict(patch_threshold=0.6, vi_bg_use_half_th=True, vi_bg_use_less_than_th=False,
          rank_int_counter=True, overlapping_inst=False)),
    ((18, 18, 18), (7, 7, 7), dict(seed=24, cell=[9, 9, 9]),
     dict(consensus_norm_prob_product=False, consensus_prob_product=True)),
    ((13, 14, 17), (9, 9, 9), dict(seed=25, cell=[12, 12, 12], overlap_frac=0.01), {}),
    ((1, 70, 75), (1, 25, 25), dict(seed=26, cell=[1, 30, 30]),
     dict(patch_threshold=0.9, vi_bg_use_inv_th=True, vi_bg_use_less_than_th=False,
          overlapping_inst=False)),     # wormbodies-like 2-d shape: the generic kernels
    # anisotropic patches through the specialised kernels (px = 7 / 9, pz, py smaller)
    ((10, 16, 30), (3, 5, 7), dict(seed=27, cell=[4, 7, 9], overlap_frac=0.02), {}),
    ((12, 17, 21), (5, 9, 9), dict(seed=28, cell=[6, 11, 11]), {}),
    # a patch width without a specialised kernel: the generic gather kernels
    ((1, 40, 44), (1, 11, 11), dict(seed=29, cell=[1, 13, 13]), dict(overlapping_inst=False)),
]


# the S1 kernel that serves each case (ppp_consensus_kernel_name): every family meets the oracle
# directly, not only through the kernel it replaced
S1_KERNEL = ["consensus_v3_kernel", "consensus_v2_kernel", "consensus_v2_kernel", "consensus_v2_kernel",
             "consensus_v3_kernel", "consensus_wide_kernel", "consensus_v3_kernel", "consensus_v3_kernel",
             "consensus_gather_kernel"]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_kernels_match_oracle_on_fresh_inputs(case, torch_cuda):
    from oracle import ppp_oracle as orc
    from patchperpix_amd import synth
    from tests_flags import FLYLIGHT
    shape, ps, skw, flags = CASES[case]
    kw = dict(FLYLIGHT, **flags)
    c = synth.make_case(shape, ps, **skw)
    # non-float16 values too: perturb in float32 so products are not exactly representable
    rng = np.random.default_rng(case)
    pred = (c["pred"] * rng.uniform(0.97, 1.0, size=c["pred"].shape)).astype(np.float32)
    ov = 1 * (c["numinst"] > 1)
    ref = orc.to_instance_seg(pred, c["foreground"], c["foreground"].copy(), c["numinst"], ps, **kw)
    assert "aff" in ref
    o = _stage_outputs(torch_cuda, pred, ov, ps, kw, ref["pairs"])
    assert o["s1_kernel"] == S1_KERNEL[case]
    assert np.array_equal(_bits(o["cons"]), _bits(orc.positive_planes(ref["cons"], ps)))
    assert np.array_equal(_bits(o["score"]), _bits(ref["scores"]))
    assert np.array_equal(_bits(o["aff"]), _bits(ref["aff"]))
    # full path on the device side
    from patchperpix_amd.vote_instances import vote_instances as vi
    inst, _ = vi.to_instance_seg(pred.copy(), c["foreground"].copy(), c["foreground"].copy(),
                                 c["numinst"].copy(), ps,
                                 **dict(kw, save_no_intermediates=True, sample=1.0,
                                        result_folder="/tmp", affinities="x.zarr"))
    assert np.array_equal(inst, ref["instances"])


def test_reference_layout_count_and_tiles(torch_cuda):
    """PPP_CONS_REFERENCE layout, the separate count output, and a consensus TILE (cons_box)
    agree with the whole-volume compact result."""
    import ctypes
    from patchperpix_amd import backend, synth
    from tests_flags import FLYLIGHT
    from oracle import ppp_oracle as orc
    torch = torch_cuda
    shape, ps = (12, 14, 16), (3, 3, 3)
    c = synth.make_case(shape, ps, seed=31, cell=[5, 5, 5], overlap_frac=0.02)
    kw = dict(FLYLIGHT)
    ov_h = (c["numinst"] > 1).astype(np.uint8)
    pred, ov = _dev(torch, c["pred"]), _dev(torch, ov_h)
    P = backend.make_params(shape, ps, **kw)
    cons, cnt = backend.consensus(pred, ov, P, want_count=True)
    ref = orc.consensus(c["pred"], ov_h, ps, **kw)
    assert np.array_equal(_bits(backend.cons_to_reference(cons, P).cpu().numpy()), _bits(ref))
    Pr = backend.make_params(shape, ps, cons_layout=backend.CONS_REFERENCE, **kw)
    assert np.array_equal(_bits(backend.consensus(pred, ov, Pr).cpu().numpy()), _bits(ref))
    # counts: the reference's -DOUTPUT_CNT pass
    kw_raw = dict(kw, consensus_norm_aff=False, consensus_interleaved_cnt=False)
    Praw = backend.make_params(shape, ps, **kw_raw)
    raw, cnt2 = backend.consensus(pred, ov, Praw, want_count=True)
    assert np.array_equal(cnt.cpu().numpy(), cnt2.cpu().numpy())
    with np.errstate(invalid="ignore", divide="ignore"):
        expect = np.where(cnt2.cpu().numpy() != 0, raw.cpu().numpy() / cnt2.cpu().numpy(),
                          raw.cpu().numpy())
    assert np.array_equal(_bits(expect), _bits(cons.cpu().numpy()))
    # a tile of bases + scores for the centres it supports
    box = (2, 3, 1, 9, 12, 13)
    Pt = backend.make_params(shape, ps, cons_box=box, **kw)
    tile = backend.consensus(pred, ov, Pt).cpu().numpy()
    assert np.array_equal(_bits(tile), _bits(cons.cpu().numpy()[:, 2:9, 3:12, 1:13]))
    full_score = backend.rank_patches(pred, cons, ov, P).cpu().numpy()
    sb = (3, 4, 2, 8, 11, 12)
    part = backend.rank_patches(pred, _dev(torch, tile), ov, Pt, score_box=sb).cpu().numpy()
    assert np.array_equal(_bits(part[3:8, 4:11, 2:12]), _bits(full_score[3:8, 4:11, 2:12]))
    with pytest.raises(RuntimeError, match="does not cover"):
        backend.rank_patches(pred, _dev(torch, tile), ov, Pt, score_box=(1, 4, 2, 8, 11, 12))


def test_union_find_labels_match_oracle(torch_cuda):
    """ppp_label_components on a random sparse graph with many components, zero and negative
    edges and self loops: same components, same enumeration order as the oracle (which is
    pinned to networkx through the goldens)."""
    from patchperpix_amd import backend
    from patchperpix_amd.vote_instances.aff_patch_graph import AffGraph
    from patchperpix_amd.vote_instances.graph_to_labeling import component_labels
    from oracle import ppp_oracle as orc
    torch = torch_cuda
    rng = np.random.default_rng(5)
    shape = (24, 24, 24)
    nodes = rng.integers(3, 21, size=(400, 3))
    i = rng.integers(0, 400, size=3000)
    j = rng.integers(0, 400, size=3000)
    pairs = np.concatenate([nodes[i], nodes[j]], axis=1).astype(np.uint32)
    aff = rng.choice([0.0, -0.5, 0.25, 0.75], p=[0.3, 0.4, 0.2, 0.1], size=3000).astype(np.float32)
    aff[::97] = 0.5
    pairs[::97, 3:] = pairs[::97, :3]          # self loops
    P = backend.make_params(shape, (3, 3, 3), patch_threshold=0.5)
    got_nodes, got_labels = component_labels(AffGraph(aff, pairs, device="cuda"), shape, "cuda",
                                             P, mws=False)
    want = {}
    for k, cc in enumerate(orc.connected_components(pairs, aff)):
        for n in cc:
            want[n] = k + 1
    got = {tuple(int(v) for v in n): int(l) for n, l in zip(got_nodes, got_labels)}
    assert got == want


def test_device_pairs_match_reference(golden, torch_cuda):
    """ppp_patch_pairs_count/_fill against the golden canonical pair list (and therefore, as
    a set, against the reference's cKDTree + filter output)."""
    from patchperpix_amd import backend
    g = golden
    if not g.has("selected_sorted") or len(g["selected_sorted"]) == 0:
        pytest.skip("no selected patches")
    P = backend.make_params(g.foreground.shape, g.patchshape, **g.kw)
    pts = _dev(torch_cuda, g["selected_sorted"].astype(np.int32))
    rows = backend.device_patch_pairs(pts, P, include_single=g.kw["includeSinglePatchCCS"])
    if int(g["early_out"]) == 3:
        assert rows is None
        return
    assert np.array_equal(rows.cpu().numpy().view(np.uint32), g["pairs"])


def test_synth_on_device_equals_numpy(torch_cuda):
    from patchperpix_amd import backend, synth
    torch = torch_cuda
    shape, ps = (9, 11, 13), (3, 5, 3)
    lab = synth.cell_labels(shape, [4, 5, 4], seed=3)
    want = synth.pred_from_labels(lab, ps, seed=7)
    P = backend.make_params(shape, ps, patch_threshold=0.5)
    got = backend.synth_pred(_dev(torch, lab.astype(np.int32)), P, seed=7, f16=True)
    assert np.array_equal(got.float().cpu().numpy(), want)


def test_synth_slab_equals_global_slice(torch_cuda):
    """A z-slab generated with voxel_offset carries the same values as the global volume."""
    from patchperpix_amd import backend, synth
    torch = torch_cuda
    shape, ps = (12, 9, 10), (3, 3, 3)
    lab = synth.cell_labels(shape, [4, 4, 4], seed=5)
    want = synth.pred_from_labels(lab, ps, seed=3)
    lo, hi = 4, 11
    Pl = backend.make_params((hi - lo, 9, 10), ps, patch_threshold=0.5)
    got = backend.synth_pred(_dev(torch, lab[lo:hi].astype(np.int32)), Pl, seed=3, f16=False,
                             voxel_offset=lo * 9 * 10).cpu().numpy()
    # interior slices only: at the slab's z-edges the neighbour labels are outside the slab
    assert np.array_equal(got[:, 1:-1], want[:, lo + 1:hi - 1])


def test_large_volume_consistency(torch_cuda):
    """96^3 / 7^3 (about 12 M pairs, every patch offset populated): the pair-per-lane and the
    workgroup-per-patch patch-graph kernels agree bit for bit, and the 3-slab tiled assembly
    equals the untiled one."""
    from patchperpix_amd import backend, synth, tiling
    from patchperpix_amd.flags import FLYLIGHT_NOTHIN_CC as FLYLIGHT
    from patchperpix_amd.vote_instances import vote_instances as vi
    torch = torch_cuda
    shape, ps = (96, 96, 96), (7, 7, 7)
    kw = dict(FLYLIGHT)
    P = backend.make_params(shape, ps, **kw)
    lab = synth.cell_labels(shape, [18, 18, 18], seed=0)
    labels = _dev(torch, lab.astype(np.int32))
    pred = backend.synth_pred(labels, P, seed=0, f16=True)
    fg = lab != 0
    res = vi.to_instance_seg(pred, fg.copy(), fg.copy(), fg.astype(np.uint8), ps,
                             **dict(kw, return_intermediates=True, _n_slabs=1))
    pairs, aff = res
    assert len(pairs) > 5_000_000
    ov = torch.zeros(shape, dtype=torch.uint8, device="cuda")
    cons = backend.consensus(pred, ov, P)
    vm, Pv = backend.cons_to_voxel_major(cons, P)
    del cons
    pd = _dev(torch, pairs.view(np.int32))
    aff_pa = backend.patch_graph_by_patch(pred, vm, pd, Pv).cpu().numpy()
    assert np.array_equal(aff_pa.view(np.uint32), aff.view(np.uint32))
    del vm, pd
    whole, _ = vi.to_instance_seg(pred, fg.copy(), fg.copy(), fg.astype(np.uint8), ps,
                                  **dict(kw, _n_slabs=1))
    tiled, _ = vi.to_instance_seg(pred, fg.copy(), fg.copy(), fg.astype(np.uint8), ps,
                                  **dict(kw, _n_slabs=3))
    assert whole.max() > 50
    assert np.array_equal(whole, tiled)


def test_tiled_equals_untiled_at_128_cubed_with_9_cubed_patches(torch_cuda):
    """128^3 / 9^3 with the shipped flags (thinning + mutex watershed, uint32 ids): the untiled
    assembly (41 GB of consensus), a 2 x 2 x 2 tile grid and a 3-slab one with the prediction behind
    a provider give the same instance map.  The sizes and flags of BASELINE config [2], an
    eighth of its edge."""
    from patchperpix_amd import backend, synth, tiling
    from patchperpix_amd.flags import FLYLIGHT
    from patchperpix_amd.vote_instances import vote_instances as vi
    torch = torch_cuda
    shape, ps = (128, 128, 128), (9, 9, 9)
    kw = dict(FLYLIGHT, _instances_dtype=np.uint32)
    P = backend.make_params(shape, ps, **kw)
    lab = synth.cell_labels(shape, [24, 24, 24], seed=0)
    pred = backend.synth_pred(_dev(torch, lab.astype(np.int32)), P, seed=0, f16=True)
    fg = lab != 0
    args = lambda: (fg.copy(), fg.copy(), fg.astype(np.uint8), ps)    # noqa: E731
    whole, _ = vi.to_instance_seg(pred, *args(), **dict(kw, _n_slabs=1))
    assert whole.dtype == np.uint32 and whole.max() > 1000
    tiled, _ = vi.to_instance_seg(pred, *args(), **dict(kw, _n_slabs=2, _yx_tiles=(2, 2)))
    assert np.array_equal(whole, tiled)

    class Provider:
        def pred_box(self, box):
            z0, z1, y0, y1, x0, x1 = box
            return pred[:, z0:z1, y0:y1, x0:x1].contiguous()
    fg_d = _dev(torch, fg.astype(np.uint8))
    flags = {k: v for k, v in kw.items()}
    prov, _ = tiling.assemble(Provider(), 0, shape, fg_d, fg_d.clone(), fg_d, ps, tiling.plan_slabs(shape[0], 3),
                              _yx_tiles=(1, 2), **flags)
    assert np.array_equal(whole, np.asarray(prov))


def _cover_inputs(torch, pred_host, foreground, numinst, ps, kw):
    """ranked list + mask exactly as to_instance_seg builds them."""
    from patchperpix_amd import backend
    from patchperpix_amd.vote_instances import ranked_patches as rp
    P = backend.make_params(pred_host.shape[1:], ps, **kw)
    pred = _dev(torch, pred_host.astype(np.float32))
    overlap = 1 * (numinst > 1)
    ov = _dev(torch, (overlap > 0).astype(np.uint8)) if P.use_overlap else None
    cons = backend.consensus(pred, ov, P)
    score = backend.rank_patches(pred, cons, ov, P)
    rad = [p // 2 for p in ps]
    radslice = tuple(slice(r, s - r) for r, s in zip(rad, pred_host.shape[1:]))
    fg = foreground.astype(bool)
    ranked = rp.rank_patches_by_score(None, score, foreground=fg, patchshape=ps)
    mask = fg.copy()
    mask[overlap > 0] = 0
    return pred, overlap, mask, ranked, radslice, rad, score


def test_device_cover_matches_reference_goldens(golden, torch_cuda, monkeypatch):
    from patchperpix_amd.vote_instances import foreground_cover as fc
    g = golden
    if not g.has("cover_coords"):
        pytest.skip("early-out case")
    pred, overlap, mask, ranked, radslice, rad, score = _cover_inputs(
        torch_cuda, g.pred, g.foreground, g.numinst, g.patchshape, g.kw)
    assert np.array_equal(ranked.coords, g["ranked_coords"])
    # (goldens with mark_close_neighboorhood / select_patches_overlap_neighborhood take the
    # sequential native loop in either mode, foreground_cover.py:53-85, 141-168)
    for mode in ("device", "host"):
        monkeypatch.setenv("PPP_COVER", mode)
        sel, n = fc.computeForegroundCover(overlap, mask.copy(), g.patchshape, ranked, radslice,
                                           pred, rad, None, score, silent=True, **g.kw)
        assert np.array_equal(sel.coords, g["cover_coords"]), mode


@pytest.mark.parametrize("variant", ["sparse", "passes", "score_threshold", "p7", "tall_window"])
def test_device_cover_matches_sequential_loop(variant, torch_cuda, monkeypatch):
    """Larger volumes and every rule of the loop: several pixel-threshold passes, overlap
    centres, the score threshold break and the stop rule.  `tall_window`: pz * py = 143 window
    rows -- more than the seven bits the 16-bit witness voxel of round 5 gave the row."""
    from patchperpix_amd import synth
    from patchperpix_amd.vote_instances import foreground_cover as fc
    from tests_flags import FLYLIGHT
    kw = dict(FLYLIGHT)
    shape, ps, skw = (40, 48, 56), (5, 5, 5), dict(seed=61, cell=[9, 10, 11], overlap_frac=0.02)
    if variant == "passes":
        kw["select_patches_for_sparse_data"] = False
    elif variant == "score_threshold":
        kw["score_threshold"] = 0.55
    elif variant == "p7":
        shape, ps, skw = (30, 44, 52), (7, 7, 7), dict(seed=62, cell=[12, 12, 12], noise=0.25)
    elif variant == "tall_window":
        shape, ps, skw = (36, 38, 20), (13, 11, 3), dict(seed=64, cell=[12, 11, 5], overlap_frac=0.02)
    c = synth.make_case(shape, ps, **skw)
    pred, overlap, mask, ranked, radslice, rad, _score = _cover_inputs(
        torch_cuda, c["pred"], c["foreground"], c["numinst"], ps, kw)
    out = {}
    for mode in ("device", "host"):
        monkeypatch.setenv("PPP_COVER", mode)
        sel, n = fc.computeForegroundCover(overlap, mask.copy(), ps, ranked, radslice,
                                           pred, rad, None, None, silent=True, **kw)
        out[mode] = sel.coords
    assert len(out["host"]) > 20
    assert np.array_equal(out["device"], out["host"])


@pytest.mark.gpu
@pytest.mark.parametrize("passes", [False, True])
def test_cover_with_a_bit_row_per_voxel_equals_rank_ordered_rows(passes, torch_cuda):
    """ppp_cover_pass_voxel_bits (the table the tiled assembly fills tile by tile under a
    prediction provider: a row per voxel, here starting a few voxels before the first centre)
    selects what ppp_cover_pass selects from the rows in rank order."""
    import torch
    from patchperpix_amd import backend, synth
    from patchperpix_amd.vote_instances import foreground_cover as fc
    from tests_flags import FLYLIGHT
    kw = dict(FLYLIGHT, select_patches_for_sparse_data=not passes)
    shape, ps = (24, 40, 44), (5, 5, 5)
    c = synth.make_case(shape, ps, seed=63, cell=[9, 10, 11], overlap_frac=0.02)
    pred, overlap, mask, ranked, radslice, rad, _score = _cover_inputs(
        torch_cuda, c["pred"], c["foreground"], c["numinst"], ps, kw)
    P = backend.params_from_kwargs(shape, ps, kw)
    dev = pred.device
    lin = torch.from_numpy(ranked.lin(shape)).to(dev)
    by_rank = backend.patch_bits(pred, torch.from_numpy(ranked.coords).to(dev), kw["fc_threshold"], P)
    first = max(0, int(lin.min()) - 5)
    V = int(np.prod(shape))
    by_voxel = torch.full((V - first, by_rank.shape[1]), -1, dtype=torch.int32, device=dev)   # junk off the list
    by_voxel[lin - first] = by_rank
    never = torch.from_numpy(fc.never_selected(overlap, ranked.lin(shape), ranked.scores,
                                               kw.get("score_threshold", False))).to(dev)
    m = torch.from_numpy((np.asarray(mask) != 0).astype(np.uint8)).to(dev)
    pix = fc._pix_thresholds(ps, kw)
    a, ra = fc.greedy_cover_device(m.clone(), by_rank, lin, never, pix, radslice, P)
    b, rb = fc.greedy_cover_device(m.clone(), by_voxel, lin, never, pix, radslice, P, bits_first_voxel=first)
    assert int(a.sum()) > 20 and ra == rb
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_patch_bits_dense_equals_per_centre(monkeypatch):
    """backend.patch_bits picks a per-voxel pass + row gather for many centres: same words as the
    wave-per-centre kernel, and as NumPy."""
    import torch
    from patchperpix_amd import backend, synth
    from tests_flags import FLYLIGHT
    shape, ps = (12, 14, 37), (5, 5, 5)
    c = synth.make_case(shape, ps, seed=71, cell=[6, 6, 6])
    P = backend.make_params(shape, ps, **dict(FLYLIGHT))
    pred = torch.from_numpy(c["pred"].astype(np.float16)).cuda()
    rng = np.random.default_rng(5)
    centres = np.stack([rng.integers(0, s, size=3000) for s in shape], axis=1).astype(np.int32)
    ct = torch.from_numpy(centres).cuda()
    dense = backend.patch_bits(pred, ct, 0.5, P).cpu().numpy().view(np.uint32)
    monkeypatch.setenv("PPP_PATCH_BITS", "sparse")
    sparse = backend.patch_bits(pred, ct, 0.5, P).cpu().numpy().view(np.uint32)
    assert np.array_equal(dense, sparse)
    vals = c["pred"].astype(np.float16).astype(np.float32)[:, centres[:, 0], centres[:, 1], centres[:, 2]].T > np.float32(0.5)
    want = np.zeros_like(dense)
    for r in range(vals.shape[1]):
        want[:, r // 32] |= vals[:, r].astype(np.uint32) << np.uint32(r % 32)
    assert np.array_equal(dense, want)


@pytest.mark.gpu
def test_patch_bits_more_centres_than_one_launch_holds(monkeypatch):
    """A HIP grid holds < 2^32 work-items: the wave-per-centre kernel (64 per centre) is launched
    in chunks of 2^24 centres.  (An unchunked launch silently ran part of the grid and left the
    rest of the table uninitialised -- found at 512^3 / 9^3, 1.2e8 cover candidates.)"""
    import torch
    from patchperpix_amd import backend, synth
    from tests_flags import FLYLIGHT
    shape, ps = (10, 12, 33), (3, 3, 3)
    c = synth.make_case(shape, ps, seed=72, cell=[5, 5, 5])
    P = backend.make_params(shape, ps, **dict(FLYLIGHT))
    pred = torch.from_numpy(c["pred"].astype(np.float16)).cuda()
    n = (1 << 24) + 4097
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    ct = torch.stack([torch.randint(0, s, (n,), device="cuda", generator=g, dtype=torch.int32)
                      for s in shape], 1).contiguous()
    dense = backend.patch_bits(pred, ct, 0.5, P)
    monkeypatch.setenv("PPP_PATCH_BITS", "sparse")
    sparse = backend.patch_bits(pred, ct, 0.5, P)
    assert torch.equal(dense, sparse)


@pytest.mark.gpu
def test_consensus_out_of_range_operands_take_exact_divisions(torch_cuda):
    """TH = 0.5: the float-only vote normalisation is checked for products up to 2^22; a tile with
    an operand above 1024 (never for probabilities) must fall back to the double divisions --
    still bit-identical to the oracle."""
    from oracle import ppp_oracle as orc
    from patchperpix_amd import synth
    from tests_flags import FLYLIGHT
    shape, ps = (12, 14, 70), (5, 5, 5)
    c = synth.make_case(shape, ps, seed=31, cell=[6, 6, 6])
    kw = dict(FLYLIGHT)
    pred = c["pred"].astype(np.float32)
    rng = np.random.default_rng(31)
    big = rng.random(pred.shape) < 0.002
    pred[big & (pred > 0.5)] *= 3000.0          # "foreground" values far outside [0, 1]
    pred[big & (pred < 0.5)] -= 2500.0          # and "background" ones (1 - v is large)
    ov = 1 * (c["numinst"] > 1)
    cons_ref = orc.consensus(pred, ov, ps, **kw)
    o = _stage_outputs(torch_cuda, pred, ov, ps, kw)
    assert np.array_equal(_bits(o["cons"]), _bits(orc.positive_planes(cons_ref, ps)))
    # an INFINITE value at a centre that votes nothing (outside the interior / not foreground): the
    # reference skips the centre (fillConsensusArray.cu:25-32); the exact path must select its
    # centre factor, not multiply by it (inf * 0 = nan would poison the accumulators)
    mid = int(np.prod(ps)) // 2
    others = [r for r in range(pred.shape[0]) if r != mid]
    pred[others[3], 0, 5, 10:30] = np.inf              # z = 0: never an interior centre
    pred[others[7], 6, 0, 20:50] = np.inf              # y = 0
    bgz, bgy, bgx = np.nonzero(pred[mid, 2:-2, 2:-2, 2:-2] < 0.5)
    if len(bgz):                                       # interior, but not foreground
        pred[others[11], bgz[0] + 2, bgy[0] + 2, bgx[0] + 2] = np.inf
    cons_ref = orc.consensus(pred, ov, ps, **kw)
    o = _stage_outputs(torch_cuda, pred, ov, ps, kw)
    assert not np.isnan(o["cons"]).any()
    assert np.array_equal(_bits(o["cons"]), _bits(orc.positive_planes(cons_ref, ps)))


@pytest.mark.gpu
@pytest.mark.parametrize("ps,shape,f16", [((9, 9, 9), (22, 24, 70), True), ((7, 7, 7), (18, 20, 90), False),
                                          ((5, 5, 5), (14, 15, 130), True)])
def test_clean_prediction_takes_the_short_classification(ps, shape, f16, torch_cuda, monkeypatch):
    """ppp_pred_check + ppp_params.pred_clean (round 6): on a prediction whose values all lie in [0, 1]
    and never equal the threshold, S1 classifies an operand as t = v - [v < 0.5] without the range
    check or the exact path -- the same consensus bits as the general kernel (PPP_S1_CLEAN=0), in the
    compact planes and in the voxel-major rows, and as the oracle.  A single value equal to 0.5, above
    1, negative zero or nan makes the check say so (bits 1 / 0) and the general kernel serve the call."""
    import ctypes
    from oracle import ppp_oracle as orc
    from patchperpix_amd import backend, synth
    from tests_flags import FLYLIGHT
    torch = torch_cuda
    kw = dict(FLYLIGHT)
    c = synth.make_case(shape, ps, seed=81, cell=[2 * p for p in ps], overlap_frac=0.02)
    pred_h = c["pred"].astype(np.float16 if f16 else np.float32)
    P = backend.make_params(shape, ps, **kw)
    ov = _dev(torch, (c["numinst"] > 1).astype(np.uint8))
    pred = _dev(torch, pred_h)
    assert backend.pred_check(pred, P) == 1
    want = orc.positive_planes(orc.consensus(pred_h.astype(np.float32), 1 * (c["numinst"] > 1), ps, **kw), ps)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PPP_S1_CLEAN", mode)
        backend.reload_env()
        Pm = P.copy()
        Pm.pred_clean = 1 if mode == "1" else 0
        cons = backend.consensus(pred, ov, Pm)
        vm, _ = backend.consensus_voxel_major(pred, ov, Pm)
        out[mode] = (cons.cpu().numpy(), vm.cpu().numpy())
        assert backend.NOTES.get("s1_kernel") == "consensus_v3_kernel"
    monkeypatch.delenv("PPP_S1_CLEAN")
    backend.reload_env()
    assert np.array_equal(_bits(out["1"][0]), _bits(want))
    assert np.array_equal(_bits(out["1"][0]), _bits(out["0"][0]))
    assert np.array_equal(_bits(out["1"][1]), _bits(out["0"][1]))
    # what makes a prediction unclean, each on its own; the consensus stays the oracle's
    mid = int(np.prod(ps)) // 2
    for value, bit in ((0.5, 2), (1.0009765625, 1), (-0.0, 1), (np.nan, 1)):
        bad_h = pred_h.copy()
        bad_h[(mid + 3) % bad_h.shape[0], shape[0] // 2, shape[1] // 2, shape[2] // 2] = value
        bad = _dev(torch, bad_h)
        assert backend.pred_check(bad, P) == 2 and backend.NOTES["pred_unclean_bits"] & bit, value
        if value == 0.5:
            got = backend.consensus(bad, ov, P).cpu().numpy()           # (decides by itself: general kernel)
            ref = orc.positive_planes(orc.consensus(bad_h.astype(np.float32), 1 * (c["numinst"] > 1), ps, **kw), ps)
            assert np.array_equal(_bits(got), _bits(ref))


@pytest.mark.gpu
def test_paint_more_nodes_than_one_launch_holds():
    """ppp_paint_instances runs a thread per (node, patch pixel) and therefore chunks its nodes
    below the 2^32 work-items a HIP grid holds (2^31 / C nodes per launch): a node list just
    over one chunk paints the same volume as the same nodes painted in two explicit calls."""
    import torch
    from patchperpix_amd import backend, synth
    from tests_flags import FLYLIGHT
    shape, ps = (14, 16, 40), (9, 9, 9)
    c = synth.make_case(shape, ps, seed=73, cell=[7, 8, 8])
    P = backend.make_params(shape, ps, **dict(FLYLIGHT))
    pred = torch.from_numpy(c["pred"].astype(np.float16)).cuda()
    per = ((1 << 31) // 729) & ~255
    n = per + 5003
    k = torch.arange(n, device="cuda", dtype=torch.int64)
    nodes = torch.stack([4 + k % 6, 4 + (k // 6) % 8, 4 + (k // 48) % 32], 1).to(torch.int32).contiguous()
    labels = (1 + k % 60000).to(torch.int32)
    labels[per:] = 60001 + torch.arange(n - per, device="cuda", dtype=torch.int32)   # the tail wins
    one = torch.zeros(shape, dtype=torch.int32, device="cuda")
    backend.paint_instances(pred, nodes, labels, one, P)
    two = torch.zeros(shape, dtype=torch.int32, device="cuda")
    backend.paint_instances(pred, nodes[:per].contiguous(), labels[:per].contiguous(), two, P)
    backend.paint_instances(pred, nodes[per:].contiguous(), labels[per:].contiguous(), two, P)
    assert torch.equal(one, two)
    assert int(one.max().item()) > 60000


@pytest.mark.gpu
@pytest.mark.parametrize("shape,ps,cell", [((40, 44, 48), (5, 5, 5), 9), ((36, 40, 44), (7, 7, 7), 12),
                                           ((1, 90, 100), (1, 9, 9), 14)])
def test_device_thinning_equals_host_loop(shape, ps, cell):
    """ppp_thin_cover (priority-parallel rounds on the device) keeps exactly the patches the
    sequential set-cover loop keeps (ppp_host_thin_cover, itself pinned to the reference by the
    thinning goldens): greedy cover first, then both thinnings of the same selected list --
    once on the full mask and once on a mask with holes the patches cannot cover."""
    import torch
    from patchperpix_amd import backend, synth
    from patchperpix_amd.vote_instances import foreground_cover as fc
    from tests_flags import FLYLIGHT
    c = synth.make_case(shape, ps, seed=81, cell=[min(cell, s) for s in shape], overlap_frac=0.02)
    kw = dict(FLYLIGHT)
    P = backend.make_params(shape, ps, **kw)
    pred = torch.from_numpy(c["pred"].astype(np.float16)).cuda()
    ov = torch.from_numpy((c["numinst"] > 1).astype(np.uint8)).cuda()
    cons = backend.consensus(pred, ov, P)
    score = backend.rank_patches(pred, cons, ov, P)
    lin, sc = backend.rank_order_device(score, c["foreground"], ps, to_host=False)
    Y, X = shape[1], shape[2]
    coords = torch.stack([lin // (Y * X), (lin // X) % Y, lin % X], 1).to(torch.int32)
    bits = backend.patch_bits(pred, coords, kw["fc_threshold"], P)
    mask0 = c["foreground"].copy()
    mask0[c["numinst"] > 1] = False
    rad = [p // 2 for p in ps]
    radslice = tuple(slice(r, s - r) for r, s in zip(rad, shape))
    mask_d = torch.from_numpy(mask0.astype(np.uint8)).cuda()
    never = torch.zeros(lin.shape, dtype=torch.bool, device="cuda")
    sel, _ = fc.greedy_cover_device(mask_d.clone(), bits, lin, never, [0], radslice, P)
    assert int(sel.sum()) > 30
    sel_lin, sel_bits = lin[sel].contiguous(), bits[sel].contiguous()
    for variant in ("full", "holes"):
        mask = mask0.copy()
        if variant == "holes":          # voxels no selected patch reaches: the loop's degenerate end
            mask[radslice][::7, ::5, ::3] = True
        want = backend.host_thin_cover(mask.astype(np.uint8), ps, sel_lin.cpu().numpy(),
                                       sel_bits.cpu().numpy().view(np.uint32))
        got = backend.thin_cover_device(torch.from_numpy(mask.astype(np.uint8)).cuda(), sel_bits,
                                        sel_lin, P).cpu().numpy()
        assert np.array_equal(got, want), variant
        assert 0 < want.sum() <= len(want)


@pytest.mark.gpu
def test_device_rank_order_equals_host(torch_cuda):
    """ppp_rank_order (rocPRIM select + stable descending radix sort inside the library) ==
    ppp_host_rank_order (C++ stable sort) == the oracle's rank_by_score, on scores with many
    ties, negative values and both zeros."""
    import torch
    from oracle import ppp_oracle as orc
    from patchperpix_amd import backend
    rng = np.random.default_rng(17)
    shape, ps = (20, 23, 70), (5, 5, 5)
    score = rng.integers(-3, 4, size=shape).astype(np.float32) * 0.25
    score[rng.random(shape) < 0.1] = -0.0
    fg = rng.random(shape) < 0.7
    lin_d, sc_d = backend.rank_order_device(torch.from_numpy(score).cuda(), fg, ps)
    lin_h = backend.host_rank_order(score, fg, ps)
    assert np.array_equal(lin_d, lin_h)
    assert np.array_equal(sc_d.view(np.uint32), score.reshape(-1)[lin_h].view(np.uint32))
    coords = orc.interior_fg_coords(fg, np.array([2, 2, 2]))
    ranked, _ = orc.rank_by_score(coords, score)
    want = (ranked[:, 0].astype(np.int64) * shape[1] + ranked[:, 1]) * shape[2] + ranked[:, 2]
    assert np.array_equal(lin_d, want)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c3d_p3_thin_mws", "c3d_p5_thin_mws", "c3d_p7_thin_mws"])
def test_device_mws_edges_and_loop_match_host_mws(name):
    """ppp_mws_edges (device: networkx edge order + stable |aff| sort) + ppp_host_mws_sorted give
    every node the label ppp_host_mws (pinned to the reference by the mws goldens) gives it --
    on the golden pair lists and on the same lists with tied |aff| values."""
    import torch
    from conftest import Golden
    from patchperpix_amd import backend
    g = Golden(name)
    pairs = np.ascontiguousarray(g["pairs"].astype(np.uint32))
    shape = g.foreground.shape
    P = backend.make_params(shape, g.patchshape, **dict(g.kw))
    nodes = np.unique(pairs.reshape(-1, 3), axis=0).astype(np.int32)
    rng = np.random.default_rng(3)
    for variant in ("golden", "ties"):
        aff = np.ascontiguousarray(g["aff"].astype(np.float32))
        if variant == "ties":       # few distinct magnitudes: the order is decided by the tie rules
            aff = (np.round(aff * 4) / 4).astype(np.float32) * rng.choice([1.0, -1.0], size=len(aff)).astype(np.float32)
        want_nodes, want_labels, want_n = backend.host_mws(pairs, aff, shape)
        lab, issued = backend.mws_labels_device(torch.from_numpy(pairs.view(np.int32)).cuda(),
                                                torch.from_numpy(aff).cuda(),
                                                torch.from_numpy(nodes).cuda(), P)
        lab = lab.cpu().numpy()
        got = {tuple(n): int(l) for n, l in zip(nodes[lab > 0], lab[lab > 0])}
        want = {tuple(n): int(l) for n, l in zip(want_nodes, want_labels)}
        assert got == want, variant
        assert issued == want_n


@pytest.mark.parametrize("ps,shape,cell", [((7, 7, 7), (40, 37, 45), 12), ((5, 5, 5), (33, 29, 70), 9),
                                           ((9, 9, 9), (30, 27, 33), 13), ((3, 3, 3), (19, 18, 40), 6)])
def test_rank_voxel_major_tiles_and_boxes(ps, shape, cell, torch_cuda):
    """ppp_rank_patches_vm == ppp_rank_patches (gather kernel, pinned to goldens / oracle) on
    volumes spanning several 8^3 centre tiles with ragged edges, with overlap voxels, for the
    whole volume and for a score box inside a consensus box that is a proper tile."""
    import torch
    from patchperpix_amd import backend, synth
    from tests_flags import FLYLIGHT
    c = synth.make_case(shape, ps, seed=91, cell=[cell] * 3, overlap_frac=0.03)
    kw = dict(FLYLIGHT)
    P = backend.make_params(shape, ps, **kw)
    pred = torch.from_numpy(c["pred"].astype(np.float16)).cuda()
    ov = torch.from_numpy((c["numinst"] > 1).astype(np.uint8)).cuda()
    cons = backend.consensus(pred, ov, P)
    want = backend.rank_patches(pred, cons, ov, P).cpu().numpy()
    vm, Pv = backend.cons_to_voxel_major(cons, P)
    got = backend.rank_patches(pred, vm, ov, Pv).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert (want > 0).sum() > 100
    del cons, vm
    # a tile: centres [z0, z1) x ..., consensus box = the tile grown by the radius
    r = ps[0] // 2
    sb = (r + 2, r + 1, r + 3, shape[0] - r - 3, shape[1] - r - 2, shape[2] - r - 5)
    box = (sb[0] - r, sb[1] - r, sb[2] - r, sb[3] + r, sb[4] + r, sb[5] + r)
    Pt = backend.make_params(shape, ps, cons_box=box, **kw)
    cons_t = backend.consensus(pred, ov, Pt)
    vm_t, Pvt = backend.cons_to_voxel_major(cons_t, Pt)
    got_t = backend.rank_patches(pred, vm_t, ov, Pvt, score_box=sb).cpu().numpy()
    sl = tuple(slice(sb[i], sb[i + 3]) for i in range(3))
    assert np.array_equal(got_t[sl].view(np.uint32), want[sl].view(np.uint32))


@pytest.mark.parametrize("ps,shape", [((7, 7, 7), (10, 12, 150)), ((5, 5, 5), (8, 11, 97)), ((9, 9, 9), (11, 12, 131)),
                                      ((1, 5, 5), (1, 30, 90))])
def test_consensus_flat_runs_equal_line_runs(ps, shape, torch_cuda, monkeypatch):
    """S1 with waves running over the flattened (y, x) order of the consensus box (a run may
    continue on the next line) == S1 with one run per line: whole volume and a sub-box whose
    lines start and end inside the volume; values and counts."""
    import torch
    from patchperpix_amd import backend, synth
    from tests_flags import FLYLIGHT
    c = synth.make_case(shape, ps, seed=95, cell=[6, 6, 9], overlap_frac=0.03)
    kw = dict(FLYLIGHT)
    pred = torch.from_numpy(c["pred"].astype(np.float16)).cuda()
    ov = torch.from_numpy((c["numinst"] > 1).astype(np.uint8)).cuda()
    boxes = [None, (min(1, shape[0] - 1), 2, 9, shape[0], shape[1] - 1, shape[2] - 7)]
    for box in boxes:
        P = backend.make_params(shape, ps, cons_box=box, **kw)
        out = {}
        for flat in ("0", "1"):
            monkeypatch.setenv("PPP_S1_FLAT", flat)
            backend.reload_env()
            cons, cnt = backend.consensus(pred, ov, P, want_count=True)
            out[flat] = (cons.cpu().numpy(), cnt.cpu().numpy())
        assert np.array_equal(out["0"][0].view(np.uint32), out["1"][0].view(np.uint32))
        assert np.array_equal(out["0"][1], out["1"][1])
        assert np.count_nonzero(out["0"][0]) > 1000


@pytest.mark.parametrize("ps,shape", [((7, 7, 7), (11, 12, 150)), ((5, 5, 5), (9, 11, 97)), ((9, 9, 9), (12, 13, 131)),
                                      ((3, 3, 3), (7, 9, 70)), ((1, 5, 5), (1, 30, 90)), ((7, 7, 7), (16, 20, 40))])
@pytest.mark.parametrize("dtype", ["f16", "f32"])
def test_consensus_two_slice_packed_kernel_equals_v2(ps, shape, dtype, torch_cuda, monkeypatch):
    """S1 v3 (two z-slices per lane, packed f32 chain, 16-bit count codes, masks applied per
    centre / per accumulator) == S1 v2, values and counts, bit for bit: whole volume (odd and even
    numbers of slices), a sub-box, flattened and per-line runs."""
    import torch
    from patchperpix_amd import backend, synth
    from tests_flags import FLYLIGHT
    c = synth.make_case(shape, ps, seed=96, cell=[6, 6, 9], overlap_frac=0.03)
    kw = dict(FLYLIGHT)
    pred = torch.from_numpy(c["pred"].astype(np.float16 if dtype == "f16" else np.float32)).cuda()
    ov = torch.from_numpy((c["numinst"] > 1).astype(np.uint8)).cuda()
    boxes = [None, (min(1, shape[0] - 1), 2, 9, shape[0], shape[1] - 1, shape[2] - 7)]
    for box in boxes:
        P = backend.make_params(shape, ps, cons_box=box, **kw)
        for flat in ("0", "1"):
            monkeypatch.setenv("PPP_S1_FLAT", flat)
            backend.reload_env()
            out = {}
            for v3 in ("0", "1"):
                monkeypatch.setenv("PPP_S1_V3", v3)
                backend.reload_env()
                cons, cnt = backend.consensus(pred, ov, P, want_count=True)
                out[v3] = (cons.cpu().numpy(), cnt.cpu().numpy())
            assert np.array_equal(out["0"][1], out["1"][1]), (box, flat, "counts")
            assert np.array_equal(out["0"][0].view(np.uint32), out["1"][0].view(np.uint32)), (box, flat)
            assert np.count_nonzero(out["0"][0]) > 1000


@pytest.mark.parametrize("ps,shape", [((7, 7, 7), (11, 12, 150)), ((5, 5, 5), (9, 11, 97)), ((9, 9, 9), (12, 13, 131)),
                                      ((5, 9, 9), (9, 19, 40))])
@pytest.mark.parametrize("dtype", ["f16", "f32"])
def test_consensus_two_wave_kernel_equals_one_wave(ps, shape, dtype, torch_cuda, monkeypatch):
    """S1 v4 (PPP_S1_V4=1: the accumulators of a run split over the two waves of a workgroup that
    share one pair of operand images) == S1 v3, bit for bit: compact values and counts, and the
    voxel-major rows it writes itself; whole volume and a sub-box, flattened and per-line runs."""
    import torch
    from patchperpix_amd import backend, synth
    from tests_flags import FLYLIGHT
    c = synth.make_case(shape, ps, seed=98, cell=[6, 6, 9], overlap_frac=0.03)
    kw = dict(FLYLIGHT)
    pred = torch.from_numpy(c["pred"].astype(np.float16 if dtype == "f16" else np.float32)).cuda()
    ov = torch.from_numpy((c["numinst"] > 1).astype(np.uint8)).cuda()
    boxes = [None, (min(1, shape[0] - 1), 2, 9, shape[0], shape[1] - 1, shape[2] - 7)]
    for box in boxes:
        P = backend.make_params(shape, ps, cons_box=box, **kw)
        for flat in ("0", "1"):
            monkeypatch.setenv("PPP_S1_FLAT", flat)
            out = {}
            for v4 in ("0", "1"):
                monkeypatch.setenv("PPP_S1_V4", v4)
                backend.reload_env()
                cons, cnt = backend.consensus(pred, ov, P, want_count=True)
                name = backend.lib().ppp_consensus_kernel_name().decode()
                if v4 == "1" and name == "consensus_v3_kernel":
                    monkeypatch.delenv("PPP_S1_V4")
                    backend.reload_env()
                    pytest.skip("the two-wave kernel is an experiment: built with PPP_BUILD_EXPERIMENTS=1 only")
                assert name == ("consensus_v4_kernel" if v4 == "1" else "consensus_v3_kernel")
                rows, _ = backend.consensus_voxel_major(pred, ov, P)
                out[v4] = (cons.cpu().numpy(), cnt.cpu().numpy(), rows)
            assert np.array_equal(out["0"][1], out["1"][1]), (box, flat, "counts")
            assert np.array_equal(out["0"][0].view(np.uint32), out["1"][0].view(np.uint32)), (box, flat)
            assert torch.equal(out["0"][2].view(torch.int32), out["1"][2].view(torch.int32)), (box, flat, "rows")
            assert np.count_nonzero(out["0"][0]) > 1000


@pytest.mark.parametrize("ps,shape", [((7, 7, 7), (11, 12, 150)), ((5, 5, 5), (9, 11, 97)), ((9, 9, 9), (12, 13, 131)),
                                      ((1, 5, 5), (1, 30, 90)), ((7, 7, 7), (16, 20, 40))])
def test_consensus_written_voxel_major_directly(ps, shape, torch_cuda, monkeypatch):
    """S1 writing the symmetric voxel-major rows itself (positive entry, mirrored entry, zero fill
    of the entries without a source voxel in the box) == compact planes + ppp_cons_to_voxel_major,
    bit for bit: whole volume and a sub-box, flattened and per-line runs."""
    import torch
    from patchperpix_amd import backend, synth
    from tests_flags import FLYLIGHT
    c = synth.make_case(shape, ps, seed=97, cell=[6, 6, 9], overlap_frac=0.03)
    kw = dict(FLYLIGHT)
    pred = torch.from_numpy(c["pred"].astype(np.float16)).cuda()
    ov = torch.from_numpy((c["numinst"] > 1).astype(np.uint8)).cuda()
    boxes = [None, (min(1, shape[0] - 1), 2, 9, shape[0], shape[1] - 1, shape[2] - 7)]
    for box in boxes:
        P = backend.make_params(shape, ps, cons_box=box, **kw)
        assert backend.lib().ppp_consensus_writes_voxel_major(P) == 1
        for flat in ("0", "1"):
            monkeypatch.setenv("PPP_S1_FLAT", flat)
            backend.reload_env()
            monkeypatch.setenv("PPP_S1_DIRECT_VM", "0")
            want, _ = backend.consensus_voxel_major(pred, ov, P)
            monkeypatch.setenv("PPP_S1_DIRECT_VM", "1")
            got, Pv = backend.consensus_voxel_major(pred, ov, P)
            assert Pv.cons_layout == backend.CONS_VOXEL_MAJOR and got.shape == want.shape
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (box, flat)
            assert int(torch.count_nonzero(want)) > 1000
            del got, want


@pytest.mark.parametrize("rule", ["flylight", "probprod", "count", "th07"])
def test_consensus_wide_patch_kernel_equals_generic(rule, torch_cuda, monkeypatch):
    """S1 for 25-wide 2-d patches (accumulator windows, rolled staging; ppp_consensus_v2.hip)
    == the generic gather kernel: values and counts, whole image and a sub-box, for the value /
    threshold rules the wide kernel instantiates."""
    import torch
    from patchperpix_amd import backend, synth
    from tests_flags import FLYLIGHT
    shape, ps = (1, 70, 150), (1, 25, 25)
    c = synth.make_case(shape, ps, seed=99, cell=[1, 30, 30], overlap_frac=0.03)
    kw = dict(FLYLIGHT)
    if rule == "probprod":
        kw.update(consensus_norm_prob_product=False, consensus_prob_product=True)
    elif rule == "count":
        kw.update(consensus_norm_prob_product=False)
    elif rule == "th07":
        kw.update(patch_threshold=0.7)
    pred = torch.from_numpy(c["pred"].astype(np.float16)).cuda()
    ov = torch.from_numpy((c["numinst"] > 1).astype(np.uint8)).cuda()
    for box in (None, (0, 3, 9, 1, shape[1] - 2, shape[2] - 7)):
        P = backend.make_params(shape, ps, cons_box=box, **kw)
        out = {}
        for wide in ("0", "1"):
            monkeypatch.setenv("PPP_S1_WIDE", wide)
            backend.reload_env()
            cons, cnt = backend.consensus(pred, ov, P, want_count=True)
            out[wide] = (cons.cpu().numpy(), cnt.cpu().numpy(), backend.NOTES.get("s1_kernel"))
        assert out["0"][2] == "consensus_gather_kernel" and out["1"][2] == "consensus_wide_kernel"
        assert np.array_equal(out["0"][1], out["1"][1]), (box, "counts")
        assert np.array_equal(out["0"][0].view(np.uint32), out["1"][0].view(np.uint32)), box
        assert np.count_nonzero(out["0"][0]) > 1000


@pytest.mark.parametrize("name", ["s96_p9", "s64_p7", "f140_p7", "w696x520_p25"])
def test_benchmark_scale_against_the_oracle(name, torch_cuda):
    """The HIP path against the ORACLE at a benchmark-like size: bench.py's generator and the
    shipped flylight flags (thinning + mutex watershed, uint32 ids) at 96^3 / 9^3 (BASELINE config
    [2]'s patch) and 64^3 / 7^3 (config [1]'s) -- and, since round 6, AT THE STATED SIZES of
    BASELINE configs [1] and [0]: `f140_p7` = the 140^3 / 7^3 volume of bench.py's flylight140_p7
    (26 minutes of 8 cores for the oracle: 96 270 cover patches, 23 946 after thinning, 2.34 M pair
    rows, 516 instances) and `w696x520_p25` = one 696 x 520 image with 25 x 25 patches, kernel
    semantics (bench.py's worm2d_p25; 159 s: 13 905 cover patches, 18 111 pair rows, 215 instances).  The expected values come from
    tests/golden/scale_<name>.npz = oracle/ppp_oracle_scale.to_instance_seg run by
    tests/golden/gen_scale_fixture.py (a quarter of an hour of 8 cores for 96^3; the scale forms
    of the oracle's host stages are held equal to the literal ones by tests/test_oracle_scale.py).
    Scores and pair affinities bit-exact (sha256 of the float32 arrays), selected patches, pair
    rows and the instance map identical -- untiled AND cut into 2 x 2 x 2 tiles."""
    import hashlib
    import json
    import zlib
    from conftest import GOLDEN_DIR
    from patchperpix_amd import backend, synth
    from patchperpix_amd.vote_instances import vote_instances as vi
    torch = torch_cuda
    z = np.load(os.path.join(GOLDEN_DIR, "scale_%s.npz" % name))
    shape, ps, cell = tuple(int(v) for v in z["shape"]), [int(v) for v in z["patchshape"]], [int(v) for v in z["cell"]]
    kw = json.loads(str(z["flags"]))
    kw["_instances_dtype"] = np.uint32
    P = backend.make_params(shape, ps, **kw)
    lab = synth.cell_labels(shape, cell, seed=int(z["seed"]))
    pred = backend.synth_pred(_dev(torch, lab.astype(np.int32)), P, seed=int(z["seed"]), f16=True)
    assert zlib.crc32(pred.cpu().numpy().tobytes()) == int(z["pred_f16_crc32"])      # same input
    fg = lab != 0
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()    # noqa: E731
    args = lambda: (fg.copy(), fg.copy(), fg.astype(np.uint8), ps)                   # noqa: E731
    # float stages
    ov = torch.zeros(shape, dtype=torch.uint8, device="cuda")
    cons = backend.consensus(pred, ov, P)
    score = backend.rank_patches(pred, cons, ov, P).cpu().numpy()
    del cons
    assert sha(score.astype(np.float32)) == str(z["scores_sha256"])
    pairs, aff = vi.to_instance_seg(pred, *args(), **dict(kw, return_intermediates=True, _n_slabs=1))
    assert len(pairs) == int(z["n_pairs"])
    assert sha(np.ascontiguousarray(pairs, dtype=np.uint32)) == str(z["pairs_sha256"])
    assert sha(np.ascontiguousarray(aff, dtype=np.float32)) == str(z["aff_sha256"])
    # the instance map: untiled, and 2 x 2 x 2 tiles
    want = z["instances"]
    whole, _ = vi.to_instance_seg(pred, *args(), **dict(kw, _n_slabs=1))
    assert whole.dtype == np.uint32 and np.array_equal(whole, want)
    tiled, _ = vi.to_instance_seg(pred, *args(), **dict(kw, _n_slabs=2 if shape[0] > 1 else 1, _yx_tiles=(2, 2)))
    assert np.array_equal(tiled, want)
    if name == "f140_p7":
        # two z-slabs, the plan the memory rule takes by itself, and the selected patches of both stages
        two, _ = vi.to_instance_seg(pred, *args(), **dict(kw, _n_slabs=2))
        auto, _ = vi.to_instance_seg(pred, *args(), **kw)
        assert np.array_equal(two, want) and np.array_equal(auto, want)
    assert len(np.unique(want)) - 1 > 20


@pytest.mark.parametrize("ps,shape,cell", [((7, 7, 7), (48, 52, 56), 18), ((9, 9, 9), (44, 48, 60), 24), ((5, 5, 5), (40, 44, 48), 12)])
def test_open_rows_never_reach_a_result(ps, shape, cell, torch_cuda, monkeypatch):
    """Tiled path: S1 leaves the voxel-major entries without a source voxel in the consensus box
    unwritten (ppp_consensus_rows, PPP_VM_OPEN=1).  With the pool poisoned with NaN before every
    tile's consensus the pair affinities (bit patterns) and the instance map equal those of fully
    zeroed rows (PPP_VM_OPEN=0): no open entry is ever read for an active lane."""
    from patchperpix_amd import backend, synth
    from patchperpix_amd.flags import FLYLIGHT
    from patchperpix_amd.vote_instances import vote_instances as vi
    torch = torch_cuda
    kw = dict(FLYLIGHT, _instances_dtype=np.uint32)
    P = backend.make_params(shape, ps, **kw)
    lab = synth.cell_labels(shape, [cell] * 3, seed=5)
    pred = backend.synth_pred(_dev(torch, lab.astype(np.int32)), P, seed=5, f16=True)
    fg = lab != 0
    args = lambda: (fg.copy(), fg.copy(), fg.astype(np.uint8), list(ps))     # noqa: E731
    grid = dict(_n_slabs=2, _yx_tiles=(2, 2))
    monkeypatch.setenv("PPP_VM_OPEN", "0")
    want = vi.to_instance_seg(pred, *args(), **dict(kw, **grid))[0]
    want_pairs, want_aff = vi.to_instance_seg(pred, *args(), **dict(kw, return_intermediates=True, **grid))
    monkeypatch.setenv("PPP_VM_OPEN", "1")
    monkeypatch.setenv("PPP_VM_POISON", "1")
    got = vi.to_instance_seg(pred, *args(), **dict(kw, **grid))[0]
    got_pairs, got_aff = vi.to_instance_seg(pred, *args(), **dict(kw, return_intermediates=True, **grid))
    assert np.array_equal(want_pairs, got_pairs)
    assert not np.isnan(got_aff).any()
    assert np.array_equal(_bits(want_aff), _bits(got_aff))
    assert np.array_equal(want, got) and want.max() > 5


@pytest.mark.parametrize("ps,shape,cell,flags", [((7, 7, 7), (48, 52, 56), 18, "shipped"), ((9, 9, 9), (44, 48, 60), 24, "shipped"),
                                                 ((5, 5, 5), (40, 44, 48), 12, "cc"), ((5, 9, 9), (30, 50, 70), 20, "cc")])
def test_consensus_cache_equals_recomputation(ps, shape, cell, flags, torch_cuda, monkeypatch):
    """Tiled path with the consensus cache (`_cons_cache`: ppp_consensus_part fills COMPACT planes
    over the whole block once, ppp_cons_planes_to_rows cuts every tile's rows from them, in both
    passes) against the tiled path that computes every tile's rows itself: pair rows, pair affinities
    (bit patterns) and instance map are the same; the row buffer is poisoned with NaN before every
    cut, so an entry the transpose leaves unwritten would show."""
    from patchperpix_amd import backend, synth
    from patchperpix_amd import flags as F
    from patchperpix_amd.vote_instances import vote_instances as vi
    torch = torch_cuda
    kw = dict(F.FLYLIGHT if flags == "shipped" else F.FLYLIGHT_CC, _instances_dtype=np.uint32)
    P = backend.make_params(shape, ps, **kw)
    lab = synth.cell_labels(shape, [cell] * 3, seed=6)
    pred = backend.synth_pred(_dev(torch, lab.astype(np.int32)), P, seed=6, f16=True)
    fg = lab != 0
    args = lambda: (fg.copy(), fg.copy(), fg.astype(np.uint8), list(ps))     # noqa: E731
    for grid in (dict(_n_slabs=2, _yx_tiles=(2, 2)), dict(_n_slabs=3, _yx_tiles=(1, 2))):
        want = vi.to_instance_seg(pred, *args(), **dict(kw, _cons_cache=False, **grid))[0]
        want_pairs, want_aff = vi.to_instance_seg(pred, *args(), **dict(kw, return_intermediates=True, _cons_cache=False, **grid))
        monkeypatch.setenv("PPP_VM_POISON", "1")
        backend.NOTES.pop("cons_cache_gb", None)
        got = vi.to_instance_seg(pred, *args(), **dict(kw, _cons_cache=True, **grid))[0]
        # (a shape without a voxel-major ranking kernel keeps the two-pass path: same result)
        assert ("cons_cache_gb" in backend.NOTES) == (ps[0] == ps[1] == ps[2])
        got_pairs, got_aff = vi.to_instance_seg(pred, *args(), **dict(kw, return_intermediates=True, _cons_cache=True, **grid))
        monkeypatch.delenv("PPP_VM_POISON")
        assert np.array_equal(want_pairs, got_pairs)
        assert not np.isnan(got_aff).any()
        assert np.array_equal(_bits(want_aff), _bits(got_aff))
        assert np.array_equal(want, got) and want.max() > 5


# (the last case, found by tools/fuzz_tiling.py: axes so narrow that the columns of tiles grow to the same
# clipped box -- the sweep mistook the second column for the continuation of the first)
@pytest.mark.parametrize("ps,shape,cell,flags", [((9, 9, 9), (70, 48, 60), 24, "shipped"), ((7, 7, 7), (64, 52, 56), 18, "shipped"),
                                                 ((5, 5, 5), (50, 44, 48), 12, "cc"), ((7, 7, 7), (46, 18, 17), 9, "shipped")])
def test_ring_sweep_equals_plain_tiles(ps, shape, cell, flags, torch_cuda, monkeypatch):
    """Tiled path with the rows in a ring (`_ring_z`: the tiles of a column bottom-up, every base
    slice computed once per pass by ppp_consensus_part, ranking and patch-graph kernels addressing
    the ring) against the plain tiled path: pair rows, pair affinities (bit patterns), instance map.
    The ring is poisoned with NaN once, so a row entry nobody wrote would show."""
    from patchperpix_amd import backend, synth
    from patchperpix_amd import flags as F
    from patchperpix_amd.vote_instances import vote_instances as vi
    torch = torch_cuda
    kw = dict(F.FLYLIGHT if flags == "shipped" else F.FLYLIGHT_CC, _instances_dtype=np.uint32, _cons_cache=False)
    P = backend.make_params(shape, ps, **kw)
    lab = synth.cell_labels(shape, [cell] * 3, seed=7)
    pred = backend.synth_pred(_dev(torch, lab.astype(np.int32)), P, seed=7, f16=True)
    fg = lab != 0
    args = lambda: (fg.copy(), fg.copy(), fg.astype(np.uint8), list(ps))     # noqa: E731
    thick = max(8, ps[0] - 1)
    n = -(-shape[0] // thick)
    for grid in (dict(_n_slabs=n, _yx_tiles=(2, 2)), dict(_n_slabs=n, _yx_tiles=(1, 1))):
        want = vi.to_instance_seg(pred, *args(), **dict(kw, **grid))[0]
        want_pairs, want_aff = vi.to_instance_seg(pred, *args(), **dict(kw, return_intermediates=True, **grid))
        ring = -(-shape[0] // n) + 28
        monkeypatch.setenv("PPP_VM_POISON", "1")
        backend.NOTES.pop("ring_z", None)
        got = vi.to_instance_seg(pred, *args(), **dict(kw, _ring_z=ring, **grid))[0]
        assert backend.NOTES.get("ring_z") == ring
        # (one ranking launch serves two or three consecutive tiles of a column: the scores pass keeps
        # its rows on a smaller box than the pool is sized for, so its ring is longer)
        assert backend.NOTES.get("rank_group", 1) > 1 and backend.NOTES.get("ring_z_scores", 0) >= ring
        got_pairs, got_aff = vi.to_instance_seg(pred, *args(), **dict(kw, return_intermediates=True, _ring_z=ring, **grid))
        monkeypatch.delenv("PPP_VM_POISON")
        assert np.array_equal(want_pairs, got_pairs)
        assert not np.isnan(got_aff).any()
        assert np.array_equal(_bits(want_aff), _bits(got_aff))
        assert np.array_equal(want, got) and want.max() > 5


def test_ring_request_falls_back_where_no_kernel_reads_a_ring(torch_cuda):
    """3^3 patches are ranked by the one-wave kernel, which reads plain boxes only (ppp_rank_workspace_bytes
    with ring_z set answers 0): a plan that asks for a ring sweeps plain tiles instead of failing in the
    first ranking launch (found by tools/fuzz_tiling.py)."""
    from patchperpix_amd import backend, synth, tiling
    from patchperpix_amd import flags as F
    from patchperpix_amd.vote_instances import vote_instances as vi
    torch = torch_cuda
    ps, shape = (3, 3, 3), (40, 30, 28)
    kw = dict(F.FLYLIGHT, _instances_dtype=np.uint32, _cons_cache=False)
    P = backend.make_params(shape, ps, **kw)
    lab = synth.cell_labels(shape, [8] * 3, seed=9)
    pred = backend.synth_pred(_dev(torch, lab.astype(np.int32)), P, seed=9, f16=True)
    fg = lab != 0
    args = lambda: (fg.copy(), fg.copy(), fg.astype(np.uint8), list(ps))     # noqa: E731
    grid = dict(_n_slabs=4, _yx_tiles=(1, 2))
    want = vi.to_instance_seg(pred, *args(), **dict(kw, **grid))[0]
    backend.NOTES.pop("ring_z", None)
    got = vi.to_instance_seg(pred, *args(), **dict(kw, _ring_z=10 + tiling.ring_margin(3) + 2, **grid))[0]
    assert "ring_z" not in backend.NOTES
    assert np.array_equal(want, got) and want.max() > 5
    Pq = backend.make_params(shape, (5, 5, 5), **kw)
    Pq.ring_z = 24
    assert backend.rank_vm_available(Pq)          # 5^3: the workgroup-per-tile kernel reads a ring


@pytest.mark.parametrize("ps,shape", [((9, 9, 9), (20, 30, 44)), ((7, 7, 7), (18, 26, 40)), ((5, 5, 5), (14, 20, 36))])
def test_rank_one_bit_masks_equal_two_bit(ps, shape, torch_cuda, monkeypatch):
    """S2 with ONE mask bit per partner (PPP_RANK_P1=1; P' = v > TH; N = not P': the rows already hold 0
    for invalid partners; measured slower, kept as a second implementation) == the two-bit masks, bit for bit -- also when some values EQUAL the threshold (neither
    P nor N: the launch then takes the two-bit kernel by itself) and with an overlap mask."""
    from patchperpix_amd import backend, synth
    from tests_flags import FLYLIGHT
    torch = torch_cuda
    c = synth.make_case(shape, ps, seed=77, cell=[7, 9, 11], overlap_frac=0.03)
    kw = dict(FLYLIGHT)
    base = c["pred"].astype(np.float16)
    with_eq = base.copy()
    rng = np.random.default_rng(5)
    idx = rng.integers(0, with_eq.size, size=200)
    with_eq.reshape(-1)[idx] = np.float16(0.5)                       # values that are neither P nor N
    ov = torch.from_numpy((c["numinst"] > 1).astype(np.uint8)).cuda()
    P = backend.make_params(shape, ps, **kw)
    for name, arr in (("plain", base), ("with values == TH", with_eq)):
        pred = torch.from_numpy(arr).cuda()
        rows, Pv = backend.consensus_voxel_major(pred, ov, P)
        out = {}
        for p1 in ("1", "0"):
            monkeypatch.setenv("PPP_RANK_P1", p1)
            backend.reload_env()
            out[p1] = backend.rank_patches(pred, rows, ov, Pv).cpu().numpy()
        assert np.array_equal(_bits(out["1"]), _bits(out["0"])), name
        assert np.count_nonzero(out["1"] > 0) > 100
        # ... and both equal the gather kernel on the compact planes
        cons = backend.consensus(pred, ov, P)
        want = backend.rank_patches(pred, cons, ov, P).cpu().numpy()
        assert np.array_equal(_bits(out["1"]), _bits(want)), name


@pytest.mark.parametrize("ps,shape", [((9, 9, 9), (28, 44, 76)), ((7, 7, 7), (26, 40, 70))])
def test_rank_tile_order_does_not_change_scores(ps, shape, torch_cuda, monkeypatch):
    """S2 deals the tiles of a launch heaviest first inside every XCD's range (weight = the chunks of
    a tile's valid rows; PPP_RANK_ORDER=c: its active centres; =0: spatial order): which workgroup
    takes which tile when must not show in the scores."""
    from patchperpix_amd import backend, synth
    from tests_flags import FLYLIGHT
    torch = torch_cuda
    c = synth.make_case(shape, ps, seed=31, cell=[8, 10, 12], overlap_frac=0.02)
    pred = torch.from_numpy(c["pred"].astype(np.float16)).cuda()
    ov = torch.from_numpy((c["numinst"] > 1).astype(np.uint8)).cuda()
    P = backend.make_params(shape, ps, **dict(FLYLIGHT))
    rows, Pv = backend.consensus_voxel_major(pred, ov, P)
    out = {}
    for mode in ("1", "c", "0"):
        monkeypatch.setenv("PPP_RANK_ORDER", mode)
        backend.reload_env()
        out[mode] = backend.rank_patches(pred, rows, ov, Pv).cpu().numpy()
    monkeypatch.delenv("PPP_RANK_ORDER")
    backend.reload_env()
    assert np.count_nonzero(out["0"] > 0) > 100
    assert np.array_equal(_bits(out["1"]), _bits(out["0"])) and np.array_equal(_bits(out["c"]), _bits(out["0"]))


def test_consensus_part_and_planes_to_rows(torch_cuda):
    """ppp_consensus_part: COMPACT planes / open VOXEL_MAJOR rows of a box filled in pieces equal the
    one-launch result; ppp_cons_planes_to_rows from a larger planes box equals the rows S1 writes for
    the sub-box (wherever both define an entry: sources inside the sub-box)."""
    from patchperpix_amd import backend, synth
    from tests_flags import FLYLIGHT
    torch = torch_cuda
    for ps, shape in (((9, 9, 9), (21, 30, 140)), ((7, 7, 7), (20, 33, 90)), ((5, 5, 5), (16, 20, 70))):
        c = synth.make_case(shape, ps, seed=99, cell=[7, 8, 9], overlap_frac=0.02)
        kw = dict(FLYLIGHT)
        pred = torch.from_numpy(c["pred"].astype(np.float16)).cuda()
        ov = torch.from_numpy((c["numinst"] > 1).astype(np.uint8)).cuda()
        box = (1, 2, 3, shape[0] - 1, shape[1] - 2, shape[2] - 5)
        P = backend.make_params(shape, ps, cons_box=box, **kw)
        whole = backend.consensus(pred, ov, P)
        # the same planes from four pieces (cut in z at an odd slice, in y and in x)
        zc, yc, xc = box[0] + 5, box[1] + 11, box[2] + (box[5] - box[2]) // 2 + 3
        parts = [(box[0], box[1], box[2], zc, box[4], box[5]), (zc, box[1], box[2], box[3], yc, box[5]),
                 (zc, yc, box[2], box[3], box[4], xc), (zc, yc, xc, box[3], box[4], box[5])]
        pieces = torch.full_like(whole, float("nan"))
        for part in parts:
            backend.consensus_part(pred, ov, P, part, pieces)
        assert torch.equal(pieces.view(torch.int32), whole.view(torch.int32)), ps
        # rows of a sub-box cut from the planes == rows S1 writes for that sub-box, where the source
        # voxel of an entry lies inside the sub-box (elsewhere S1's open rows are undefined)
        sub = (box[0] + 2, box[1] + 3, box[2] + 9, box[3] - 1, box[4] - 2, box[5] - 4)
        Ps = backend.make_params(shape, ps, cons_box=sub, **kw)
        rows_cut, Pv = backend.cons_planes_to_rows(whole, box, Ps)
        Pz = Ps.copy()
        Pz.cons_layout = backend.CONS_VOXEL_MAJOR
        rows_s1 = backend.consensus(pred, ov, Pz)                    # zero-filled rows (not open)
        W = rows_s1.shape[-1]
        Lc = (W - 1) // 2
        assert torch.equal(rows_cut[..., Lc:].view(torch.int32), rows_s1[..., Lc:].view(torch.int32)), ps
        # negative entries: equal wherever S1's are not the zero fill of a missing source
        inner = (slice(ps[0] - 1, None), slice(ps[1] - 1, rows_s1.shape[1] - (ps[1] - 1)),
                 slice(ps[2] - 1, rows_s1.shape[2] - (ps[2] - 1)))
        assert torch.equal(rows_cut[inner].view(torch.int32), rows_s1[inner].view(torch.int32)), ps
        assert int(torch.count_nonzero(rows_cut[inner])) > 1000


@pytest.mark.parametrize("name", ["c2d_p5_mark", "c3d_p3_mark_nosparse", "c3d_p3_near_overlap"])
def test_marked_cover_options_tiled(name, torch_cuda):
    """The two optional branches of the greedy cover (foreground_cover.py:53-85, 141-168) through the
    tiled assembly (y/x tiles, z-slabs where the case has slices): the reference's instances."""
    from conftest import Golden
    from patchperpix_amd.vote_instances import vote_instances as vi
    g = Golden(name)
    kw = dict(g.kw, debug=False, isbiHack=False, save_no_intermediates=True, sample=1.0,
              result_folder="/tmp", affinities="x.zarr")
    for grid in (dict(_n_slabs=1), dict(_n_slabs=2 if g.foreground.shape[0] > 1 else 1, _yx_tiles=(2, 2))):
        inst, fg = vi.to_instance_seg(g.pred.copy(), g.foreground.copy(), g.foreground.copy(), g.numinst.copy(),
                                      g.patchshape, **dict(kw, **grid))
        assert np.array_equal(inst, g["instances"]) and inst.any(), grid
        assert np.array_equal(fg, g["foreground_out"])


def test_resume_from_a_saved_consensus(torch_cuda, tmp_path):
    """Kernel path: `save_consensus` writes the reference-layout array (consensus_array.py:202-206);
    a later call with `consensus=<that file>` (:213-218) loads it instead of running S1 -- same
    planes bit for bit, same instances; a stored ranking (`ranked_patches`, ranked_patches.py:
    137-139) is taken as is."""
    import pickle
    from conftest import Golden
    from patchperpix_amd import backend
    from patchperpix_amd.vote_instances import consensus_array as ca
    from patchperpix_amd.vote_instances import vote_instances as vi
    torch = torch_cuda
    g = Golden("c3d_p5_cells")
    kw = dict(g.kw, debug=False, isbiHack=False, save_no_intermediates=True, sample=1.0,
              result_folder=str(tmp_path), affinities="vol.zarr")
    args = lambda: (g.pred.copy(), g.foreground.copy(), g.foreground.copy(), g.numinst.copy(), g.patchshape)   # noqa: E731
    assert vi.to_instance_seg(*args(), **dict(kw, save_consensus=True)) == (None, None)
    path = str(tmp_path / "vol_consensus.npy")
    assert os.path.exists(path)
    pred = _dev(torch, g.pred)
    P = backend.make_params(g.pred.shape[1:], g.patchshape, **g.kw)
    ov = _dev(torch, (g.overlap_mask > 0).astype(np.uint8)) if P.use_overlap else None
    want = backend.consensus(pred, ov, P).cpu().numpy()
    got, _, _ = ca.loadOrComputeConsensus(None, g.patchshape, None, None, pred, None, None, None, g.overlap_mask,
                                          **dict(kw, consensus=path))
    assert np.array_equal(_bits(got.cpu().numpy()), _bits(want))
    inst, _ = vi.to_instance_seg(*args(), **dict(kw, consensus=path))
    assert np.array_equal(inst, g["instances"])
    # a stored ranking in the reference's pickle format
    ranking = str(tmp_path / "ranking.pickle")
    with open(ranking, "wb") as f:
        pickle.dump([(np.array(c), float(s)) for c, s in zip(g["ranked_coords"], g["ranked_scores"])], f)
    inst2, _ = vi.to_instance_seg(*args(), **dict(kw, consensus=path, ranked_patches=ranking))
    assert np.array_equal(inst2, g["instances"])
