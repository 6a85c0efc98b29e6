"""CPU: the C-ABI library loads and exports every symbol include/ppp_mi355x.h declares, the
compute entry points refuse to run without a GPU (no CPU fallback), and the native HOST
stages (sort, cover, thinning, pair enumeration, mutex watershed) reproduce the reference's
golden vectors.  No device compute happens here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO
from patchperpix_amd import backend
from patchperpix_amd.vote_instances import graph_mws

HEADER = os.path.join(REPO, "include", "ppp_mi355x.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ppp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(backend.library_path())
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), "libppp_mi355x.so does not export %s" % n
    assert L.ppp_abi_version() == backend.ABI_VERSION
    # and the Python binding covers exactly the same set
    assert sorted(backend._SIGNATURES) == names


def test_params_struct_matches_header_size():
    # int32 x7, pad, double x2, int32 x8, box 6 x int32  (natural C alignment)
    # (+ origin x3, ring_z, pred_clean, rank_tile)
    assert ctypes.sizeof(backend.Params) == 8 * 4 + 2 * 8 + 8 * 4 + 6 * 4 + 4 * 4 + 2 * 4
    P = backend.make_params((4, 5, 6), (3, 3, 3), patch_threshold=0.9)
    assert (P.bg_rule, P.thi) == (backend.BG_INV_TH, 1.0 - 0.9)
    P = backend.make_params((4, 5, 6), (3, 3, 3), patch_threshold=0.4)
    assert (P.bg_rule, P.thi) == (backend.BG_LESS_THAN_TH, 0.4)   # silent switch, :391-398
    with pytest.raises(RuntimeError, match="how is bg defined"):
        backend.make_params((4, 5, 6), (3, 3, 3), patch_threshold=0.9, vi_bg_use_inv_th=False)
    assert backend.lib().ppp_cons_planes(ctypes.byref(P)) == (5 * 5 * 5 - 1) // 2


def test_compute_entry_points_fail_loudly_without_gpu():
    if backend.device_count() > 0:
        pytest.skip("a GPU is present")
    P = backend.make_params((4, 5, 6), (3, 3, 3), patch_threshold=0.5)
    buf = np.zeros(8, dtype=np.float32)
    p = buf.ctypes.data_as(ctypes.c_void_p)
    rc = backend.lib().ppp_consensus(p, 0, None, p, None, ctypes.byref(P), None)
    assert rc == -2  # PPP_ERR_NO_DEVICE
    assert b"no CPU path" in backend.lib().ppp_last_error()
    with pytest.raises(RuntimeError):
        backend.check(rc)


def test_bad_arguments_are_rejected():
    P = backend.make_params((4, 5, 6), (3, 3, 3), patch_threshold=0.5)
    P.px = 4
    assert backend.lib().ppp_consensus(None, 0, None, None, None, ctypes.byref(P), None) == -1
    assert b"odd" in backend.lib().ppp_last_error()


def _bits(pred, coords, thresh):
    C = pred.shape[0]
    words = (C + 31) // 32
    vals = pred[(slice(None),) + tuple(coords.T)].T > np.float32(thresh)     # [n, C]
    out = np.zeros((len(coords), words), dtype=np.uint32)
    for r in range(C):
        out[:, r // 32] |= vals[:, r].astype(np.uint32) << np.uint32(r % 32)
    return out


def test_host_stages_match_reference(golden):
    g = golden
    if int(g["early_out"]) in (1, 2):
        pytest.skip("early-out case")
    ps = g.patchshape
    rad = [p // 2 for p in ps]
    shape = g.foreground.shape
    # ranking order (stable sort, score descending)
    lin = backend.host_rank_order(g["scores"], g.foreground, ps)
    coords = np.stack(np.unravel_index(lin, shape), axis=1)
    assert np.array_equal(coords, g["ranked_coords"])
    # cover
    mask = g.foreground.copy()
    mask[g.overlap_mask > 0] = 0
    running, _owner = backend.padded_mask(mask)
    radslice = tuple(slice(rad[i], shape[i] - rad[i]) for i in range(3))
    remaining = int(np.count_nonzero(running[radslice]))
    bits = _bits(g.pred, coords, g.kw["fc_threshold"])
    selected = np.zeros(len(lin), dtype=np.uint8)
    if g.kw["select_patches_for_sparse_data"]:
        pix_ths = [0]
    else:
        pix_ths = [t for t in [500, 100, 50, 10, 0] if t < int(np.prod(ps) / 2)]
    # (mark_close_neighboorhood: one mark volume shared by the passes)
    marked = np.zeros(shape, dtype=np.uint8) if g.kw.get("mark_close_neighboorhood") else None
    ov8 = (g.overlap_mask > 0).astype(np.uint8)
    for t in pix_ths:
        remaining, _ = backend.host_cover_pass(
            running, ov8, ps, lin,
            np.ascontiguousarray(g["ranked_scores"]), bits, t, None, selected, remaining, marked=marked)
        if remaining < 1:
            break
    cover = coords[selected.astype(bool)]
    cover_lin, cover_bits = lin[selected.astype(bool)], bits[selected.astype(bool)]
    if g.kw.get("select_patches_overlap_neighborhood"):
        # foreground_cover.py:53-85: a second cover of the ring around the overlap voxels
        from scipy import ndimage
        ring = ~ndimage.binary_dilation(g.overlap_mask, iterations=2) & \
            ndimage.binary_dilation(g.overlap_mask, iterations=5) & mask
        keep = ~selected.astype(bool) & ring.reshape(-1)[lin]
        run2, _owner2 = backend.padded_mask(ring)
        sel2 = np.zeros(int(keep.sum()), dtype=np.uint8)
        backend.host_cover_pass(run2, ov8, ps, np.ascontiguousarray(lin[keep]),
                                np.ascontiguousarray(g["ranked_scores"][keep]), np.ascontiguousarray(bits[keep]),
                                t, None, sel2, int(np.count_nonzero(run2[radslice])), marked=marked)
        chosen = np.zeros(int(np.prod(shape)), dtype=bool)
        chosen[cover_lin] = True
        chosen[lin[keep][sel2.astype(bool)]] = True
        cover_lin = np.flatnonzero(chosen)                             # raster order (np.argwhere)
        cover = np.stack(np.unravel_index(cover_lin, shape), axis=1)
        cover_bits = _bits(g.pred, cover, g.kw["fc_threshold"])
    assert np.array_equal(cover, g["cover_coords"])
    sel = cover
    if g.has("thin_coords"):
        keep = backend.host_thin_cover(mask.astype(np.uint8), ps, np.ascontiguousarray(cover_lin),
                                       np.ascontiguousarray(cover_bits))
        sel = cover[keep]
        assert np.array_equal(sel, g["thin_coords"])
    # pairs
    sorted_zyx, pairs = backend.host_patch_pairs(
        sel, ps, include_single=g.kw["includeSinglePatchCCS"])
    assert np.array_equal(sorted_zyx, g["selected_sorted"])
    if int(g["early_out"]) == 3:
        assert pairs is None
        return
    assert np.array_equal(pairs, g["pairs"])


def test_mutex_watershed_matches_reference(golden):
    g = golden
    if not g.kw.get("mws") or int(g["early_out"]) != 0:
        pytest.skip("not an mws case")
    from oracle import ppp_oracle as orc
    ccs = graph_mws.mws_from_pairs(g["pairs"], g["aff"])
    inst = orc.paint_instances(ccs, g.pred, g.patchshape, g.foreground.shape,
                               g.kw["patch_threshold"])
    assert np.array_equal(inst, g["instances"])


def test_size_limits_are_reported():
    """Volumes of 2^31 voxels or more and > 160 KB of patch bits are refused with a message,
    not mis-addressed."""
    P = backend.make_params((2048, 1024, 1024), (3, 3, 3), patch_threshold=0.5)
    rc = backend.lib().ppp_cons_to_reference(ctypes.c_void_p(8), ctypes.c_void_p(8),
                                             ctypes.byref(P), None)
    assert rc == -4 and b"2^31" in backend.lib().ppp_last_error()


def _labels_from_ccs(ccs):
    """node -> label exactly as graph_to_labeling paints the reference's ccs list."""
    out = {}
    for k, cc in enumerate(ccs):
        for n in cc:
            out[tuple(int(v) for v in n)] = k + 1
    return out


@pytest.mark.parametrize("seed", range(12))
def test_native_mws_equals_python_restatement(seed):
    """ppp_host_mws against graph_mws.mws_from_pairs (itself pinned to the reference's golden,
    tests/test_oracle_golden.py): same label for every node, same number of issued labels, on
    random graphs with tied weights, self loops, zero and repeated rows."""
    from patchperpix_amd import backend
    from patchperpix_amd.vote_instances.graph_mws import mws_from_pairs
    rng = np.random.default_rng(seed)
    shape = (6, 7, 8)
    n_nodes = int(rng.integers(5, 60))
    lin = rng.choice(int(np.prod(shape)), size=n_nodes, replace=False)
    coords = np.stack(np.unravel_index(lin, shape), axis=1).astype(np.uint32)
    n_rows = int(rng.integers(n_nodes, 6 * n_nodes))
    a = rng.integers(0, n_nodes, size=n_rows)
    b = rng.integers(0, n_nodes, size=n_rows)
    if seed % 3 == 0:
        b[: n_rows // 8] = a[: n_rows // 8]          # self loops
    pairs = np.concatenate([coords[a], coords[b]], axis=1)
    # few distinct magnitudes -> many ties; some exact zeros; both signs
    aff = (rng.integers(-6, 7, size=n_rows) / 8.0).astype(np.float32)
    if seed % 2:
        aff = (aff * rng.uniform(0.5, 1.0, size=n_rows)).astype(np.float32)
    ref = _labels_from_ccs(mws_from_pairs(pairs, aff))
    n_ref = len(mws_from_pairs(pairs, aff))
    nodes, labels, n_labels = backend.host_mws(pairs, aff, shape)
    got = {tuple(int(v) for v in n): int(l) for n, l in zip(nodes, labels)}
    assert got == ref
    assert n_labels == n_ref


def test_native_mws_empty_and_all_repulsive():
    from patchperpix_amd import backend
    nodes, labels, n = backend.host_mws(np.zeros((0, 6), np.uint32), np.zeros((0,), np.float32), (4, 4, 4))
    assert len(nodes) == 0 and n == 0
    pairs = np.array([[0, 0, 0, 0, 0, 1], [0, 0, 1, 0, 0, 2]], dtype=np.uint32)
    nodes, labels, n = backend.host_mws(pairs, np.array([-0.5, -0.25], np.float32), (4, 4, 4))
    assert len(nodes) == 0 and n == 0


def test_th05_quotient(tmp_path):
    """The S1 kernel's float-only normalisation for TH = 0.5 equals the reference's double
    arithmetic for every float product in [0.25, 2^22] (exhaustive, ~1 s of C)."""
    import subprocess
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "th05_quotient.c")
    exe = str(tmp_path / "th05_quotient")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, src, "-lm"])
    out = subprocess.check_output([exe]).decode().split()
    assert int(out[0]) == 201326593 and int(out[1]) == 0 and int(out[2]) == 0


def test_blockwise_io_helpers():
    """load_input / verify_shape / clean_mask / replace / merge_dicts (the helpers the reference
    package re-exports, vote_instances/__init__.py:2): checked against straightforward NumPy
    constructions of what they must return."""
    from scipy import ndimage
    from patchperpix_amd.vote_instances import stitch_patch_graph as spg
    from patchperpix_amd.vote_instances.vote_instances import merge_dicts, replace
    rng = np.random.default_rng(0)

    class IO:
        def __init__(self, a, channel_order):
            self.a, self.shape, self.channel_order, self.keys = a, a.shape, channel_order, ["k"]

        def read(self, bb, key):
            return np.array(self.a[bb])

    vol = rng.random((3, 20, 22, 24))
    big = np.pad(vol, ((0, 0), (3, 3), (3, 3), (3, 3)))          # margin = context + overlap = 3
    for off in ([0, 0, 0], [8, 8, 8], [16, 16, 16], [12, 16, 20]):
        data, margin = spg.load_input(IO(vol, [slice(0, 3)]), "k", off, [1, 1, 1], [2, 2, 2], [8, 8, 8])
        want = big[(slice(None),) + tuple(slice(o, o + 14) for o in off)]
        want = want[(slice(None),) + tuple(slice(0, min(14, 3 + s - o + 3)) for o, s in zip(off, vol.shape[1:]))]
        assert data.shape[0] == 3 and np.array_equal(data[:, :want.shape[1], :want.shape[2], :want.shape[3]], want)
        assert list(margin) == [0 if o - 3 < 0 else 3 for o in off]
    data, _ = spg.load_input(IO(vol[0], None), "k", [8, 8, 8], [0, 0, 0], [2, 2, 2], [8, 8, 8], padding=False)
    assert np.array_equal(data, vol[0][6:18, 6:18, 6:18])
    out = rng.random((12, 12, 12))                               # 8^3 block with an overlap of 2
    res, box = spg.verify_shape([1, 8, 8, 8], out, (20, 22, 24), (8, 8, 8))
    assert res.shape == (1, 12, 12, 12) and box == (slice(1, 2), slice(6, 18), slice(6, 18), slice(6, 18))
    res, box = spg.verify_shape([0, 0, 0, 0], out, (20, 22, 24), (8, 8, 8))
    assert res.shape == (1, 10, 10, 10) and box[1:] == (slice(0, 10),) * 3 and np.array_equal(res[0], out[2:, 2:, 2:])
    mask = rng.random((12, 13, 14)) > 0.7
    lab, n = ndimage.label(mask, np.ones((3, 3, 3)))
    sizes = np.bincount(lab.ravel())
    want = np.isin(lab, [i for i in range(1, n + 1) if sizes[i] > 4])
    assert np.array_equal(spg.clean_mask(mask, np.ones((3, 3, 3)), 4), want)
    arr = rng.integers(0, 9, size=(5, 6))
    got = replace(arr, np.array([1, 3]), np.array([0, 7]))
    assert np.array_equal(got, np.where(arr == 1, 0, np.where(arr == 3, 7, arr)))
    a = {"a": 1, "b": {"c": 2, "d": {"e": 3}}, "f": [1]}
    assert merge_dicts(a, {"b": {"c": 5, "d": {"g": 1}}, "f": {"x": 1}, "h": 2}) == \
        {"a": 1, "b": {"c": 5, "d": {"e": 3, "g": 1}}, "f": {"x": 1}, "h": 2}


def test_post_steps():
    """remove_small_components / relabel / dilate_instances against literal loops over the labels
    (PatchPerPix/util/postprocess.py:24-52, stitch_patch_graph.py:871-880)."""
    from scipy import ndimage
    from patchperpix_amd import postprocess as pp
    rng = np.random.default_rng(3)
    a = rng.integers(0, 12, size=(9, 10, 11)).astype(np.uint16) * (rng.random((9, 10, 11)) < 0.3)
    a = a.astype(np.uint16)
    want = a.copy()
    for lab in np.unique(a):
        if np.count_nonzero(a == lab) <= 20:
            want[a == lab] = 0
    got = pp.remove_small_components(a, 20)
    assert np.array_equal(got, want)
    rl = pp.relabel(got)
    labs = [l for l in np.unique(got) if l != 0]
    want_rl = np.zeros_like(got)
    for i, l in enumerate(labs):
        want_rl[got == l] = i + 1
    assert np.array_equal(rl, want_rl)
    assert np.array_equal(pp.relabel(got, start=5), np.where(want_rl > 0, want_rl + 4, 0))
    d = pp.dilate_instances(rl)
    ref = rl.copy()
    for l in np.unique(rl):
        if l:
            ref[ndimage.binary_dilation(ref == l, iterations=1)] = l
    assert np.array_equal(d, ref)


@pytest.mark.parametrize("seed", range(6))
def test_native_mws_on_dense_noisy_graphs(seed):
    """Hundreds of nodes, thousands of edges, half of them repulsive, many tied weights: the regime
    in which clusters carry long mutex neighbour lists and merge often (what the hash set of root
    pairs in ppp_host_mws.cpp is for) -- against the oracle's union-find restatement and, on the
    smaller ones, the literal one."""
    from oracle import ppp_oracle as orc
    from oracle import ppp_oracle_scale as ors
    from patchperpix_amd import backend
    rng = np.random.default_rng(1000 + seed)
    shape = (12, 14, 16)
    n_nodes = int(rng.integers(150, 500))
    lin = rng.choice(int(np.prod(shape)), size=n_nodes, replace=False)
    coords = np.stack(np.unravel_index(lin, shape), axis=1).astype(np.uint32)
    n_rows = int(rng.integers(6 * n_nodes, 14 * n_nodes))
    a = rng.integers(0, n_nodes, size=n_rows)
    b = rng.integers(0, n_nodes, size=n_rows)
    pairs = np.concatenate([coords[a], coords[b]], axis=1)
    mag = rng.integers(1, 40, size=n_rows) / 64.0
    sign = np.where(rng.random(n_rows) < (0.5 if seed % 2 else 0.25), -1.0, 1.0)
    aff = (mag * sign).astype(np.float32)
    nodes, labels, n_labels = backend.host_mws(pairs, aff, shape)
    got = {int(np.ravel_multi_index(tuple(int(v) for v in n), shape)): int(l) for n, l in zip(nodes, labels) if l}
    nodes_o, labels_o, n_ids = ors.mutex_watershed(pairs, aff, shape)
    assert n_labels == n_ids
    assert got == {int(n): int(l) for n, l in zip(nodes_o, labels_o) if l}
    if n_nodes < 260:
        ccs = orc.mutex_watershed(pairs, aff)
        lit = {}
        for k, cc in enumerate(ccs):
            for n in cc:
                lit[int(np.ravel_multi_index(n, shape))] = k + 1
        assert got == lit and n_labels == len(ccs)


def test_lcg_mask_words_formula_equals_the_library():
    """backend.lcg_words (array arithmetic on the device) = ppp_patch_graph_lcg_words: per pair the
    masks [intersection pixels of A][intersection planes of B][64-bit chunks of a plane]."""
    import ctypes
    for ps in [(3, 3, 3), (5, 5, 5), (7, 7, 7), (9, 9, 9), (1, 5, 5), (1, 25, 25), (3, 5, 5)]:
        P = backend.make_params((40, 60, 60), ps, patch_threshold=0.5)
        rng = np.random.default_rng(sum(ps))
        d = rng.integers(-2 * max(ps), 2 * max(ps) + 1, size=(200, 3))
        d[0] = 0
        got = backend.lcg_words(d[:, 0], d[:, 1], d[:, 2], P)
        want = [int(backend.lib().ppp_patch_graph_lcg_words(int(a), int(b), int(c), ctypes.byref(P)))
                for a, b, c in d]
        nch = -(-ps[2] // (64 // ps[2]))
        if ps[2] <= 9:
            assert got.tolist() == want
            assert want[0] == ps[0] * ps[1] * ps[2] * ps[0] * nch
        else:       # the 25-wide 2-d kernel runs the generator itself: no masks
            assert not any(want)


@pytest.mark.parametrize("budget_words,max_batches", [(1 << 40, 64), (6000, 64), (6000, 3), (900, 64)])
def test_lcg_mask_plan_cuts_the_groups_into_batches_that_fit(budget_words, max_batches, monkeypatch):
    """backend._lcg_plan: every batch's masks fit the budget, the offsets of a batch's rows tile its
    part of the buffer without overlap, only rows with intersecting windows are served, rows of one
    group never straddle two batches."""
    import torch
    ps = (5, 5, 5)
    P = backend.make_params((40, 60, 60), ps, patch_threshold=0.5)
    rng = np.random.default_rng(5)
    counts = rng.integers(1, 40, size=300)
    n = int(counts.sum())
    d = rng.integers(-2 * (ps[0] - 1), 2 * (ps[0] - 1) + 1, size=(n, 3))
    dkey = ((d[:, 0] + 2 * ps[0]) * (4 * ps[1] + 1) + (d[:, 1] + 2 * ps[1])) * (4 * ps[2] + 1) + d[:, 2] + 2 * ps[2]
    group_start = np.concatenate([[0], np.cumsum(counts)])
    words = backend.lcg_words(d[:, 0], d[:, 1], d[:, 2], P)
    monkeypatch.setenv("PPP_PA_LCG_BYTES", str(8 * budget_words))
    monkeypatch.setenv("PPP_PA_LCG_BATCHES", str(max_batches))
    plan = backend._lcg_plan(torch.from_numpy(dkey), torch.from_numpy(group_start), P)
    if budget_words <= int(max(np.add.reduceat(words, group_start[:-1]))):
        assert plan is None          # a single group does not fit: the kernel keeps the generator
        return
    cuts, pos_cuts = plan["group_cuts"], plan["pos_cuts"]
    assert cuts[0] == 0 and cuts[-1] == len(counts) and len(cuts) == len(pos_cuts)
    assert all(a <= b for a, b in zip(cuts[:-1], cuts[1:]))
    off, pos = plan["drop_off"].numpy(), plan["pos"].numpy()
    served = off >= 0
    assert np.all(words[served] > 0)
    n_served_batches = min(len(cuts) - 1, max_batches)
    seen = np.zeros(n, bool)
    for b in range(len(cuts) - 1):
        r0, r1 = group_start[cuts[b]], group_start[cuts[b + 1]]
        rows = pos[pos_cuts[b]:pos_cuts[b + 1]]
        assert np.all((rows >= r0) & (rows < r1))
        seen[rows] = True
        if b >= n_served_batches:
            assert len(rows) == 0 and not served[r0:r1].any()
            continue
        # all intersecting rows of the batch are served, back to back in row order
        want = np.nonzero(words[r0:r1] > 0)[0] + r0
        assert sorted(rows.tolist()) == want.tolist()
        starts, sizes = off[want], words[want]
        assert starts[0] == 0 and np.array_equal(starts[1:], np.cumsum(sizes)[:-1])
        assert int(starts[-1] + sizes[-1]) <= min(budget_words, plan["buffer_words"])
        # inside the batch: sorted by offset code
        assert np.all(np.diff(dkey[rows]) >= 0)
    assert np.array_equal(seen, served)
    if max_batches >= len(cuts) - 1:
        assert np.array_equal(served, words > 0)


def test_kernel_build_options_mirror_the_references_decisions():
    """setKernelBuildOptions (utilVoteInstances.py:389-449) as the list of -D switches the reference would compile
    with, derived from the SAME decisions the kernels run with (backend.make_params): precedence of the background
    rules with the defaults of absent keys (inv_th is on unless switched off), the th < 0.5 switch, the value rule,
    and -- found by tests/golden/fuzz_build_options_vs_reference.py, 6 000 random flag sets against the reference's
    own function -- the 'counted votes are not normalised' assertion belongs to the consensus step alone."""
    from patchperpix_amd.vote_instances import utilVoteInstances as util
    assert util.setKernelBuildOptions(step="consensus", patch_threshold=0.9) == ["-DUSE_INV_TH", "-DNORM_PROB_PRODUCT"]
    assert util.setKernelBuildOptions(step="consensus", patch_threshold=0.3, overlapping_inst=True,
                                      consensus_norm_prob_product=False) == ["-DUSE_LESS_THAN_TH", "-DOVERLAP", "-DPROB_PRODUCT"]
    assert util.setKernelBuildOptions(step="rank", patch_threshold=0.5, vi_bg_use_inv_th=False, vi_bg_use_half_th=True,
                                      rank_int_counter=True) == ["-DUSE_HALF_TH", "-DNORM_PATCH_RANK", "-DCOUNT_POS_NEG"]
    assert util.setKernelBuildOptions(step="patch_graph", patch_threshold=0.5) == ["-DNORM_PATCH_AFFINITY"]
    counted = dict(patch_threshold=0.5, consensus_norm_prob_product=False, consensus_prob_product=False, consensus_norm_aff=True)
    with pytest.raises(AssertionError):
        util.setKernelBuildOptions(step="consensus", **counted)
    assert util.setKernelBuildOptions(step="rank", **counted) == ["-DUSE_INV_TH", "-DNORM_PATCH_RANK"]
    with pytest.raises(RuntimeError):
        util.setKernelBuildOptions(patch_threshold=0.5, vi_bg_use_inv_th=False)


def test_host_stage_fuzzer_runs_clean_on_a_few_trials():
    """tests/fuzz_host_stages.py (native host stages against the oracle's Python on random tiny cases; 1 650 clean
    trials in round 6) as a smoke test: twenty trials."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "fuzz_host_stages.py"), "--trials", "20", "--seed", "11"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "20 trials, 0 failures" in out, out[-2000:]
