"""Steps 3 and 4: greedy foreground cover and set-cover thinning
(reference: PatchPerPix/vote_instances/foreground_cover.py).

The greedy cover runs on the device as an exact priority-parallel algorithm
(ppp_cover_count / _ready / _select, see csrc/ppp_cover.hip); ``PPP_COVER=host`` selects the
sequential native host loop (ppp_host_cover_pass) instead, which is also what the thinning
step uses.  Both work on bit masks ``pred[:, c] > fc_threshold`` that the device packs
(ppp_patch_bits), so the (C,Z,Y,X) prediction never leaves the GPU.
"""
import logging
import os

import numpy as np

from .. import backend
from .ranked_patches import PatchList

logger = logging.getLogger(__name__)

COVER_CHUNK = 1 << 20  # ranked patches per device->host bit transfer


def _bits_for(pred_affs, coords, thresh, P):
    import torch
    c = torch.from_numpy(np.ascontiguousarray(coords, dtype=np.int32)).to(pred_affs.device)
    return backend.patch_bits(pred_affs, c, thresh, P).cpu().numpy().view(np.uint32)


def computeForegroundCover(overlap_mask, mask_to_cover, patchshape, ranked_patches_list,
                           radslice, pred_affs, rad, debug_output1, scores_array,
                           silent=False, **kwargs):
    """foreground_cover.py:15-126.  Returns (selected PatchList in rank order, count)."""
    ranked = PatchList.from_any(ranked_patches_list)
    P = backend.params_from_kwargs(pred_affs.shape[1:], patchshape, kwargs)
    # `mark_close_neighboorhood` (:141-143, 162-168) and `select_patches_overlap_neighborhood`
    # (:53-85) make a patch's fate depend on marks left by earlier selections anywhere in its
    # slice, not only on overlapping windows: they take the sequential native loop
    mark = bool(kwargs.get("mark_close_neighboorhood", False))
    near_overlap = bool(kwargs.get("select_patches_overlap_neighborhood", False))
    # the device form packs a window row into one 32-bit word; wider patches (none of the
    # reference's configurations) take the sequential native loop
    if os.environ.get("PPP_COVER", "device") != "host" and int(patchshape[2]) <= 32 and not mark \
            and not near_overlap:
        return _cover_on_device(overlap_mask, mask_to_cover, patchshape, ranked, radslice,
                                pred_affs, P, silent, **kwargs)
    return cover_sequential(overlap_mask, mask_to_cover, patchshape, ranked, radslice,
                            lambda coords: _bits_for(pred_affs, coords, kwargs["fc_threshold"], P),
                            scores_array, silent=silent, **kwargs)


def cover_sequential(overlap_mask, mask_to_cover, patchshape, ranked, radslice, bits_of, scores_array,
                     silent=True, **kwargs):
    """The sequential native cover (ppp_host_cover_pass[_marked]) with the reference's two optional
    branches, `mark_close_neighboorhood` (foreground_cover.py:141-143, 162-168) and
    `select_patches_overlap_neighborhood` (:53-85).  bits_of(coords [m, 3]) -> uint32 [m, words]:
    the patch bits of a chunk of centres (from the prediction, wherever it lives: the stage path
    packs them from the resident block, the tiled assembly from its frames).  Host arrays in,
    (selected PatchList, count) out."""
    mark = bool(kwargs.get("mark_close_neighboorhood", False))
    near_overlap = bool(kwargs.get("select_patches_overlap_neighborhood", False))
    running, _owner = backend.padded_mask(mask_to_cover)
    overlap = np.ascontiguousarray(np.asarray(overlap_mask) > 0).astype(np.uint8)
    selected = np.zeros(len(ranked), dtype=np.uint8)
    marked = np.zeros(running.shape, dtype=np.uint8) if mark else None
    pix_ths = _pix_thresholds(patchshape, kwargs)
    thr = kwargs.get("score_threshold", False)
    thr = thr if isinstance(thr, float) else None
    lin = ranked.lin(running.shape)
    remaining = int(np.count_nonzero(running[radslice]))
    for pix_th in pix_ths:
        if not silent:
            logger.info("compute foreground cover, threshold %s", pix_th)
        # every pass restarts at rank 0 (the reference passes rpidx by value)
        for s in range(0, len(ranked), COVER_CHUNK):
            if remaining <= 0:
                break
            e = min(len(ranked), s + COVER_CHUNK)
            bits = bits_of(ranked.coords[s:e])
            remaining, stopped = backend.host_cover_pass(
                running, overlap, patchshape, lin[s:e], ranked.scores[s:e], bits, pix_th, thr,
                selected[s:e], remaining, marked=marked)
            if stopped:
                break  # the pass hit the score threshold
        if remaining < 1:
            break
    if near_overlap:
        return _select_near_overlap(overlap_mask, mask_to_cover, patchshape, ranked, selected, radslice,
                                    bits_of, pix_th, thr, marked, scores_array, **kwargs)
    sel = ranked[np.flatnonzero(selected)]
    if len(sel) and not silent:
        logger.info("num patches to cover foreground: %s best score: %s, worst score: %s, "
                    "uncovered: %s", len(sel), sel.scores[0], sel.scores[-1], remaining)
    return sel, len(sel)


def _select_near_overlap(overlap_mask, mask_to_cover, patchshape, ranked, selected, radslice, bits_of,
                         pix_th, thr, marked, scores_array, **kwargs):
    """foreground_cover.py:53-85 (`select_patches_overlap_neighborhood`): a second cover of the
    foreground ring between 2 and 5 dilations of the overlap voxels, by the not yet selected
    ranked patches whose centre lies in that ring, with the LAST pixel threshold of the first
    cover and restarting at rank 0.  The result is every selected centre in RASTER order with
    its score from the score volume (the reference rebuilds the list with np.argwhere)."""
    import scipy.ndimage
    import torch
    shape = tuple(np.asarray(mask_to_cover).shape)
    chosen = np.zeros(shape, dtype=bool)
    first = ranked.coords[np.flatnonzero(selected)]
    chosen[tuple(first.T)] = True
    overlap = np.asarray(overlap_mask).copy()
    overlap_t = scipy.ndimage.binary_dilation(overlap, iterations=2)
    overlap_dil = scipy.ndimage.binary_dilation(overlap, iterations=5)
    fg_dil_mask = np.logical_and(np.logical_and(np.logical_not(overlap_t), overlap_dil), mask_to_cover)
    keep = ~chosen[tuple(ranked.coords.T)] & fg_dil_mask[tuple(ranked.coords.T)]
    sub = ranked[np.flatnonzero(keep)]
    if len(sub):
        running, _owner = backend.padded_mask(fg_dil_mask)
        sel2 = np.zeros(len(sub), dtype=np.uint8)
        remaining = int(np.count_nonzero(running[radslice]))
        lin = sub.lin(shape)
        ov8 = np.ascontiguousarray(np.asarray(overlap_mask) > 0).astype(np.uint8)
        for s in range(0, len(sub), COVER_CHUNK):
            if remaining <= 0:
                break
            e = min(len(sub), s + COVER_CHUNK)
            bits = bits_of(sub.coords[s:e])
            remaining, stopped = backend.host_cover_pass(running, ov8, patchshape, lin[s:e], sub.scores[s:e], bits,
                                                         pix_th, thr, sel2[s:e], remaining, marked=marked)
            if stopped:
                break
        more = sub.coords[np.flatnonzero(sel2)]
        chosen[tuple(more.T)] = True
    coords = np.argwhere(chosen)
    if torch.is_tensor(scores_array):
        scores_array = scores_array.cpu().numpy()
    scores = np.asarray(scores_array)[tuple(coords.T)] if len(coords) else np.zeros(0, np.float32)
    out = PatchList(coords, scores)
    return out, len(out)


def _pix_thresholds(patchshape, kwargs):
    if kwargs["select_patches_for_sparse_data"]:
        return [0]
    mid = int(np.prod(patchshape) / 2)
    return [t for t in [500, 100, 50, 10, 0] if t < mid]


def greedy_cover_device(mask, bits, lin, never, pix_ths, radslice, P, silent=True, bits_first_voxel=None):
    """All passes of the cover on the device.  mask uint8 (Z,Y,X) device tensor (cleared in
    place), bits int32 [n, words] patch bits in rank order (or a row per voxel starting at linear
    voxel `bits_first_voxel`), lin int64 [n] centres, never bool [n] patches that do not take
    part.  Returns (selected bool [n] device, uncovered)."""
    import torch
    n = int(lin.numel())
    remaining = int(torch.count_nonzero(mask[tuple(radslice)]).item())
    selected = torch.zeros(n, dtype=torch.bool, device=mask.device)
    total_rounds = 0
    for pix_th in pix_ths:
        if remaining <= 0:
            break
        if not silent:
            logger.info("compute foreground cover, threshold %s", pix_th)
        # every pass restarts at rank 0 (the reference passes rpidx by value)
        state = torch.where(selected, 1, torch.where(never, 2, 0)).to(torch.int32)
        cleared, rounds = backend.cover_pass_device(mask, bits, lin, state, pix_th, P,
                                                    bits_first_voxel=bits_first_voxel)
        total_rounds += rounds
        idx = torch.nonzero((state == 1) & ~selected).flatten()       # rank order
        left = remaining - torch.cumsum(cleared[idx].long(), 0)
        done = torch.nonzero(left <= 0).flatten()
        if done.numel():
            # the sequential loop ends right after the patch that empties the interior
            idx = idx[:int(done[0].item()) + 1]
            remaining = 0
        elif idx.numel():
            remaining = int(left[-1].item())
        selected[idx] = True
        if remaining < 1:
            break
    backend.note("cover_rounds", total_rounds)
    return selected, remaining


def never_selected(overlap_mask, lin_h, scores, score_threshold):
    """Ranked patches the loop never looks at: centre on an overlap voxel
    (foreground_cover.py:140-141), and everything from the first score below score_threshold
    on (the pass breaks there, foreground_cover.py:136-138)."""
    never = np.zeros(len(lin_h), dtype=bool)
    ov = np.asarray(overlap_mask)
    if ov.any():
        never |= ov.reshape(-1)[lin_h] > 0
    if isinstance(score_threshold, float):
        below = np.flatnonzero(np.asarray(scores, dtype=np.float64) < score_threshold)
        if len(below):
            never[int(below[0]):] = True
    return never


def _cover_on_device(overlap_mask, mask_to_cover, patchshape, ranked, radslice, pred_affs, P,
                     silent, **kwargs):
    """Same result as the sequential loop (see csrc/ppp_cover.hip for the argument)."""
    import torch
    dev = pred_affs.device
    if len(ranked) == 0:
        return ranked, 0
    mask = torch.from_numpy(np.ascontiguousarray(np.asarray(mask_to_cover) != 0)
                            .astype(np.uint8)).to(dev)
    lin_h = ranked.lin(mask.shape)
    bits = backend.patch_bits(pred_affs, torch.from_numpy(ranked.coords).to(dev),
                              kwargs["fc_threshold"], P)
    never = never_selected(overlap_mask, lin_h, ranked.scores, kwargs.get("score_threshold", False))
    selected, remaining = greedy_cover_device(
        mask, bits, torch.from_numpy(lin_h).to(dev), torch.from_numpy(never).to(dev),
        _pix_thresholds(patchshape, kwargs), radslice, P, silent)
    sel = ranked[np.flatnonzero(selected.cpu().numpy())]
    if len(sel) and not silent:
        logger.info("num patches to cover foreground: %s best score: %s, worst score: %s, "
                    "uncovered: %s", len(sel), sel.scores[0], sel.scores[-1], remaining)
    return sel, len(sel)


def thinOutForegroundCover(mask_to_cover, selected_patches_list, radslice, pred_affs, rad,
                           patchshape, **kwargs):
    """foreground_cover.py:183-256 (sample == 1.0, no kd-tree shortcut)."""
    if kwargs.get("sample", 1.0) < 1.0:
        raise NotImplementedError("sample < 1 uses unseeded random sampling in the reference")
    sel = PatchList.from_any(selected_patches_list)
    P = backend.params_from_kwargs(pred_affs.shape[1:], patchshape, kwargs)
    mask = np.ascontiguousarray(mask_to_cover).astype(np.uint8)
    if os.environ.get("PPP_THIN", "device") != "host" and int(patchshape[2]) <= 32:
        import torch
        dev = pred_affs.device
        c = torch.from_numpy(np.ascontiguousarray(sel.coords, dtype=np.int32)).to(dev)
        bits_d = backend.patch_bits(pred_affs, c, kwargs["fc_threshold"], P)
        keep = backend.thin_cover_device(torch.from_numpy((mask != 0).astype(np.uint8)).to(dev), bits_d,
                                         torch.from_numpy(sel.lin(mask.shape)).to(dev), P)
        keep = keep.cpu().numpy()
    else:
        bits = _bits_for(pred_affs, sel.coords, kwargs["fc_threshold"], P)
        keep = backend.host_thin_cover(mask, patchshape, sel.lin(mask.shape), bits)
    out = sel[np.flatnonzero(keep)]
    logger.info("num_selected: %s", len(out))
    return out, len(out)
