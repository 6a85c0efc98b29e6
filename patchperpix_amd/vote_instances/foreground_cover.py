"""Steps 3 and 4: greedy foreground cover and set-cover thinning
(reference: PatchPerPix/vote_instances/foreground_cover.py).

Host stages, as in the reference, but the per-patch work runs natively
(ppp_host_cover_pass / ppp_host_thin_cover) on bit masks ``pred[:, c] > fc_threshold`` that
the device packs (ppp_patch_bits), chunk by chunk, so the (C,Z,Y,X) prediction never leaves
the GPU.
"""
import logging

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
    for opt in ("mark_close_neighboorhood", "select_patches_overlap_neighborhood"):
        if kwargs.get(opt, False):
            raise NotImplementedError("%s is not supported" % opt)
    ranked = PatchList.from_any(ranked_patches_list)
    P = backend.params_from_kwargs(pred_affs.shape[1:], patchshape, kwargs)
    running, _owner = backend.padded_mask(mask_to_cover)
    overlap = np.ascontiguousarray(np.asarray(overlap_mask) > 0).astype(np.uint8)
    selected = np.zeros(len(ranked), dtype=np.uint8)
    if kwargs["select_patches_for_sparse_data"]:
        pix_ths = [0]
    else:
        mid = int(np.prod(patchshape) / 2)
        pix_ths = [t for t in [500, 100, 50, 10, 0] if t < mid]
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
            bits = _bits_for(pred_affs, ranked.coords[s:e], kwargs["fc_threshold"], P)
            remaining, stopped = backend.host_cover_pass(
                running, overlap, patchshape, lin[s:e], ranked.scores[s:e], bits, pix_th, thr,
                selected[s:e], remaining)
            if stopped:
                break  # the pass hit the score threshold
        if remaining < 1:
            break
    sel = ranked[np.flatnonzero(selected)]
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
    bits = _bits_for(pred_affs, sel.coords, kwargs["fc_threshold"], P)
    keep = backend.host_thin_cover(mask, patchshape, sel.lin(mask.shape), bits)
    out = sel[np.flatnonzero(keep)]
    logger.info("num_selected: %s", len(out))
    return out, len(out)
