"""Step 1: consensus array (reference: PatchPerPix/vote_instances/consensus_array.py).

``create_consensus_array_cuda`` keeps the reference's name and arguments
(consensus_array.py:71-206).  Where the reference JIT-compiles fillConsensusArray.cu,
launches it once or twice (value pass, -DOUTPUT_CNT pass) and then normConsensusArray.cu,
this calls ONE fused, deterministic HIP kernel (``ppp_consensus``).  The result is a device
tensor in the compact plane layout (include/ppp_mi355x.h); the reference's
[NSZ,NSY,NSX,Z,Y,X] array is materialised only for ``save_consensus``.
"""
import logging
import os

import numpy as np

from .. import backend

logger = logging.getLogger(__name__)


def create_consensus_array_cuda(pred_affs, overlap_mask, patchshape, neighshape, **kwargs):
    """pred_affs: device tensor (C,Z,Y,X) f32/f16; overlap_mask: device uint8 (Z,Y,X) or a
    host array.  Flags consumed: patch_threshold, vi_bg_use_*, overlapping_inst,
    consensus_norm_prob_product, consensus_prob_product, consensus_norm_aff,
    consensus_interleaved_cnt (both settings give identical results and are served by the
    same fused kernel), flip_cons_arr_axes (layout is internal here; accepted)."""
    if kwargs.get("consensus_interleaved_cnt", True):
        assert kwargs.get("consensus_norm_aff", True), \
            "consensus aff not normalized so no computation required"
    P = backend.params_from_kwargs(pred_affs.shape[1:], patchshape, kwargs)
    ov = _device_overlap(overlap_mask, pred_affs) if P.use_overlap else None
    logger.info("creating consensus array %s", kwargs.get("affinities"))
    cons = backend.consensus(pred_affs, ov, P)
    logger.info("consensus array shape %s", tuple(cons.shape))
    if kwargs.get("save_consensus", False):
        fn = os.path.splitext(os.path.basename(kwargs["affinities"]))[0]
        ref = backend.cons_to_reference(cons, P).cpu().numpy()
        if kwargs.get("flip_cons_arr_axes", False):
            ref = np.ascontiguousarray(np.moveaxis(ref, (0, 1, 2), (3, 4, 5)))
        np.save(os.path.join(kwargs["result_folder"], fn + "_consensus.npy"), ref)
    return cons


def _device_overlap(overlap_mask, like):
    import torch
    if isinstance(overlap_mask, torch.Tensor):
        return overlap_mask.to(device=like.device, dtype=torch.uint8).contiguous()
    return torch.from_numpy(np.ascontiguousarray(overlap_mask != 0).astype(np.uint8)).to(like.device)


def loadOrComputeConsensus(instances, patchshape, neighshape, all_patches, pred_affs, rad,
                           foreground, lookup, overlap_mask, **kwargs):
    """consensus_array.py:209-246.  Only the device branch exists in this package."""
    if not kwargs["cuda"]:
        raise RuntimeError("the NumPy-semantics stages (cuda=False) live in numpy_semantics.py; "
                           "this function is the kernel path's (cuda=True)")
    path = kwargs.get("consensus")
    if path is not None and os.path.exists(path):
        # consensus_array.py:213-218 (resume).  What the kernel path stores is the array
        # `save_consensus` writes (:202-206): float32 [NSZ, NSY, NSX, Z, Y, X] (.npy / hdf / zarr
        # with `consensus_key`); a `consensus.pickle` of the NumPy path holds int16 votes of another
        # function and is refused here.  (The reference unpacks the loaded object into three names,
        # which only works for that pickle: its own .npy cannot be resumed from.)
        from .utilVoteInstances import loadFromFile
        arr = loadFromFile(path, key=kwargs.get("consensus_key"))
        if isinstance(arr, (list, tuple)) or np.asarray(arr).dtype != np.float32:
            raise ValueError("%s does not hold a float32 consensus array of the kernel path" % path)
        return load_reference_layout(np.asarray(arr), pred_affs, patchshape, **kwargs), None, None
    cons = create_consensus_array_cuda(pred_affs, overlap_mask, patchshape, neighshape, **kwargs)
    return cons, None, None


def load_reference_layout(arr, pred_affs, patchshape, **kwargs):
    """[NSZ, NSY, NSX, Z, Y, X] (or flipped, `flip_cons_arr_axes`) float32 host array -> the
    compact plane layout on the device: plane (dz, dy, dx) lexicographically positive =
    arr[dz + pz - 1, dy + py - 1, dx + px - 1]."""
    import torch
    ps = [int(p) for p in patchshape]
    shape = tuple(int(v) for v in pred_affs.shape[1:])
    if kwargs.get("flip_cons_arr_axes", False):
        arr = np.moveaxis(arr, (3, 4, 5), (0, 1, 2))
    ns = (2 * ps[0] if ps[0] > 1 else ps[0], 2 * ps[1], 2 * ps[2])
    if tuple(arr.shape) != ns + shape:
        raise ValueError("consensus array of shape %s, expected %s" % (tuple(arr.shape), ns + shape))
    planes = []
    for dz in range(0, ps[0]):
        for dy in range(-(ps[1] - 1), ps[1]):
            for dx in range(-(ps[2] - 1), ps[2]):
                if (dz, dy, dx) > (0, 0, 0):
                    planes.append(arr[dz + ps[0] - 1, dy + ps[1] - 1, dx + ps[2] - 1])
    return torch.from_numpy(np.ascontiguousarray(np.stack(planes, axis=0), dtype=np.float32)).to(pred_affs.device)
