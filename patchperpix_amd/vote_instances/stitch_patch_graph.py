"""Blockwise driver (reference: PatchPerPix/vote_instances/stitch_patch_graph.py).

The reference scales to large volumes by cutting them into ``chunksize`` blocks, running
``do_block(return_intermediates)`` per block, recomputing inter-block edges on the face
overlaps and labelling one global patch graph (:110-399, :553-669, :672-894).  On MI355X a
288 GB device holds what the reference needed blocks for, and scale-out is done by the
spatial tiling in ``patchperpix_amd.tiling``.  This module
keeps the reference's helper functions and the ``main`` entry point.
"""
import logging
import os

import numpy as np
from scipy import ndimage

from .vote_instances import do_block, to_instance_seg, write_result
from .graph_to_labeling import affGraphToInstances
from .utilVoteInstances import loadFg, returnFg, loadAffinities
from . import io_hdflike

logger = logging.getLogger(__name__)


def replace(array, old_values, new_values):
    """Relabel: every occurrence of old_values[i] becomes new_values[i] (lookup table over the
    value range of `array`; same contract as stitch_patch_graph.py:38-43)."""
    new_values = np.asarray(new_values)
    lut = np.arange(int(np.max(array)) + 1).astype(new_values.dtype)
    np.put(lut, np.asarray(old_values, dtype=np.int64), new_values)
    return np.take(lut, array)


def clean_mask(mask, structure, size):
    """Mask without its connected components of at most `size` voxels
    (stitch_patch_graph.py:46-57; `structure` is scipy.ndimage's connectivity)."""
    labeled, n = ndimage.label(mask, structure)
    big = np.bincount(labeled.ravel(), minlength=n + 1) > size
    big[0] = False
    logger.info("removing %i of small components.", int(n - np.count_nonzero(big)))
    return big[labeled]


def get_offset_str(offset):
    return "_".join("%s" % o for o in offset)


def get_offsets(total_shape, chunksize):
    """stitch_patch_graph.py:425-440: block origins in raster order."""
    if len(total_shape) not in (2, 3):
        raise NotImplementedError
    grids = np.meshgrid(*[np.arange(0, total_shape[i], chunksize[i])
                          for i in range(len(total_shape))], indexing="ij")
    return [np.array(o) for o in np.stack([g.ravel() for g in grids], axis=1)]


def load_input(io, key, offset, context, overlap, output_shape, padding=True,
               padding_mode='constant'):
    """One block of `key` with its margin (stitch_patch_graph.py:443-516): the window
    [offset - context - overlap, offset + output_shape + overlap + context) per axis, read where it
    intersects the array and -- with ``padding`` -- padded where it does not.  Returns
    (data, margin) where margin = context + overlap, 0 on axes whose window starts before the
    array (the reference's convention for locating the block inside the returned data)."""
    has_channels = io.channel_order is not None
    full = tuple(io.shape[1:] if has_channels else io.shape)
    flat = len(full) == 2                       # 2-d data: a z axis of one slice is implied
    if flat:
        full = (1,) + full
    full = np.array(full)
    margin = np.array(context) + np.array(overlap)
    lo = np.array(offset) - margin
    hi = np.array(offset) + np.array(output_shape) + margin
    before, after = np.maximum(-lo, 0), np.maximum(hi - full, 0)
    margin = np.where(lo < 0, 0, margin)
    lo, hi = np.maximum(lo, 0), np.minimum(hi, full)
    spatial = [slice(int(a), int(b)) for a, b in zip(lo, hi)]
    if flat:
        spatial = spatial[1:]
    bb = tuple(spatial)
    if has_channels:
        try:
            bb = (io.channel_order[io.keys.index(key)],) + bb
        except Exception:
            pass
    data = io.read(bb, key)
    if flat:
        data = np.expand_dims(data, axis=1)
    if padding and (before.any() or after.any()):
        widths = [(int(b), int(a)) for b, a in zip(before, after)]
        if has_channels:
            widths = [(0, 0)] + widths
        data = np.pad(data, tuple(widths), mode=padding_mode)
    return data, margin


def verify_shape(offset, output, shape, chunksize):
    """Fit a block result (with its overlap on every side) into the volume
    (stitch_patch_graph.py:519-551).  offset = (channel, z, y, x); returns (output[None], box)."""
    channel, origin = offset[0], np.array(offset[1:])
    chunk, dims = np.array(chunksize), np.array(shape)
    overlap = ((np.array(output.shape) - chunk) / 2).astype(int)
    first = (origin - overlap).astype(int)
    if (first < 0).any():                       # the block sticks out in front: drop that part
        output = output[tuple(slice(int(max(0, -f)), None) for f in first)]
    last = origin + chunk + overlap
    if (last > dims).any():                     # ... or behind (the reference cuts at dim - origin)
        output = output[tuple(slice(0, int(d - o)) if l > d else slice(0, None)
                              for l, d, o in zip(last, dims, origin))]
    first = np.maximum(first, 0)
    box = (slice(channel, channel + 1),) + tuple(slice(int(f), int(f) + n)
                                                 for f, n in zip(first, output.shape))
    return output[np.newaxis], box


def write_output(io_out, output, output_bounding_box):
    io_out.write(output, output_bounding_box)


def main(pred_file, result_folder='.', **kwargs):
    """stitch_patch_graph.py:672-894 entry point.

    Default (``blockwise_semantics="whole_volume"``): the whole (bounding-boxed) volume is
    assembled in ONE pass on the device -- tiled internally when the consensus array does not
    fit (patchperpix_amd.tiling) -- which equals the whole-volume result.
    ``blockwise_semantics="reference"``: the reference's per-block cover, on-disk block graphs
    (``volumes/blocks/<z_y_x>/{patch_pairs, aff_graph_mat}``), inter-block edges and global
    labelling (patchperpix_amd.blockwise; pinned to goldens of the reference's own driver).
    Output datasets and dtypes follow the reference: ``vote_instances``, ``vote_foreground``,
    ``vote_instances_masked`` (uint16)."""
    from patchperpix_amd import backend as _backend
    _backend.tune_host_allocator(cli=__name__ == "__main__")
    if kwargs.pop("blockwise_semantics", "whole_volume") == "reference":
        # the reference's own function of the input: per-block cover, block graphs on disk,
        # inter-block edges, one global labelling (patchperpix_amd/blockwise.py)
        from .. import blockwise
        return blockwise.main(pred_file, result_folder=result_folder, **kwargs)
    from .. import tiling
    return tiling.stitch_main(pred_file, result_folder=result_folder, **kwargs)
