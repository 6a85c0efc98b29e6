"""Blockwise driver (reference: PatchPerPix/vote_instances/stitch_patch_graph.py).

The reference scales to large volumes by cutting them into ``chunksize`` blocks, running
``do_block(return_intermediates)`` per block, recomputing inter-block edges on the face
overlaps and labelling one global patch graph (:110-399, :553-669, :672-894).  On MI355X a
288 GB device holds what the reference needed blocks for, and scale-out is done by the
spatial tiling in ``patchperpix_amd.tiling`` / ``patchperpix_amd.distributed``.  This module
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
    values_map = np.arange(int(array.max() + 1), dtype=new_values.dtype)
    values_map[old_values] = new_values
    return values_map[array]


def clean_mask(mask, structure, size):
    """stitch_patch_graph.py:46-57: drop connected components of at most `size` voxels."""
    labeled = ndimage.label(mask, structure)[0]
    labels, counts = np.unique(labeled, return_counts=True)
    labels = labels[counts <= size]
    labeled = replace(labeled, np.array(labels), np.array([0] * len(labels)))
    logger.info('removing %i of small components.' % len(labels))
    return labeled > 0


def get_offset_str(offset):
    return "_".join(str(off) for off in offset)


def get_offsets(total_shape, chunksize):
    """stitch_patch_graph.py:425-440: block origins in raster order."""
    if len(total_shape) not in (2, 3):
        raise NotImplementedError
    grids = np.meshgrid(*[np.arange(0, total_shape[i], chunksize[i])
                          for i in range(len(total_shape))], indexing="ij")
    return [np.array(o) for o in np.stack([g.ravel() for g in grids], axis=1)]


def load_input(io, key, offset, context, overlap, output_shape, padding=True,
               padding_mode='constant'):
    """stitch_patch_graph.py:443-516: read block + margin, optionally padding at the borders.
    Returns (data, padded margin per axis)."""
    starts = [off - context[i] - overlap[i] for i, off in enumerate(offset)]
    stops = [off + output_shape[i] + overlap[i] + context[i] for i, off in enumerate(offset)]
    shape = io.shape[1:] if io.channel_order is not None else io.shape
    unsqueezed = len(shape) == 2
    if unsqueezed:
        shape = (1,) + tuple(shape)
    padded = np.array(context) + np.array(overlap)
    if np.any(np.array(starts) < 0):
        padded[np.array(starts) < 0] = 0
    pad_left = pad_right = None
    if padding:
        if any(s < 0 for s in starts):
            pad_left = tuple(abs(s) if s < 0 else 0 for s in starts)
            starts = [max(0, s) for s in starts]
        if any(stop > shape[i] for i, stop in enumerate(stops)):
            pad_right = tuple(stop - shape[i] if stop > shape[i] else 0
                              for i, stop in enumerate(stops))
            stops = [min(shape[i], stop) for i, stop in enumerate(stops)]
    else:
        starts = list(np.maximum([0, 0, 0], starts))
        stops = list(np.minimum(shape, stops))
    if unsqueezed:
        del starts[0]
        del stops[0]
    bb = tuple(slice(int(a), int(b)) for a, b in zip(starts, stops))
    if io.channel_order is not None:
        try:
            bb = (io.channel_order[io.keys.index(key)],) + bb
        except Exception:
            pass
    data = io.read(bb, key)
    if unsqueezed:
        data = np.expand_dims(data, axis=1)
    if pad_left is not None or pad_right is not None:
        pad_left = (0, 0, 0) if pad_left is None else pad_left
        pad_right = (0, 0, 0) if pad_right is None else pad_right
        pad_width = tuple((pl, pr) for pl, pr in zip(pad_left, pad_right))
        if io.channel_order is not None:
            pad_width = ((0, 0),) + pad_width
        data = np.pad(data, pad_width, mode=padding_mode)
    return data, padded


def verify_shape(offset, output, shape, chunksize):
    """stitch_patch_graph.py:519-551: crop a block result to the volume."""
    tmp_channel = offset[0]
    offset = np.array(offset[1:])
    actual = np.array(output.shape)
    overlap = ((actual - np.array(chunksize)) / 2).astype(int)
    starts = (offset - overlap).astype(int)
    if np.any(starts < 0):
        bb = tuple(slice(a, b) for a, b in
                   zip(np.abs(np.minimum(np.zeros(len(offset), dtype=int), starts)), actual))
        output = output[bb]
    stops = offset + np.array(chunksize) + overlap
    if np.any(stops > np.array(shape)):
        bb = tuple(slice(0, dim - off if stop > dim else None)
                   for stop, dim, off in zip(stops, shape, offset))
        output = output[bb]
    starts = np.maximum(np.zeros(len(offset), dtype=int), starts)
    bounding_box = (slice(tmp_channel, tmp_channel + 1),) + tuple(
        slice(s, s + o) for s, o in zip(starts, output.shape))
    return np.reshape(output, (1,) + output.shape), bounding_box


def write_output(io_out, output, output_bounding_box):
    io_out.write(output, output_bounding_box)


def main(pred_file, result_folder='.', **kwargs):
    """stitch_patch_graph.py:672-894 entry point.

    The whole (bounding-boxed) volume is assembled in ONE pass on the device -- tiled
    internally when the consensus array does not fit (patchperpix_amd.tiling) -- instead of
    per-block graphs that are stitched afterwards.  Output datasets and dtypes follow the
    reference: ``vote_instances``, ``vote_foreground``, ``vote_instances_masked`` (uint16)."""
    from .. import tiling
    return tiling.stitch_main(pred_file, result_folder=result_folder, **kwargs)
