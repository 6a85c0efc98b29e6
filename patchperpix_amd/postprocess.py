"""Post-steps of the blockwise `label` driver (reference: PatchPerPix/util/postprocess.py:24-52 and
PatchPerPix/vote_instances/stitch_patch_graph.py:831-894): drop small instances, renumber,
dilate.  Host NumPy like the reference -- they touch the finished uint16 map once."""
import numpy as np
from scipy import ndimage


def remove_small_components(array, compsize=5):
    """Instances of at most `compsize` voxels become background (postprocess.py:24-37)."""
    labels, inverse, counts = np.unique(array, return_inverse=True, return_counts=True)
    keep = np.where(counts <= compsize, 0, labels).astype(array.dtype)
    return keep[inverse].reshape(array.shape)


def relabel(array, start=None):
    """Consecutive ids in ascending order of the old ones, from `start` (default 1); 0 stays 0
    (postprocess.py:40-52)."""
    labels, inverse = np.unique(array, return_inverse=True)
    first = 1 if start is None else int(start)
    new = np.zeros(len(labels), dtype=array.dtype)
    nz = labels != 0
    new[nz] = first + np.arange(int(np.count_nonzero(nz)))
    return new[inverse].reshape(array.shape)


def dilate_instances(instances, iterations=1):
    """stitch_patch_graph.py:871-880: every instance, in ascending id order, is dilated by one
    step of the cross-shaped structuring element and painted over what is there (later ids win,
    and an earlier dilation is seen by the later masks -- the loop works in place)."""
    out = np.copy(instances)
    for lbl in np.unique(instances):
        if lbl == 0:
            continue
        out[ndimage.binary_dilation(out == lbl, iterations=iterations)] = lbl
    return out
