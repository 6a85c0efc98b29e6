"""Step 2: patch ranking (reference: PatchPerPix/vote_instances/ranked_patches.py)."""
import logging
import os

import numpy as np

from .. import backend
from .consensus_array import _device_overlap

logger = logging.getLogger(__name__)


class PatchList:
    """The reference's ``[(coord, score), ...]`` list, held as arrays.  Indexing and
    iteration yield ``(coord ndarray[3], score)`` tuples like the reference's list does."""

    def __init__(self, coords, scores):
        self.coords = np.ascontiguousarray(np.asarray(coords).reshape(-1, 3), dtype=np.int32)
        self.scores = np.ascontiguousarray(np.asarray(scores).reshape(-1), dtype=np.float32)
        assert len(self.coords) == len(self.scores)

    def __len__(self):
        return len(self.coords)

    def __getitem__(self, i):
        if isinstance(i, slice) or isinstance(i, np.ndarray):
            return PatchList(self.coords[i], self.scores[i])
        return (self.coords[i], self.scores[i])

    def __iter__(self):
        for i in range(len(self)):
            yield (self.coords[i], self.scores[i])

    def lin(self, shape):
        c = self.coords.astype(np.int64)
        return np.ascontiguousarray((c[:, 0] * shape[1] + c[:, 1]) * shape[2] + c[:, 2])

    @staticmethod
    def from_any(obj):
        if isinstance(obj, PatchList):
            return obj
        obj = list(obj)
        if len(obj) == 0:
            return PatchList(np.zeros((0, 3)), np.zeros((0,)))
        return PatchList(np.array([np.asarray(o[0]) for o in obj]),
                         np.array([o[1] for o in obj]))


def rank_patches_cuda(pred_affs, consensus_vote_array, patchshape, neighshape, overlap_mask,
                      **kwargs):
    """ranked_patches.py:33-74: returns the (Z,Y,X) float32 score volume (device tensor)."""
    P = backend.params_from_kwargs(pred_affs.shape[1:], patchshape, kwargs)
    ov = _device_overlap(overlap_mask, pred_affs) if P.use_overlap else None
    if P.cons_layout == backend.CONS_COMPACT and backend.rank_vm_available(P):
        # row-stationary ranking on the voxel-major re-layout; the patch-graph stage needs the
        # same re-layout and takes it from the consensus tensor (backend.patch_graph_auto)
        vm, Pv = backend.cons_to_voxel_major(consensus_vote_array, P)
        consensus_vote_array._ppp_vm = (vm, Pv)
        return backend.rank_patches(pred_affs, vm, ov, Pv, score_box=kwargs.get("score_box"))
    return backend.rank_patches(pred_affs, consensus_vote_array, ov, P,
                                score_box=kwargs.get("score_box"))


def rank_patches_by_score(all_patches_idx, rank_scores, foreground=None, patchshape=None):
    """ranked_patches.py:21-30: stable sort, best score first.

    Fast path (``foreground`` given, ``all_patches_idx`` None): the interior foreground
    voxels are enumerated in raster order and stably sorted natively (ppp_host_rank_order),
    which is the list the reference builds at vote_instances.py:276,286-287."""
    scores = rank_scores.detach().cpu().numpy() if hasattr(rank_scores, "detach") \
        else np.asarray(rank_scores)
    if all_patches_idx is None:
        lin = backend.host_rank_order(scores, foreground, patchshape)
        coords = np.stack(np.unravel_index(lin, scores.shape), axis=1)
        return PatchList(coords, scores.reshape(-1)[lin])
    coords = np.asarray(all_patches_idx).reshape(-1, 3)
    s = scores[tuple(coords.T)]
    order = np.argsort(-s.astype(np.float64), kind="stable")
    return PatchList(coords[order], s[order])


def load_ranked_patches(**kwargs):
    """ranked_patches.py:137-139: the stored ranked list, or None."""
    path = kwargs.get("ranked_patches")
    if path is None or not os.path.exists(path):
        return None
    from .utilVoteInstances import loadFromFile
    obj = loadFromFile(path)
    if isinstance(obj, np.ndarray) and obj.ndim == 2 and obj.shape[1] == 4:
        return PatchList(obj[:, :3].astype(np.int32), obj[:, 3])
    return PatchList.from_any(obj)


def store_ranked_patches(ranked, **kwargs):
    """ranked_patches.py:188-192: ``ranking.pickle`` in the result folder -- the reference's own
    format (a list of (coordinate array, score)), unless save_no_intermediates."""
    if kwargs.get("save_no_intermediates", True):
        return None
    import pickle
    fn = os.path.join(kwargs["result_folder"], "ranking.pickle")
    with open(fn, "wb") as f:
        pickle.dump([(np.array(c), s.item() if hasattr(s, "item") else s) for c, s in ranked], f, protocol=4)
    return fn


def loadOrComputePatchRanking(pred_affs=None, consensus_vote_array=None, offsets_bases_ff=None,
                              offsets_bases_fb=None, overlap_mask=None, all_patches=None,
                              patchshape=None, neighshape=None, rad=None, **kwargs):
    """ranked_patches.py:108-213 (device branch).  A stored ranking (`ranked_patches`: the
    reference's ``ranking.pickle`` -- a list of (coordinate, score) in rank order -- or an .npy of
    rows (z, y, x, score)) is loaded instead of computed (:137-139); the score volume is then
    unknown (None), as in the reference."""
    stored = load_ranked_patches(**kwargs)
    if stored is not None:
        return stored, None
    if not kwargs["cuda"]:
        raise RuntimeError("the NumPy-semantics ranking (cuda=False) lives in numpy_semantics.py")
    scores = rank_patches_cuda(pred_affs, consensus_vote_array, patchshape, neighshape,
                               overlap_mask, **kwargs)
    scores_array = scores.cpu().numpy()
    if all_patches is None and kwargs.get("_foreground") is not None:
        # stable descending sort on the device (same order as Python's sorted(..., reverse=True))
        lin, s = backend.rank_order_device(scores, kwargs["_foreground"], patchshape)
        ranked = PatchList(np.stack(np.unravel_index(lin, scores_array.shape), axis=1), s)
    else:
        ranked = rank_patches_by_score(all_patches, scores_array,
                                       foreground=kwargs.get("_foreground"),
                                       patchshape=patchshape)
    if len(ranked):
        logger.info("best/worst score: %s %s", ranked.scores[0], ranked.scores[-1])
    return ranked, scores_array
