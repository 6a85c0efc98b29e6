"""Patch pairs and step 5, the patch graph
(reference: PatchPerPix/vote_instances/aff_patch_graph.py)."""
import logging
import os

import numpy as np

from .. import backend
from .ranked_patches import PatchList

logger = logging.getLogger(__name__)


class AffGraph:
    """What the reference keeps as an ``nx.Graph`` (setAffgraph, aff_patch_graph.py:31-40):
    the pair rows and their affinities.  Rows with aff == 0 are not edges."""

    def __init__(self, graph_mat, computed_pairs):
        self.pairs = np.ascontiguousarray(np.asarray(computed_pairs).reshape(-1, 6),
                                          dtype=np.uint32)
        self.aff = np.ascontiguousarray(np.asarray(graph_mat).reshape(-1), dtype=np.float32)
        assert len(self.pairs) == len(self.aff)

    def number_of_edges(self):
        return int(np.count_nonzero(self.aff != 0))

    def to_networkx(self):
        import networkx as nx
        g = nx.Graph()
        for idx in np.flatnonzero(self.aff != 0):
            g.add_edge(tuple(int(v) for v in self.pairs[idx, :3]),
                       tuple(int(v) for v in self.pairs[idx, 3:6]), aff=self.aff[idx])
        return g


def setAffgraph(graphMat, computed_pairs):
    logger.info("len graphmat %s, num pairs %s", len(graphMat), np.asarray(computed_pairs).shape[0])
    return AffGraph(graphMat, computed_pairs)


def loadAffgraph(affgraph, selected_patch_pairs):
    """aff_patch_graph.py:20-28."""
    if affgraph.endswith(".npy"):
        return setAffgraph(np.load(affgraph), np.load(selected_patch_pairs))
    logger.error("invalid affgraph file")
    raise SystemExit(-1)


def computeAndStorePatchPairs(selected_patches_list, patchshape, **kwargs):
    """aff_patch_graph.py:43-110.  The candidate set (cKDTree L1 ball, then the per-axis box
    ``|d_i| <= max_total_patch_distance_in_ps_multiples * p_i``) is enumerated natively with
    a grid hash.  Rows keep the reference's orientation (A before B in the x-sorted list);
    their ORDER is canonical -- sorted by (index A, index B) -- where the reference's is the
    iteration order of a Python set.  Returns uint32 [N, 6] (host) or None."""
    sel = PatchList.from_any(selected_patches_list)
    sorted_zyx, pairs = backend.host_patch_pairs(
        sel.coords, patchshape,
        max_ps_dist=kwargs.get("max_total_patch_distance_in_ps_multiples", 2),
        include_single=kwargs["includeSinglePatchCCS"])
    if pairs is None:
        logger.info("Sorry, no patch pairs in sample! Returning...")
        return None
    logger.info("num pairs (incl single patch ccs) %s", len(pairs))
    if not kwargs["save_no_intermediates"]:
        np.save(os.path.join(kwargs["result_folder"], "selected_patch_pairs.npy"), pairs)
        np.save(os.path.join(kwargs["result_folder"], "selected_patches_list.npy"),
                sorted_zyx.astype(np.uint32))
    return pairs


def computePatchGraph_cuda(pred_affs, consensus_vote_array, selected_patch_pairsIDs, patchshape,
                           neighshape, **kwargs):
    """aff_patch_graph.py:113-187: one launch instead of the reference's 512-pair batches."""
    import torch
    P = backend.params_from_kwargs(pred_affs.shape[1:], patchshape, kwargs)
    pairs_host = np.ascontiguousarray(selected_patch_pairsIDs, dtype=np.uint32)
    pairs_dev = torch.from_numpy(pairs_host.view(np.int32)).to(pred_affs.device)
    aff = backend.patch_graph(pred_affs, consensus_vote_array, pairs_dev, P)
    if kwargs.get("_keep_on_device", False):
        return aff, pairs_dev
    affinity_graph_mat = aff.cpu().numpy()
    if kwargs.get("save_patch_graph", False) or kwargs.get("termAfterPatchGraph", False):
        fn = os.path.splitext(os.path.basename(kwargs["affinities"]))[0]
        np.save(os.path.join(kwargs["result_folder"], fn + "_selected_patch_pairs.npy"),
                pairs_host)
        np.save(os.path.join(kwargs["result_folder"], fn + "_aff_graph.npy"), affinity_graph_mat)
    if kwargs.get("return_intermediates"):
        return affinity_graph_mat
    if len(affinity_graph_mat):
        logger.info("affinity_graph_mat: %s %s", np.min(affinity_graph_mat),
                    np.max(affinity_graph_mat))
    return setAffgraph(affinity_graph_mat, pairs_host)


def computePatchGraph(selected_patches_list, num_selected, selected_patch_pairsIDs, pred_affs,
                      mask_to_cover, patchshape, neighshape, rad, multiple_worms, lookup,
                      consensus_vote_array, **kwargs):
    """aff_patch_graph.py:190-282 (device branch only)."""
    if not kwargs["cuda"]:
        raise RuntimeError("patchperpix_amd only implements the device path (cuda=True)")
    return computePatchGraph_cuda(pred_affs, consensus_vote_array, selected_patch_pairsIDs,
                                  patchshape, neighshape, **kwargs)
