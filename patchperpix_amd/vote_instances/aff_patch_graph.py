"""Patch pairs and step 5, the patch graph
(reference: PatchPerPix/vote_instances/aff_patch_graph.py).

The pair rows (tens of millions on a dense 140^3 volume) are produced, ordered, scored and
labelled ON THE DEVICE; they are copied to the host only when a caller asks for them
(``return_intermediates``, ``save_patch_graph``, mutex watershed)."""
import logging
import os

import numpy as np

from .. import backend
from .ranked_patches import PatchList

logger = logging.getLogger(__name__)


class PatchPairs:
    """The reference's ``selected_patch_pairsIDs`` array (uint32 [N, 6]) living on the device,
    plus the node list (x-sorted selected patches) it was built from."""

    def __init__(self, rows_dev, nodes_host):
        self.rows_dev = rows_dev                       # device int32 [N, 6]
        self.nodes = np.ascontiguousarray(nodes_host, dtype=np.int32).reshape(-1, 3)
        self._host = None
        self.unique_pairs = False      # True for lists made by computeAndStorePatchPairs

    @staticmethod
    def from_host(rows, device):
        import torch
        rows = np.ascontiguousarray(np.asarray(rows, dtype=np.uint32).reshape(-1, 6))
        nodes = np.unique(rows.reshape(-1, 3), axis=0)
        pp = PatchPairs(torch.from_numpy(rows.view(np.int32)).to(device), nodes)
        pp._host = rows
        return pp

    def __len__(self):
        return int(self.rows_dev.shape[0])

    @property
    def shape(self):
        return (len(self), 6)

    def numpy(self):
        if self._host is None:
            self._host = self.rows_dev.cpu().numpy().view(np.uint32)
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)


class AffGraph:
    """What the reference keeps as an ``nx.Graph`` (setAffgraph, aff_patch_graph.py:31-40):
    the pair rows and their affinities.  Rows with aff == 0 are not edges."""

    def __init__(self, graph_mat, computed_pairs, device="cuda"):
        import torch
        if not isinstance(computed_pairs, PatchPairs):
            computed_pairs = PatchPairs.from_host(computed_pairs, device)
        self.pairs_obj = computed_pairs
        if torch.is_tensor(graph_mat):
            self.aff_dev, self._aff = graph_mat, None
        else:
            self._aff = np.ascontiguousarray(np.asarray(graph_mat).reshape(-1), dtype=np.float32)
            self.aff_dev = torch.from_numpy(self._aff).to(computed_pairs.rows_dev.device)
        assert len(self.pairs_obj) == int(self.aff_dev.shape[0])

    @property
    def pairs(self):
        return self.pairs_obj.numpy()

    @property
    def aff(self):
        if self._aff is None:
            self._aff = self.aff_dev.cpu().numpy()
        return self._aff

    def number_of_edges(self):
        return int((self.aff_dev != 0).sum().item())

    def to_networkx(self):
        import networkx as nx
        g = nx.Graph()
        for idx in np.flatnonzero(self.aff != 0):
            g.add_edge(tuple(int(v) for v in self.pairs[idx, :3]),
                       tuple(int(v) for v in self.pairs[idx, 3:6]), aff=self.aff[idx])
        return g


def setAffgraph(graphMat, computed_pairs):
    logger.info("len graphmat %s, num pairs %s", len(graphMat), len(computed_pairs))
    return AffGraph(graphMat, computed_pairs)


def loadAffgraph(affgraph, selected_patch_pairs):
    """aff_patch_graph.py:20-28."""
    if affgraph.endswith(".npy"):
        return setAffgraph(np.load(affgraph), np.load(selected_patch_pairs))
    logger.error("invalid affgraph file")
    raise SystemExit(-1)


def computeAndStorePatchPairs(selected_patches_list, patchshape, **kwargs):
    """aff_patch_graph.py:43-110.  The selected list is stably sorted by x on the host (it is
    small); the candidate pairs (cKDTree L1 ball, then the per-axis box
    ``|d_i| <= max_total_patch_distance_in_ps_multiples * p_i``) are enumerated on the device
    (ppp_patch_pairs_count / _fill).  Rows keep the reference's orientation (A before B in the
    x-sorted list); their ORDER is canonical -- sorted by (index A, index B) -- where the
    reference's is the iteration order of a Python set.  Returns PatchPairs or None."""
    import torch
    sel = PatchList.from_any(selected_patches_list)
    order = np.argsort(sel.coords[:, 2], kind="stable")
    sorted_zyx = np.ascontiguousarray(sel.coords[order])
    shape = kwargs.get("_volume_shape")
    P = backend.params_from_kwargs(shape, patchshape, kwargs)
    device = kwargs.get("_device", "cuda")
    pts_dev = torch.from_numpy(sorted_zyx).to(device)
    rows = backend.device_patch_pairs(
        pts_dev, P, max_ps_dist=kwargs.get("max_total_patch_distance_in_ps_multiples", 2),
        include_single=kwargs["includeSinglePatchCCS"])
    if rows is None:
        logger.info("Sorry, no patch pairs in sample! Returning...")
        return None
    pairs = PatchPairs(rows, sorted_zyx)
    pairs.unique_pairs = True
    logger.info("num pairs (incl single patch ccs) %s", len(pairs))
    if not kwargs["save_no_intermediates"]:
        np.save(os.path.join(kwargs["result_folder"], "selected_patch_pairs.npy"), pairs.numpy())
        np.save(os.path.join(kwargs["result_folder"], "selected_patches_list.npy"),
                sorted_zyx.astype(np.uint32))
    return pairs


def computePatchGraph_cuda(pred_affs, consensus_vote_array, selected_patch_pairsIDs, patchshape,
                           neighshape, **kwargs):
    """aff_patch_graph.py:113-187: one launch instead of the reference's 512-pair batches; rows
    are assigned to lanes grouped by patch offset (backend.pair_order)."""
    P = backend.params_from_kwargs(pred_affs.shape[1:], patchshape, kwargs)
    pairs = selected_patch_pairsIDs
    if not isinstance(pairs, PatchPairs):
        pairs = PatchPairs.from_host(pairs, pred_affs.device)
    if P.cons_layout == backend.CONS_COMPACT:
        # one streaming re-layout to voxel-major, then one workgroup per patch A serving all its
        # pairs from LDS-staged consensus rows (or pair-per-lane kernels for other patch shapes)
        aff = backend.patch_graph_auto(pred_affs, consensus_vote_array, pairs.rows_dev, P)
    else:
        aff = backend.patch_graph(pred_affs, consensus_vote_array, pairs.rows_dev, P,
                                  order=backend.pair_order(pairs.rows_dev, P))
    if kwargs.get("save_patch_graph", False) or kwargs.get("termAfterPatchGraph", False):
        fn = os.path.splitext(os.path.basename(kwargs["affinities"]))[0]
        np.save(os.path.join(kwargs["result_folder"], fn + "_selected_patch_pairs.npy"),
                pairs.numpy())
        np.save(os.path.join(kwargs["result_folder"], fn + "_aff_graph.npy"), aff.cpu().numpy())
    if kwargs.get("return_intermediates"):
        return aff.cpu().numpy()
    return AffGraph(aff, pairs)


def computePatchGraph(selected_patches_list, num_selected, selected_patch_pairsIDs, pred_affs,
                      mask_to_cover, patchshape, neighshape, rad, multiple_worms, lookup,
                      consensus_vote_array, **kwargs):
    """aff_patch_graph.py:190-282 (device branch only)."""
    if not kwargs["cuda"]:
        raise RuntimeError("patchperpix_amd only implements the device path (cuda=True)")
    return computePatchGraph_cuda(pred_affs, consensus_vote_array, selected_patch_pairsIDs,
                                  patchshape, neighshape, **kwargs)
