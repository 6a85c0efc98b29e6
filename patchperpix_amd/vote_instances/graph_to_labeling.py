"""Step 6: patch graph -> instance labels
(reference: PatchPerPix/vote_instances/graph_to_labeling.py)."""
import logging

import numpy as np

from .. import backend
from .aff_patch_graph import AffGraph, loadAffgraph
from .graph_mws import mws, mws_from_pairs

logger = logging.getLogger(__name__)


def affGraphToInstancesT(pred_affs, patchshape, rad, debug_output1, debug_output2, instances,
                         foreground, affgraph, selected_patch_pairs, **kwargs):
    affgraph = loadAffgraph(affgraph, selected_patch_pairs)
    return affGraphToInstances(affgraph, pred_affs, patchshape, rad, debug_output1,
                               debug_output2, instances, foreground, **kwargs)


def component_labels(affinity_graph, shape, device, P, **kwargs):
    """Node coordinates (int32 [K, 3]) and their instance ids (component rank + 1).

    ``mws=False``: union-find on the device (ppp_label_components) gives every node the order
    key of its component; ranking the distinct keys (as many numbers as components) happens
    here.  ``mws=True``: mutex watershed on the host (ppp_host_mws)."""
    import torch
    if kwargs["mws"]:
        po = getattr(affinity_graph, "pairs_obj", None)
        if po is not None and getattr(po, "unique_pairs", False) and len(po.nodes):
            # pair list made by this package (no repeated node pair): edge order + |aff| sort on
            # the device (ppp_mws_edges), the sequential loop on the host (ppp_host_mws_sorted)
            nodes_dev = torch.from_numpy(po.nodes).to(device)
            lab, _ = backend.mws_labels_device(po.rows_dev, affinity_graph.aff_dev, nodes_dev, P)
            lab = lab.cpu().numpy()
            return po.nodes[lab > 0], lab[lab > 0].astype(np.int64)
        # anything else (injected / loaded pair lists): csrc/ppp_host_mws.cpp does all of it on
        # the host; graph_mws.mws_from_pairs is the same in Python
        nodes, labels, _ = backend.host_mws(affinity_graph.pairs, affinity_graph.aff, shape)
        return nodes, labels
    nodes = affinity_graph.pairs_obj.nodes
    if len(nodes) == 0:
        return np.zeros((0, 3), np.int32), np.zeros((0,), np.int64)
    nodes_dev = torch.from_numpy(nodes).to(device)
    keys = backend.label_components(affinity_graph.pairs_obj.rows_dev, affinity_graph.aff_dev,
                                    nodes_dev, P).cpu().numpy()
    valid = keys != backend.NONE_KEY
    uniq = np.unique(keys[valid])                       # ascending = networkx's order
    labels = np.searchsorted(uniq, keys[valid]) + 1
    return nodes[valid], labels.astype(np.int64)


def affGraphToInstances(affinity_graph, pred_affs, patchshape, rad, debug_output1,
                        debug_output2, instances, foreground, **kwargs):
    """graph_to_labeling.py:34-155.  ``instances`` gives shape and dtype of the output
    (uint16 single volume, uint32 when stitching); painting happens on the device with
    "largest component id wins", which is what the reference's in-order overwrite yields."""
    import torch
    for opt in ("one_instance_per_channel", "no_overlap_per_channel", "sparse_labels"):
        if kwargs.get(opt, False):
            raise NotImplementedError("%s is not supported" % opt)
    if not isinstance(affinity_graph, AffGraph):   # a networkx graph from outside
        rows = [(tuple(a) + tuple(b), w) for a, b, w in affinity_graph.edges.data("aff")]
        affinity_graph = AffGraph([w for _, w in rows], [r for r, _ in rows],
                                  device=pred_affs.device)
    logger.info("compute labeling")
    P = backend.params_from_kwargs(pred_affs.shape[1:], patchshape, kwargs)
    nodes, labels = component_labels(affinity_graph, instances.shape, pred_affs.device, P,
                                     **kwargs)
    out_dtype = instances.dtype
    if len(labels) and labels.max() > np.iinfo(out_dtype).max:
        raise OverflowError("%d instances do not fit %s" % (labels.max(), out_dtype))
    inst_dev = torch.from_numpy(np.ascontiguousarray(instances).astype(np.int32)).to(pred_affs.device)
    if len(nodes):
        backend.paint_instances(pred_affs,
                                torch.from_numpy(np.ascontiguousarray(nodes)).to(pred_affs.device),
                                torch.from_numpy(labels.astype(np.int32)).to(pred_affs.device),
                                inst_dev, P)
    instances = inst_dev.cpu().numpy().astype(out_dtype)
    logger.info("done compute labeling")
    if kwargs.get("pad_with_ps", False):
        sl = tuple(slice(int(rad[i]), instances.shape[i] - int(rad[i])) for i in range(3))
        instances = instances[sl]
        foreground = foreground[sl]
    return instances, foreground.astype(np.uint8)
