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
    if kwargs.get("sparse_labels", False):
        # (patches handed in as a dict keyed by centre: only stitch_vote_instances does that,
        # stitch_patch_graph.py:380-399 -- served by patchperpix_amd.blockwise.label_graph)
        raise NotImplementedError("sparse_labels is served by patchperpix_amd.blockwise, not here")
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
    dev = pred_affs.device
    per_channel = kwargs.get("one_instance_per_channel", False)
    packed = kwargs.get("no_overlap_per_channel", False)
    if per_channel or packed:
        # graph_to_labeling.py:57-115: every component is painted into a volume of its own;
        # one_instance_per_channel stacks them, no_overlap_per_channel puts a component of more
        # than 2000 voxels into the first channel it does not overlap (a new one if none) and
        # every smaller one into channel 0 (whatever is there)
        nodes_dev = torch.from_numpy(np.ascontiguousarray(nodes)).to(dev)
        labels_dev = torch.from_numpy(labels.astype(np.int32)).to(dev)
        channels = []
        n_comp = int(labels.max()) if len(labels) else 0
        for value in range(1, n_comp + 1):
            own = torch.nonzero(labels_dev == value).reshape(-1)
            cur = torch.zeros(instances.shape, dtype=torch.int32, device=dev)
            if own.numel():
                backend.paint_instances(pred_affs, nodes_dev[own].contiguous(), labels_dev[own].contiguous(), cur, P)
            if per_channel:
                channels.append(cur)
            if packed:
                if not channels:
                    channels.append(cur)
                    continue
                m = cur > 0
                if int(m.sum().item()) > 2000:
                    for ch in channels:
                        if not bool((ch[m] != 0).any().item()):
                            ch[m] = value
                            break
                    else:
                        channels.append(cur)
                else:
                    channels[0][m] = value
        if channels:
            instances = torch.stack(channels, 0).cpu().numpy().astype(out_dtype)
        else:
            # (np.stack of an empty list raises in the reference; a volume without components
            # leaves the early-outs of to_instance_seg before it gets here)
            instances = np.zeros((0,) + tuple(instances.shape), dtype=out_dtype)
    else:
        inst_dev = torch.from_numpy(np.ascontiguousarray(instances).astype(np.int32)).to(dev)
        if len(nodes):
            backend.paint_instances(pred_affs, torch.from_numpy(np.ascontiguousarray(nodes)).to(dev),
                                    torch.from_numpy(labels.astype(np.int32)).to(dev), inst_dev, P)
        instances = inst_dev.cpu().numpy().astype(out_dtype)
    logger.info("done compute labeling")
    if kwargs.get("pad_with_ps", False):
        # (:139-151: the spatial axes are the last three, also of the stacked map)
        sp = instances.shape[-3:]
        sl = tuple(slice(int(rad[i]), sp[i] - int(rad[i])) for i in range(3))
        instances = instances[(Ellipsis,) + sl]
        foreground = foreground[sl]
    return instances, foreground.astype(np.uint8)
