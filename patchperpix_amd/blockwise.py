"""The reference's BLOCKWISE semantics (PatchPerPix/vote_instances/stitch_patch_graph.py): a patch
graph per block, on-disk block graphs, inter-block edges on the face overlaps, one global
labelling.

``stitch_patch_graph.main`` of this package assembles the bounding-boxed volume as a whole by
default (tiled on the device, ``patchperpix_amd.tiling``: same result as the whole-volume path).
With ``blockwise_semantics="reference"`` it runs THIS module instead, which reproduces what the
reference computes -- a different function of the input, because the greedy cover and the
thinning are done per block:

* ``blockwise_vote_instances`` (:553-669): for every block of ``chunksize`` (offsets in raster
  order, :425-440) the prediction of the block + a margin of the patch radius (clipped at the
  volume, no padding) goes through ``to_instance_seg(return_intermediates)``; pair rows are made
  block-relative and margin-free (``patch_pairs -= padded``, :650) and stored with the
  affinities in the result zarr under ``volumes/blocks/<z_y_x>/{patch_pairs, aff_graph_mat}``
  (Blosc zstd bit-shuffle, attribute ``block_shape``); a block whose ``patch_pairs`` exists is
  skipped (resume, :584-587).
* ``stitch_vote_instances`` (:110-399): the blocks' edges enter one graph in block order; for
  every face neighbour that was processed EARLIER, the selected patches within a patch shape of
  the face on either side are paired across the face (L1 <= sum(p + 1), every |delta| <= p + 1,
  not both from the same block), the prediction of their bounding box + patch shape + radius is
  loaded and S1 + S5 run on the injected patches / pairs (``skipRanking``, ``skipThinCover``);
  these inter-block rows are stored under ``volumes/blocks/<off>/<neighbour off>/`` and reused.
* the global graph is partitioned (connected components of the positive edges or the mutex
  watershed, in networkx's insertion order; a repeated edge keeps its first position and takes
  its last value) and painted into a uint32 volume, later components over earlier ones.

Two quirks of the reference are kept because they decide the result: the inter-block region is
located with ``bb_start - margin`` even where the margin was clipped at the volume's low faces
(:307-311 ignore the margin ``load_input`` returns), and the candidate pairs come out of a Python
``set`` whose iteration order is unspecified -- here they are taken in sorted order (the golden
vectors are generated with the same order injected into the reference).
"""
import logging
import os

import numpy as np

from . import backend, minizarr
from .vote_instances import utilVoteInstances as util

logger = logging.getLogger(__name__)

BLOCKS_KEY = "volumes/blocks"
COMPRESSOR = {"id": "blosc", "cname": "zstd", "clevel": 3, "shuffle": 2, "blocksize": 0}
NEIGHBORHOOD = [[-1, 0, 0], [0, -1, 0], [0, 0, -1], [1, 0, 0], [0, 1, 0], [0, 0, 1]]


def get_offset_str(offset):
    return "_".join(str(int(o)) for o in offset)


def get_offsets(total_shape, chunksize):
    """stitch_patch_graph.py:425-440: block origins in raster order."""
    grids = np.meshgrid(*[np.arange(0, total_shape[i], chunksize[i]) for i in range(len(total_shape))],
                        indexing="ij")
    return [np.array(o) for o in np.stack([g.ravel() for g in grids], axis=1)]


class _Volume:
    """The prediction and its companions as (C, Z, Y, X) / (Z, Y, X) array-likes (zarr arrays or
    NumPy arrays): everything the blocks read."""

    def __init__(self, affs, numinst_prob=None, fg=None):
        self.affs, self.numinst_prob, self.fg = affs, numinst_prob, fg
        self.shape = tuple(int(s) for s in affs.shape[1:])

    def box(self, arr, lo, hi):
        sl = tuple(slice(int(a), int(b)) for a, b in zip(lo, hi))
        if arr.ndim == 4:
            return np.asarray(arr[(slice(None),) + sl])
        return np.asarray(arr[sl])


def _load_block(vol, offset, size, margin, **kwargs):
    """load_input(..., padding=False) of affinities, numinst and foreground for the block at
    `offset` (global) of `size`, grown by `margin` where the volume allows.  Returns (block f32
    (C, ...), foreground bool, numinst u8, padded = margin actually present on the low sides,
    lo = global coordinate of the block's first voxel)."""
    offset, size, margin = np.asarray(offset), np.asarray(size), np.asarray(margin)
    lo = np.maximum(offset - margin, 0)
    hi = np.minimum(offset + size + margin, vol.shape)
    padded = np.where(offset - margin < 0, 0, margin)
    block = vol.box(vol.affs, lo, hi).astype(np.float32)
    numinst = None
    if vol.numinst_prob is not None:
        prob = vol.box(vol.numinst_prob, lo, hi)
        numinst = util._numinst_from_prob(prob, **kwargs)
    fg = vol.box(vol.fg, lo, hi) if (vol.fg is not None and numinst is None) else None
    foreground = np.asarray(util.returnFg(block, numinst, fg, **kwargs)).astype(bool)
    if numinst is None:
        numinst = foreground.astype(np.uint8)
    return block, foreground, np.asarray(numinst), padded, lo


def _do_block(block, foreground, numinst, patchshape, kw, **extra):
    """vote_instances.do_block with return_intermediates (vote_instances.py:455-483)."""
    from .vote_instances.vote_instances import to_instance_seg
    k = dict(kw, return_intermediates=True, blockwise=False, **extra)
    k.pop("patchshape", None)
    return to_instance_seg(block, foreground, foreground.copy(), numinst, patchshape, **k)


def blockwise_vote_instances(vol, store, offset, bb_offset, bb_shape, kwargs):
    """One block (stitch_patch_graph.py:553-669).  `offset` is relative to the bounding box."""
    patchshape = np.asarray(kwargs["patchshape"])
    chunksize = np.minimum(np.asarray(kwargs["chunksize"]), bb_shape)
    margin = patchshape // 2
    in_offset = np.asarray(offset) + np.asarray(bb_offset)
    key = BLOCKS_KEY + "/" + get_offset_str(in_offset)
    if key + "/patch_pairs" in store:
        logger.info("%s already processed.", key)
        return
    block, foreground, numinst, padded, _lo = _load_block(vol, in_offset, chunksize, margin, **kwargs)
    pairs, aff = _do_block(block, foreground, numinst, patchshape, kwargs)
    if pairs is None:
        return
    pairs = (pairs.astype(np.int64) - np.array(list(padded) * 2)).astype(np.uint32)
    ds = store.create_dataset(key + "/patch_pairs", data=pairs, chunks=(max(1, len(pairs)), 6),
                              compressor=COMPRESSOR)
    ds.attrs["block_shape"] = [int(v) for v in block.shape[1:]]
    store.create_dataset(key + "/aff_graph_mat", data=np.asarray(aff, dtype=np.float32),
                         chunks=(max(1, len(aff)),), compressor=COMPRESSOR)


def cross_face_pairs(current, neighbour, patchshape):
    """Index pairs (i < j) into concatenate([current, neighbour]) of patches that may share an
    edge across the face: L1 distance <= sum(p + 1) (cKDTree.query_pairs, :239-241), every
    |delta| <= p + 1 (remove_pairs, :73-88), not both listed by the same block
    (remove_intra_block_pairs, :91-107).  Sorted -- the reference iterates a set."""
    from scipy import spatial
    cand = np.concatenate([current, neighbour])
    p1 = np.asarray(patchshape) + 1
    raw = spatial.cKDTree(cand, leafsize=4).query_pairs(float(np.sum(p1)), p=1)
    if not raw:
        return cand, np.zeros((0, 2), dtype=np.int64)
    pr = np.array(sorted(raw), dtype=np.int64)
    d = np.abs(cand[pr[:, 0]].astype(np.float32) - cand[pr[:, 1]].astype(np.float32))
    pr = pr[~np.any(d > p1, axis=1)]

    def member(points, block):
        keys = {tuple(int(v) for v in q) for q in block}
        return np.array([tuple(int(v) for v in q) in keys for q in points], dtype=bool)
    in_cur, in_nb = member(cand, current), member(cand, neighbour)
    same = (in_cur[pr[:, 0]] & in_cur[pr[:, 1]]) | (in_nb[pr[:, 0]] & in_nb[pr[:, 1]])
    return cand, pr[~same]


def stitch_vote_instances(vol, store, output_shape, bb_offset, bb_shape, kwargs):
    """stitch_patch_graph.py:110-399.  Returns (rows uint32 [n, 6] global, aff float32 [n]) of the
    global graph in insertion order (a repeated node pair may occur: first position, last
    value)."""
    chunksize = np.asarray(kwargs["chunksize"])
    patchshape = np.asarray(kwargs["patchshape"])
    offsets = [off + np.asarray(bb_offset) for off in get_offsets(bb_shape, chunksize)]
    rows_all, aff_all = [], []
    global_patches = []
    seen_any = False
    for block_id, offset in enumerate(offsets):
        key = BLOCKS_KEY + "/" + get_offset_str(offset)
        if key + "/patch_pairs" not in store:
            global_patches.append(None)          # the block was empty
            continue
        pairs = np.asarray(store[key + "/patch_pairs"][...]).astype(np.int64) + np.array(list(offset) * 2)
        aff = np.asarray(store[key + "/aff_graph_mat"][...])
        selected = np.unique(pairs.reshape(-1, 3), axis=0)
        global_patches.append(selected)
        rows_all.append(pairs)
        aff_all.append(aff)
        if not seen_any:
            seen_any = True                      # (:176-179: the first block only opens the graph)
            continue
        for neighbor in NEIGHBORHOOD:
            nb_off = offset + np.asarray(neighbor) * chunksize
            for nb_id in range(block_id):
                if not np.all(offsets[nb_id] == nb_off):
                    continue
                ikey = key + "/" + get_offset_str(nb_off)
                if ikey in store:
                    rows_all.append(np.asarray(store[ikey + "/patch_pairs"][...]).astype(np.int64))
                    aff_all.append(np.asarray(store[ikey + "/aff_graph_mat"][...]))
                    continue
                if global_patches[nb_id] is None:
                    continue
                nb_patches = global_patches[nb_id]
                dim = int(np.argmax(np.asarray(neighbor) != 0))
                if neighbor[dim] > 0:
                    cur = selected[selected[:, dim] >= offset[dim] - patchshape[dim]]
                    nbc = nb_patches[nb_patches[:, dim] <= offset[dim] + patchshape[dim]]
                else:
                    cur = selected[selected[:, dim] <= offset[dim] + patchshape[dim]]
                    nbc = nb_patches[nb_patches[:, dim] >= offset[dim] - patchshape[dim]]
                if len(cur) == 0 or len(nbc) == 0:
                    continue
                cand, pr = cross_face_pairs(cur, nbc, patchshape)
                if len(pr) == 0:
                    continue
                cleaned = cand[np.unique(pr.reshape(-1))]
                bb_start = np.maximum(cleaned.min(axis=0) - patchshape, 0)
                bb_stop = np.minimum(cleaned.max(axis=0) + patchshape, output_shape)
                margin = patchshape // 2
                bb_size = np.maximum(bb_stop - bb_start, 1)
                block, foreground, numinst, _padded, _lo = _load_block(vol, bb_start, bb_size, margin, **kwargs)
                overlapping = np.concatenate([cand[pr[:, 0]], cand[pr[:, 1]]], axis=1).astype(np.int64)
                # (:307-311: relative to bb_start - margin, also where the margin was clipped)
                shift = bb_start - margin
                cand_rel = cleaned - shift
                pairs_rel = overlapping - np.array(list(shift) * 2)
                if (cand_rel < 0).any() or (cand_rel >= np.array(block.shape[1:])).any():
                    raise ValueError("inter-block candidates fall outside the loaded region "
                                     "(the reference would index out of bounds here)")
                _p, a = _do_block(block, foreground, numinst, patchshape, kwargs,
                                  selected_patches=cand_rel, selected_patch_pairs=pairs_rel.astype(np.uint32),
                                  skipRanking=True, skipThinCover=True)
                a = np.asarray(a, dtype=np.float32)
                rows_all.append(overlapping)
                aff_all.append(a)
                ds = store.create_dataset(ikey + "/patch_pairs", data=overlapping.astype(np.uint32),
                                          chunks=(max(1, len(overlapping)), 6), compressor=COMPRESSOR)
                ds.attrs["block_shape"] = [int(v) for v in block.shape[1:]]
                store.create_dataset(ikey + "/aff_graph_mat", data=a, chunks=(max(1, len(a)),),
                                     compressor=COMPRESSOR)
    if not rows_all:
        return np.zeros((0, 6), np.uint32), np.zeros((0,), np.float32)
    return np.concatenate(rows_all).astype(np.uint32), np.concatenate(aff_all).astype(np.float32)


def dedupe_edges(rows, aff, shape):
    """networkx semantics of add_edge on a repeated (unordered) node pair: the edge keeps its
    first position and takes its last value (update_graph, :58-66).  Rows with aff == 0 never
    enter the graph (and do not overwrite)."""
    live = aff != 0
    rows, aff = rows[live], aff[live]
    if len(rows) == 0:
        return rows, aff
    Y, X = int(shape[1]), int(shape[2])
    r = rows.astype(np.int64)
    a = (r[:, 0] * Y + r[:, 1]) * X + r[:, 2]
    b = (r[:, 3] * Y + r[:, 4]) * X + r[:, 5]
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    V = int(shape[0]) * Y * X
    key = lo * V + hi
    _, first = np.unique(key, return_index=True)
    _, last_rev = np.unique(key[::-1], return_index=True)
    last = len(key) - 1 - last_rev
    order = np.argsort(first)                    # edges in order of first appearance
    return rows[first[order]], aff[last[order]]


def label_graph(vol, rows, aff, shape, kwargs):
    """affGraphToInstances(sparse_labels=True) on the global graph into a uint32 volume
    (:388-396; graph_to_labeling.py:34-86): later components overwrite earlier ones."""
    import torch
    instances = np.zeros(shape, dtype=np.uint32)
    rows, aff = dedupe_edges(rows, aff, shape)
    if len(rows) == 0:
        return instances
    patchshape = [int(p) for p in kwargs["patchshape"]]
    kwargs = {k: v for k, v in kwargs.items() if k != "patchshape"}
    if kwargs.get("mws"):
        nodes, labels, _n = backend.host_mws(rows, aff, shape)
    else:
        from .vote_instances.aff_patch_graph import PatchPairs
        dev = torch.device("cuda")
        P = backend.params_from_kwargs(shape, patchshape, kwargs)
        nodes = np.unique(rows.reshape(-1, 3), axis=0).astype(np.int32)
        keys = backend.label_components(torch.from_numpy(rows.view(np.int32)).to(dev),
                                        torch.from_numpy(aff).to(dev),
                                        torch.from_numpy(nodes).to(dev), P).cpu().numpy()
        valid = keys != backend.NONE_KEY
        uniq = np.unique(keys[valid])
        nodes, labels = nodes[valid], np.searchsorted(uniq, keys[valid]) + 1
    if len(nodes) == 0:
        return instances
    # paint: the patch of every node is read from the prediction (sparse_labels).  The patches
    # are gathered into a TABLE (n, C) -- one read per touched chunk of the store, never a dense
    # (C, Z, Y, X) block -- and painted from the table (ppp_paint_patch_rows); "largest id wins"
    # = the reference's in-order overwrite
    dev = torch.device("cuda")
    nodes = np.asarray(nodes, dtype=np.int64)
    labels = np.asarray(labels, dtype=np.int32)
    inst_dev = torch.zeros(shape, dtype=torch.int32, device=dev)
    P = backend.params_from_kwargs(shape, patchshape, kwargs)
    # batches bounded by table bytes (256 MB of float32 rows), nodes grouped by chunk inside
    C = int(np.prod(patchshape))
    per = max(1, (256 << 20) // (4 * C))
    order = _chunk_order(vol.affs, nodes)
    for s0 in range(0, len(nodes), per):
        idx = order[s0:s0 + per]
        rows = gather_patch_rows(vol.affs, nodes[idx])
        backend.paint_patch_rows(torch.from_numpy(rows).to(dev),
                                 torch.from_numpy(nodes[idx].astype(np.int32)).to(dev),
                                 torch.from_numpy(labels[idx]).to(dev), inst_dev, P)
    return inst_dev.cpu().numpy().view(np.uint32)


def _chunk_grid(affs):
    """spatial chunk shape of the prediction store (one chunk = everything for an ndarray)"""
    ch = getattr(affs, "chunks", None)
    if ch is None or len(ch) != 4:
        return tuple(int(v) for v in affs.shape[1:])
    return tuple(max(1, int(c)) for c in ch[1:])


def _chunk_order(affs, nodes):
    """node indices sorted by the chunk that holds the node (then z, y, x)"""
    cz, cy, cx = _chunk_grid(affs)
    key = np.stack([nodes[:, 2], nodes[:, 1], nodes[:, 0], nodes[:, 2] // cx, nodes[:, 1] // cy, nodes[:, 0] // cz])
    return np.lexsort(key)


def gather_patch_rows(affs, nodes):
    """(n, C) float32 table: row k = affs[:, node k].  Every chunk that holds a node is read ONCE,
    and only the bounding box of its nodes."""
    cz, cy, cx = _chunk_grid(affs)
    n = len(nodes)
    out = np.empty((n, int(affs.shape[0])), dtype=np.float32)
    cid = (nodes[:, 0] // cz, nodes[:, 1] // cy, nodes[:, 2] // cx)
    key = (cid[0] * (1 << 40)) + (cid[1] * (1 << 20)) + cid[2]
    order = np.argsort(key, kind="stable")
    bounds = np.flatnonzero(np.diff(key[order])) + 1
    for grp in np.split(order, bounds):
        nd = nodes[grp]
        lo, hi = nd.min(axis=0), nd.max(axis=0) + 1
        block = np.asarray(affs[(slice(None),) + tuple(slice(int(a), int(b)) for a, b in zip(lo, hi))])
        rel = nd - lo
        out[grp] = block[:, rel[:, 0], rel[:, 1], rel[:, 2]].T
    return out


def main(pred_file, result_folder=".", **kwargs):
    """stitch_patch_graph.main (:672-894) with the reference's per-block semantics.  Returns the
    uint32 instance volume (before the uint16 cast of the written datasets)."""
    from scipy import ndimage  # noqa: F401  (clean_mask)
    from .vote_instances.stitch_patch_graph import clean_mask
    from .vote_instances.vote_instances import write_result, _skeletonize
    from . import postprocess
    kwargs = dict(kwargs)
    kwargs["return_intermediates"] = True
    sample = os.path.basename(pred_file.rstrip("/")).split(".")[0]
    result_file = os.path.join(result_folder, sample + ".zarr")
    os.makedirs(result_folder, exist_ok=True)
    aff_key = kwargs.get("aff_key", "volumes/pred_affs")
    numinst_key = kwargs.get("numinst_key")
    res_key = kwargs.get("res_key", "vote_instances")
    if not pred_file.rstrip("/").endswith(".zarr"):
        raise NotImplementedError("the blockwise driver reads zarr predictions (stitch_patch_graph.py:712-715)")
    in_f = minizarr.open(pred_file, "r")
    affs = in_f[aff_key]
    vol = _Volume(affs, in_f[numinst_key] if numinst_key is not None else None,
                  in_f[kwargs["fg_key"]] if (numinst_key is None and kwargs.get("fg_key") is not None) else None)
    input_shape = vol.shape
    if kwargs.get("only_bb"):
        mask, _key = util.loadFg(in_f, **kwargs)
        mask = np.squeeze(mask)
        if np.count_nonzero(mask) == 0:
            logger.info("Volume has no foreground voxel, returning...")
            return None
        if kwargs.get("ignore_small_comps", 0) > 0:
            mask = clean_mask(mask, np.ones([3] * mask.ndim), kwargs["ignore_small_comps"]).astype(np.uint8)
        if kwargs.get("skeletonize_foreground"):
            mask = _skeletonize(mask, kwargs.get("skeletonize_backend")).astype(np.uint8)
        nz = np.transpose(np.nonzero(mask))
        bb_offset, shape = nz.min(axis=0), nz.max(axis=0) - nz.min(axis=0) + 1
    else:
        shape, bb_offset = np.array(input_shape), np.zeros(3, dtype=np.int64)
    store = minizarr.open(result_file, "a" if os.path.exists(result_file) else "w")
    if not kwargs.get("graphToInst", False):
        for offset in get_offsets(shape, kwargs["chunksize"]):
            blockwise_vote_instances(vol, store, offset, bb_offset, shape, kwargs)
    rows, aff = stitch_vote_instances(vol, store, np.array(input_shape), bb_offset, shape, kwargs)
    instances = label_graph(vol, rows, aff, input_shape, kwargs)
    foreground, _ = util.loadFg(in_f, **kwargs)
    foreground = np.squeeze(foreground)
    if kwargs.get("remove_small_comps", 0) > 0:
        instances = postprocess.relabel(postprocess.remove_small_components(instances, kwargs["remove_small_comps"]))
    if kwargs.get("output_format", "hdf") == "hdf":
        masked = instances.copy()
        masked[foreground == 0] = 0
        datasets = {res_key: instances.astype(np.uint16), "vote_foreground": np.asarray(foreground).astype(np.uint16),
                    res_key + "_masked": masked.astype(np.uint16)}
        if kwargs.get("dilate_instances", False):
            dil = postprocess.dilate_instances(instances)
            datasets[res_key + "_dil_1"] = dil.astype(np.uint16)
            datasets[res_key + "_masked_dil_1"] = np.where(foreground == 0, 0, dil).astype(np.uint16)
        write_result(os.path.join(result_folder, sample + ".hdf"), datasets)
    return instances
