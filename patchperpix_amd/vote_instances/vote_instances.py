"""Instance assembly orchestration (reference: PatchPerPix/vote_instances/vote_instances.py).

Same entry points and keyword flags as the reference -- ``main`` (:557), ``do_all`` (:486),
``do_block`` (:455), ``to_instance_seg`` (:150), ``get_arguments`` (:62) -- driving the
gfx950 kernels of libppp_mi355x.so instead of pycuda-JITed CUDA.  Only the device path
(``cuda=True``) exists; asking for the NumPy path raises instead of silently falling back.
"""
import argparse
import glob
import logging
import os

import numpy as np

from .. import backend
from .aff_patch_graph import (PatchPairs, computeAndStorePatchPairs, computePatchGraph,
                              loadAffgraph, setAffgraph)
from .consensus_array import loadOrComputeConsensus
from .cuda_code import delete_cuda, init_cuda
from .foreground_cover import computeForegroundCover, thinOutForegroundCover
from .graph_to_labeling import affGraphToInstances, affGraphToInstancesT
from .ranked_patches import PatchList, loadOrComputePatchRanking
from .utilVoteInstances import getResKey, loadAffinities
from . import io_hdflike

logger = logging.getLogger(__name__)


def replace(array, old_values, new_values):
    """Relabel through a lookup table over the value range of `array` (vote_instances.py:41-46)."""
    new_values = np.asarray(new_values)
    lut = np.arange(int(np.max(array)) + 1).astype(new_values.dtype)
    np.put(lut, np.asarray(old_values, dtype=np.int64), new_values)
    return np.take(lut, array)


def merge_dicts(sink, source):
    """Recursive update of `sink` with `source` (nested dicts are merged, everything else is
    overwritten); returns `sink` (vote_instances.py:49-58)."""
    if not (isinstance(sink, dict) and isinstance(source, dict)):
        raise TypeError('Args to merge_dicts should be dicts')
    for key, value in source.items():
        both_dicts = isinstance(value, dict) and isinstance(sink.get(key), dict)
        sink[key] = merge_dicts(sink[key], value) if both_dicts else value
    return sink


def get_arguments(check_required=True, argv=None):
    """Command line of the reference (vote_instances.py:62-147), same names and defaults."""
    parser = argparse.ArgumentParser()
    parser.add_argument('--affinities', type=str, required=False)
    parser.add_argument('--affinities_key', type=str, required=False, default="images/pred_affs")
    parser.add_argument('--basedir', type=str, required=False)
    parser.add_argument('--mode', type=str, required=False)
    parser.add_argument('--result_folder', type=str, required=check_required)
    parser.add_argument('--checkpoint', type=str, required=False)
    parser.add_argument("--debug", action="store_true")
    parser.add_argument('--patch_threshold', type=float, default=0.9)
    parser.add_argument('--fc_threshold', type=float, default=0.5)
    parser.add_argument('-p', '--patchshape', type=int, action='append', required=check_required)
    parser.add_argument('--consensus', type=str)
    parser.add_argument('--consensus_key', type=str, required=False, default="images/consensus")
    parser.add_argument('--scores', type=str)
    parser.add_argument('--ranked_patches', type=str)
    parser.add_argument('--aff_graph', type=str, required=False)
    parser.add_argument('--selected_patches', type=str, required=False)
    parser.add_argument('--selected_patch_pairs', type=str, required=False)
    for flag in ("select_patches_for_sparse_data", "cuda", "skipLookup", "skipThinCover",
                 "skipRanking", "skipConsensus", "termAfterThinCover", "graphToInst", "mws",
                 "includeSinglePatchCCS", "removeIntersection", "isbiHack", "mask_fg_border",
                 "parallel", "save_no_intermediates"):
        parser.add_argument("--" + flag, action="store_true")
    for ax in "xyz":
        parser.add_argument('--crop_%s_s' % ax, type=int, default=0)
        parser.add_argument('--crop_%s_e' % ax, type=int, default=None)
    args, _ = parser.parse_known_args(argv)
    return args


SKELETONIZE_SERVED_BY = None     # which implementation thinned the last mask (written to the result's attrs)


def _skeletonize(mask, backend_name=None):
    """skimage.morphology.skeletonize_3d of the mask (vote_instances.py:219-224,
    stitch_patch_graph.py:757-760).  With scikit-image importable that is what runs.  Without it
    the option RAISES, as the reference's import would -- unless the caller opts in to the
    library's own thinning with ``skeletonize_backend="ppp"`` (kwarg / TOML entry, or
    PPP_SKELETONIZE=ppp): ``ppp_host_skeletonize_3d`` restates the same published algorithm (Lee /
    Kashyap / Chu 1994; csrc/ppp_host_skel.cpp) but is PARITY UNPINNED against scikit-image, which
    this image lacks, and deliberately deviates in its sequential re-check (all three deletion
    conditions instead of the simple-point test alone, which erases plates two voxels thick) --
    so the cover mask / bounding box may differ from a scikit-image run.  The choice is logged at
    WARNING and recorded in ``SKELETONIZE_SERVED_BY`` (the drivers copy it into the output attrs)."""
    global SKELETONIZE_SERVED_BY
    import os
    choice = backend_name or os.environ.get("PPP_SKELETONIZE") or "skimage"
    if choice not in ("skimage", "ppp"):
        raise ValueError("skeletonize_backend must be 'skimage' or 'ppp', not %r" % (choice,))
    if choice == "skimage":
        try:
            from skimage.morphology import skeletonize_3d
        except ImportError as e:
            raise ImportError(
                "skeletonize_foreground needs scikit-image (skimage.morphology.skeletonize_3d); pass "
                "skeletonize_backend='ppp' (or PPP_SKELETONIZE=ppp) to use the library's own thinning, "
                "whose result is not pinned to scikit-image's") from e
        SKELETONIZE_SERVED_BY = "skimage.morphology.skeletonize_3d"
        return skeletonize_3d(mask) > 0
    logger.warning("skeletonize_foreground: thinning with ppp_host_skeletonize_3d (skeletonize_backend='ppp'); "
                   "not pinned to scikit-image's skeletonize_3d -- the cover mask / bounding box may differ")
    SKELETONIZE_SERVED_BY = "ppp_host_skeletonize_3d"
    return backend.host_skeletonize_3d(mask)


def _pad(a, rad, channels=False):
    width = [(int(r), int(r)) for r in rad]
    if channels:
        width = [(0, 0)] + width
    return np.pad(a, width, mode='constant')


def to_instance_seg(pred_affs, foreground, mask_to_cover, numinst, patchshape, **kwargs):
    """vote_instances.py:150-452.

    pred_affs     (C,Z,Y,X) float32/float16 ndarray, or an already-resident device tensor
    foreground    (Z,Y,X) bool       mask_to_cover (Z,Y,X) bool, modified in place (:226)
    numinst       (Z,Y,X) integer    patchshape    int[3]
    (the three fields may also be device tensors, resident like the prediction)
    Returns (instances uint16 (Z,Y,X), foreground uint8) -- or (pairs uint32 [N,6],
    aff float32 [N]) with ``return_intermediates`` -- with the reference's early-outs.
    """
    import torch
    # cuda=False selects the reference's NumPy SEMANTICS (int16 votes, integer ranks, all-pairs
    # graph weights: a different function, numpy_semantics.py) -- computed on the device as well;
    # nothing in this package falls back to the CPU
    numpy_path = not kwargs.get('cuda', False)
    for opt in ("debug", "isbiHack"):
        if kwargs.get(opt, False):
            raise NotImplementedError("%s is not supported" % opt)
    patchshape = np.array([int(p) for p in patchshape])
    rad = np.array([p // 2 for p in patchshape])
    if int(patchshape[0]) == 1 and int(np.shape(foreground)[0]) > 1:
        # 2-d patches are for 2-d data, which the reference carries with a Z axis of 1
        # (utilVoteInstances.py:187).  On a stack of slices its pair enumeration links patches of
        # neighbouring slices (|dz| <= 2 p_z = 2, aff_patch_graph.py:61-69) and computePatchGraph
        # then indexes the consensus array at zo = |dz| < 2 * PSZ although it has NSZ = 1 plane
        # (computePatchGraph.cu:98-105): an out-of-bounds read, i.e. no defined result to match.
        raise ValueError("patch shape %s with %d slices: 2-d patches need 2-d data (Z = 1); call once per "
                         "slice" % (patchshape.tolist(), int(np.shape(foreground)[0])))

    if kwargs.get("pad_with_ps", False):
        assert not kwargs.get('blockwise'), "can only pad whole volumes"
        if torch.is_tensor(pred_affs):
            pred_affs = torch.nn.functional.pad(
                pred_affs, (int(rad[2]),) * 2 + (int(rad[1]),) * 2 + (int(rad[0]),) * 2)
        else:
            pred_affs = _pad(pred_affs, rad, channels=True)
        foreground = _pad(foreground, rad)
        mask_to_cover = _pad(mask_to_cover, rad)
        numinst = _pad(numinst, rad)

    pred_affs = backend.to_device_pred(pred_affs)   # host->HBM once; f16 stays f16 (exact)
    shape = tuple(foreground.shape)
    # vote_instances.py:219-224: the mask is thinned BEFORE anything else looks at it, so this
    # happens ahead of the dispatch to the tiled path (which must cover the same mask)
    if not kwargs.get('blockwise', False) and kwargs.get('skeletonize_foreground'):
        mask_to_cover = _skeletonize(mask_to_cover, kwargs.get("skeletonize_backend"))
    if numpy_path:
        from . import numpy_semantics
        host = lambda a: a.cpu().numpy() if torch.is_tensor(a) else np.asarray(a)      # noqa: E731
        inst, fg = numpy_semantics.to_instance_seg(pred_affs, host(foreground), host(mask_to_cover),
                                                   host(numinst), patchshape, **kwargs)
        if kwargs.get("pad_with_ps", False):
            sl = tuple(slice(int(rad[i]), inst.shape[i] - int(rad[i])) for i in range(3))
            inst, fg = inst[sl], fg[sl]
        return inst, fg
    # Volumes whose consensus does not fit in HBM are assembled slab by slab (identical result,
    # patchperpix_amd/tiling.py); `_n_slabs` forces a slab count.
    n_slabs = kwargs.get("_n_slabs")
    yx_tiles = kwargs.get("_yx_tiles")
    if n_slabs is None and not kwargs.get("save_consensus", False):
        from .. import tiling
        # free HBM + what the caching allocator holds but has not handed out
        avail = torch.cuda.mem_get_info()[0] + \
            (torch.cuda.memory_reserved() - torch.cuda.memory_allocated())
        # working set per tile: voxel-major rows only when S1 can write them directly
        Pq = backend.params_from_kwargs(shape, patchshape, kwargs)
        Pq.cons_layout = backend.CONS_VOXEL_MAJOR
        direct = os.environ.get("PPP_S1_DIRECT_VM", "1") != "0" and \
            backend.lib().ppp_consensus_writes_voxel_major(Pq) == 1
        # the pooled consensus buffer is held through the global stage, whose other buffers
        # (ranked lists, cover / sort work space: ~70 bytes per voxel) and the pair rows of the
        # patch-graph stage must fit next to it
        reserve = 70.0 * float(np.prod(shape)) + 4e9
        # (a consensus cache -- every base voxel computed once, tiling.assemble -- is taken when its
        # planes fit next to the rows of one tile; only the packed S1 kernel fills one)
        n_slabs, ny_t, nx_t, use_cache = tiling.plan_tiles(
            shape, patchshape, max(avail - reserve, 0.25 * avail),
            safety=float(os.environ.get("PPP_TILE_SAFETY", "0.92")), copies=2.0 if direct else 3.0,
            cache_shape=shape if direct and torch.is_tensor(pred_affs) else None)
        kwargs.setdefault("_cons_cache", use_cache)
        if not use_cache and direct and torch.is_tensor(pred_affs) and "_ring_z" not in kwargs:
            # neither whole nor cached: sweep the columns of tiles bottom-up with the rows in a ring
            # (no z-halo in either S1 pass) where that is less work than the plain grid
            ring = tiling.plan_ring(shape, patchshape, max(avail - reserve, 0.25 * avail),
                                    safety=float(os.environ.get("PPP_TILE_SAFETY", "0.92")), copies=2.0)
            if ring is not None:
                n_slabs, ny_t, nx_t, kwargs["_ring_z"] = ring
        if yx_tiles is None and (ny_t > 1 or nx_t > 1):
            yx_tiles = (ny_t, nx_t)
        # what was decided, and on what: read by bench.py (`config.plan`) so that a silent change of
        # plan -- less free HBM: thinner tiles, a shorter ring -- shows next to the number it changes
        backend.LAST_PLAN = {"tiles": [int(n_slabs), int(ny_t), int(nx_t)], "ring_z": int(kwargs.get("_ring_z") or 0),
                             "cons_cache": bool(kwargs.get("_cons_cache")), "s1_writes_voxel_major": bool(direct),
                             "free_hbm_gb_at_plan_time": round(avail / 1e9, 2), "reserve_gb": round(reserve / 1e9, 2),
                             "budget_gb": round(max(avail - reserve, 0.25 * avail) / 1e9, 2)}
    # With nothing to store or load between the stages, the single-slab case takes the same
    # code path: it keeps the ranked patch list on the device instead of materialising the
    # reference's host lists between the stage functions (PPP_PIPELINE=stages keeps them).
    resume = any(kwargs.get(k) is not None and os.path.exists(str(kwargs.get(k)))
                 for k in ("consensus", "ranked_patches"))     # stored stages: stage path
    plain = kwargs.get("save_no_intermediates", False) and not kwargs.get("debug", False) and not resume \
        and not any(kwargs.get(k) for k in ("skipConsensus", "skipRanking",
                                            "termAfterThinCover", "termAfterPatchGraph",
                                            "save_consensus", "blockwise",
                                            "one_instance_per_channel", "no_overlap_per_channel")) \
        and os.environ.get("PPP_PIPELINE", "fused") != "stages"
    if n_slabs and (n_slabs > 1 or yx_tiles or plain) and not kwargs.get("graphToInst") \
            and kwargs.get("aff_graph") is None and not kwargs.get("pad_with_ps", False):
        from .. import tiling
        logger.info("assembling in %d z-slabs", n_slabs)
        kw = {k: v for k, v in kwargs.items() if k not in ("_n_slabs", "_yx_tiles")}
        if yx_tiles:
            kw["_yx_tiles"] = tuple(int(v) for v in yx_tiles)
        return tiling.assemble(pred_affs, 0, shape, foreground, mask_to_cover, numinst,
                               patchshape, tiling.plan_slabs(shape[0], n_slabs), **kw)
    # the stage-by-stage path below keeps the reference's host arrays
    if torch.is_tensor(foreground):
        foreground = foreground.cpu().numpy().astype(bool)
    if torch.is_tensor(mask_to_cover):
        mask_to_cover = mask_to_cover.cpu().numpy().astype(bool)
    if torch.is_tensor(numinst):
        numinst = numinst.cpu().numpy()
    radslice = tuple(slice(int(rad[i]), shape[i] - int(rad[i])) for i in range(3))
    overlap_mask = 1 * (numinst > 1)

    mask_to_cover[overlap_mask > 0] = 0
    # uint16 ids (vote_instances.py:230); the blockwise driver carries uint32
    # (stitch_patch_graph.py:120) and asks for it with _instances_dtype
    id_dtype = np.dtype(kwargs.get("_instances_dtype") or np.uint16)
    instances = np.zeros(shape, dtype=id_dtype)

    def unpadded(inst, fg):
        if kwargs.get("pad_with_ps", False):
            sl = tuple(slice(int(rad[i]), inst.shape[i] - int(rad[i])) for i in range(3))
            return inst[sl], fg[sl]
        return inst, fg

    if np.count_nonzero(mask_to_cover[radslice]) == 0:
        logger.info("no fg found, returning...")
        if kwargs.get('return_intermediates', False):
            return None, None
        inst, fg = unpadded(instances, foreground)
        return inst.astype(id_dtype), fg.astype(np.uint8)

    neighshape = patchshape.copy()
    if neighshape[0] > 1:
        neighshape *= 2
    else:
        neighshape[1:] *= 2

    if kwargs.get('graphToInst'):
        fn = os.path.splitext(os.path.basename(kwargs['affinities']))[0]
        kwargs['affgraph'] = os.path.join(kwargs['result_folder'], fn + "_aff_graph.npy")
        kwargs['selected_patch_pairs'] = os.path.join(kwargs['result_folder'],
                                                      fn + "_selected_patch_pairs.npy")
        return affGraphToInstancesT(pred_affs, patchshape, rad, None, None, instances,
                                    foreground, **kwargs)

    if np.count_nonzero(foreground[radslice]) == 0:
        logger.info("no patches found, returning...")
        if kwargs.get('return_intermediates', False):
            return None, None
        return instances.astype(id_dtype), foreground.astype(np.uint8)

    # (1) consensus
    if not kwargs.get('skipConsensus'):
        with backend.host_timer("s1_consensus"):
            consensus_vote_array, _, _ = loadOrComputeConsensus(
                instances, patchshape, neighshape, None, pred_affs, rad, foreground, None,
                overlap_mask, **kwargs)
    else:
        consensus_vote_array = None
    if kwargs.get('save_consensus', False):
        return None, None

    # (2) ranking
    ranked_patches_list, scores_array = None, None
    if not kwargs.get('skipRanking'):
        with backend.host_timer("s2_rank_and_sort"):
            ranked_patches_list, scores_array = loadOrComputePatchRanking(
                pred_affs=pred_affs, consensus_vote_array=consensus_vote_array,
                overlap_mask=overlap_mask, all_patches=None, patchshape=patchshape,
                neighshape=neighshape, rad=rad, _foreground=foreground, **kwargs)
        logger.info("num ranked patches %s ", len(ranked_patches_list))

    if kwargs.get('aff_graph') is None:
        if kwargs.get('selected_patches') is not None:
            coords = np.array(list(kwargs.get('selected_patches'))).reshape(-1, 3)
            selected_patches_list = PatchList(coords, np.ones(len(coords), np.float32))
            num_selected = len(selected_patches_list)
        elif kwargs.get('skipSelection', False):
            selected_patches_list = ranked_patches_list
            num_selected = len(ranked_patches_list)
        else:
            # (3) greedy cover
            with backend.host_timer("s3_cover"):
                selected_patches_list, num_selected = computeForegroundCover(
                    overlap_mask, mask_to_cover, patchshape, ranked_patches_list, radslice,
                    pred_affs, rad, None, scores_array, **kwargs)
        # (4) thinning
        if not kwargs.get('skipThinCover') and num_selected > 0:
            with backend.host_timer("s4_thin"):
                selected_patches_list, num_selected = thinOutForegroundCover(
                    mask_to_cover, selected_patches_list, radslice, pred_affs, rad,
                    patchshape, **kwargs)

        if kwargs.get('selected_patch_pairs') is not None:
            selected_patch_pairsIDs = PatchPairs.from_host(
                np.array(kwargs.get('selected_patch_pairs'), dtype=np.uint32).reshape(-1, 6),
                pred_affs.device)
            if len(selected_patch_pairsIDs) == 0:
                selected_patch_pairsIDs = None
        else:
            with backend.host_timer("pairs"):
                selected_patch_pairsIDs = computeAndStorePatchPairs(
                    selected_patches_list, patchshape, _volume_shape=shape,
                    _device=pred_affs.device, **kwargs)
        if selected_patch_pairsIDs is None:
            if kwargs.get('return_intermediates', False):
                return None, None
            return instances.astype(id_dtype), foreground.astype(np.uint8)
        if kwargs.get('termAfterThinCover'):
            raise SystemExit(0)

        # (5) patch graph
        with backend.host_timer("s5_patch_graph"):
            affinity_graph = computePatchGraph(
                selected_patches_list, num_selected, selected_patch_pairsIDs, pred_affs,
                mask_to_cover, patchshape, neighshape, rad, overlap_mask, None,
                consensus_vote_array, **kwargs)
        backend.note("n_selected", num_selected)
        backend.note("n_pairs", len(selected_patch_pairsIDs))
        if kwargs.get('return_intermediates'):
            return selected_patch_pairsIDs.numpy(), affinity_graph
        if kwargs.get('termAfterPatchGraph', False):
            return None, None
    else:
        affinity_graph = loadAffgraph(kwargs['aff_graph'], kwargs['selected_patch_pairs'])
    del consensus_vote_array

    # (6) labelling
    with backend.host_timer("s6_label_paint"):
        return affGraphToInstances(affinity_graph, pred_affs, patchshape, rad, None, None,
                                   instances, foreground, **kwargs)


def do_block(block, foreground, mask, numinst, **kwargs):
    """One block of a blockwise run (vote_instances.py:455-483): the intermediates when asked
    for, otherwise the instance map without its patch-radius border."""
    patchshape = np.asarray(kwargs.pop('patchshape'))
    result = to_instance_seg(block, foreground, mask, numinst, patchshape, **kwargs)
    if kwargs.get('return_intermediates'):
        return result
    instances = result[0]
    inner = tuple(slice(int(p) // 2, n - int(p) // 2) for p, n in zip(patchshape, instances.shape))
    return instances[inner]


def do_all(aff_file, patchshape=np.array([1, 25, 25]), **kwargs):
    """vote_instances.py:486-554: load one prediction file, assemble, write the result HDF
    (datasets ``<res_key>`` and ``vote_foreground``, gzip, attrs offset/resolution)."""
    logger.info("processing %s into %s", aff_file, kwargs['result_folder'])
    if type(patchshape) is not np.ndarray:
        patchshape = np.array(patchshape)
    res_ext = getResKey(**kwargs) if kwargs.get('add_suffix', False) else ''
    loaded = loadAffinities(aff_file, res_ext, patchshape=patchshape, **kwargs)
    if loaded is None:
        return
    affinities, numinst, foreground = loaded
    if foreground.ndim == 4:   # loadFg's leading axis on 3-d data
        foreground = foreground[0]
    mask = np.copy(foreground)
    if numinst is None:
        numinst = np.copy(foreground)
    kwargs['aff_file'] = aff_file
    res = to_instance_seg(affinities, foreground, mask, numinst, patchshape, **kwargs)
    res_key = kwargs.get('res_key', 'vote_instances')
    instances, foreground = res
    if instances is None and foreground is None:
        return
    foreground = foreground.astype(np.uint8)
    if kwargs.get('crop_to_foreground', True):
        # (:535-540: channel by channel for the stacked map; boolean indexing of the leading
        # axes would not broadcast)
        instances[..., foreground == 0] = 0
    fn = os.path.splitext(os.path.basename(aff_file))[0]
    out_fn = os.path.join(kwargs['result_folder'], fn + ".hdf")
    write_result(out_fn, {res_key + res_ext: instances, 'vote_foreground' + res_ext: foreground})


def write_result(out_fn, datasets):
    """HDF5 (the reference's format, vote_instances.py:542-554) through h5py or the HDF5 C library;
    without either a zarr store ``<stem>.zarr`` with the same dataset names, dtypes and attributes
    (io_hdflike.write_datasets)."""
    attrs = None
    if SKELETONIZE_SERVED_BY is not None:
        # which thinning produced the cover mask / bounding box of this result
        attrs = {"offset": (0, 0, 0), "resolution": (1, 1, 1), "skeletonize_foreground": SKELETONIZE_SERVED_BY}
    return io_hdflike.write_datasets(out_fn, datasets, attrs)


def main(**kwargs):
    """Whole-volume driver (vote_instances.py:557-604): command-line arguments overridden by
    keyword arguments; ``affinities`` is one prediction file or a directory of ``*.hdf``
    files, otherwise ``<basedir>/<mode>/processed/<checkpoint>/*.hdf``."""
    from_cli = not kwargs
    backend.tune_host_allocator(cli=from_cli or __name__ == "__main__")   # (no kwargs: argv drives it)
    required = kwargs['check_required'] if 'check_required' in kwargs else True
    args = vars(get_arguments(check_required=required, argv=None if from_cli else []))
    if kwargs:
        args = merge_dicts(args, kwargs)
    if 'check_required' in kwargs:
        assert type(args['patchshape']) in [np.ndarray, tuple, list], \
            "Please check type of patchshape {}".format(type(args['patchshape']))
        assert type(args['result_folder']) == str, \
            "Please check type of result_folder {}".format(type(args['result_folder']))
    if args.get('cuda') and not args.get('graphToInst', False):
        args['context'] = init_cuda()
    os.makedirs(args['result_folder'], exist_ok=True)

    source = args['affinities']
    if source is not None and (source.endswith(".zarr") or os.path.isfile(source)):
        do_all(source, **args)          # (the reference returns here without delete_cuda, :584-586)
        return
    if source is not None:
        if not os.path.isdir(source):
            raise RuntimeError("affinities (%s) should be file or dir" % source)
        pattern = os.path.join(source, "*.hdf")
    elif args['mode'] is not None and args['checkpoint'] is not None:
        pattern = os.path.join(args['basedir'], args['mode'], "processed", args['checkpoint'], "*.hdf")
    else:
        pattern = None
    if args.get('parallel'):
        raise NotImplementedError
    for aff_file in (glob.glob(pattern) if pattern else []):
        do_all(aff_file, **args)
    delete_cuda(args.get('context'))


if __name__ == "__main__":
    main()
