"""Loaders and flag helpers (reference: PatchPerPix/vote_instances/utilVoteInstances.py).

The kernel templating half of the reference file (loadKernelFromFile, setKernelBuildOptions,
get_block_shape/get_grid_shape, :340-462) has no run-time counterpart here: shapes,
thresholds and build flags are fields of ``ppp_params`` (``backend.make_params``).
"""
import logging
import os
import pickle

import numpy as np
import scipy.special

from .. import backend
from . import io_hdflike

logger = logging.getLogger(__name__)


def setKernelBuildOptions(step=None, **kwargs):
    """Kept for API parity (utilVoteInstances.py:389-449): returns the -D style flag list the
    reference would pass to nvcc; the same decisions are taken by backend.make_params."""
    P = backend.make_params((1, 1, 1), (1, 1, 1), **dict(kwargs))
    opts = [{backend.BG_INV_TH: "-DUSE_INV_TH", backend.BG_HALF_TH: "-DUSE_HALF_TH",
             backend.BG_LESS_THAN_TH: "-DUSE_LESS_THAN_TH"}[P.bg_rule]]
    if P.use_overlap:
        opts.append("-DOVERLAP")
    if step == "consensus":
        if P.value_rule == backend.VAL_NORM_PROB_PRODUCT:
            opts.append("-DNORM_PROB_PRODUCT")
        elif P.value_rule == backend.VAL_PROB_PRODUCT:
            opts.append("-DPROB_PRODUCT")
    if step == "rank":
        if P.norm_rank:
            opts.append("-DNORM_PATCH_RANK")
        if P.count_pos_neg:
            opts.append("-DCOUNT_POS_NEG")
    if step == "patch_graph":
        opts = ["-DNORM_PATCH_AFFINITY"] if P.norm_aff else opts
    return opts


def loadFromFile(filename, shape=None, key=None):
    """utilVoteInstances.py:95-133."""
    logger.info("reading %s", filename)
    if filename.endswith("pickle"):
        with open(filename, "rb") as f:
            return pickle.load(f)
    if filename.endswith(("hdf", "zarr")):
        if key is None:
            raise SystemExit("provide hdf key for array")
        with io_hdflike.open_container(filename, "r") as f:
            return np.array(f[key])
    if filename.endswith("npy"):
        return np.load(filename)
    if "bin" in filename:
        with open(filename, "rb") as f:
            array = np.frombuffer(f.read(), dtype=np.int32)
        array.shape = shape
        return array
    raise SystemExit("invalid file")


def getFgThreshold(**kwargs):
    if kwargs.get("fg_thresh_vi", -1) > 0:
        return kwargs["fg_thresh_vi"]
    return kwargs["patch_threshold"]


def _numinst_from_prob(numinst_prob, **kwargs):
    numinst = np.argmax(numinst_prob, axis=0).astype(np.uint8)
    if kwargs.get("numinst_threshs"):
        numinst = np.zeros(numinst_prob.shape[1:], dtype=np.uint8)
        for i in range(len(kwargs["numinst_threshs"])):
            numinst[numinst_prob[i + 1] > kwargs["numinst_threshs"][i]] = i + 1
    return numinst


def maybeLoadNuminst(f, **kwargs):
    """utilVoteInstances.py:260-272."""
    if kwargs.get("numinst_key") is None:
        return None
    numinst_prob = np.squeeze(np.array(f[kwargs["numinst_key"]]))
    if len(numinst_prob.shape) == 3:
        numinst_prob = np.expand_dims(numinst_prob, axis=1)
    return _numinst_from_prob(numinst_prob, **kwargs)


def loadFg(f, **kwargs):
    """utilVoteInstances.py:275-303 (note the leading axis of size 1 it adds)."""
    aff_key = kwargs["aff_key"]
    fg_key = kwargs.get("fg_key", None)
    numinst_key = kwargs.get("numinst_key", None)
    fg_thresh = getFgThreshold(**kwargs)
    if fg_key is not None:
        foreground = np.array(f[fg_key])
        key = fg_key
    elif numinst_key is not None:
        numinst = _numinst_from_prob(np.array(f[numinst_key]), **kwargs)
        foreground = np.expand_dims((numinst > 0).astype(np.float32), axis=0)
        key = numinst_key
    else:
        mid = np.prod(kwargs["patchshape"]) // 2
        foreground = np.expand_dims(np.array(f[aff_key][mid]), axis=0)
        key = aff_key
    return foreground > fg_thresh, key


def returnFg(affs, numinst, fg, **kwargs):
    """utilVoteInstances.py:306-322."""
    fg_thresh = getFgThreshold(**kwargs)
    if kwargs.get("fg_key", None) is not None:
        foreground = np.squeeze(fg)
    elif kwargs.get("numinst_key", None) is not None:
        foreground = numinst > 0
    else:
        mid = np.prod(kwargs["patchshape"]) // 2
        foreground = affs[mid]
    return foreground > fg_thresh


def getResKey(**kwargs):
    """utilVoteInstances.py:325-337."""
    res_ext = "_" + str(kwargs["patch_threshold"]).replace(".", "")
    if not kwargs.get("skipThinCover", False):
        res_ext += "_tfgc"
    if kwargs["mws"]:
        res_ext += "_mws"
    if kwargs["sample"] < 1.0:
        res_ext += "_smp" + str(kwargs["sample"]).replace(".", "")
    return res_ext


def _crop(kwargs, axes):
    return tuple(slice(kwargs.get("crop_%s_s" % a, 0), kwargs.get("crop_%s_e" % a, None))
                 for a in axes)


def loadAffinities(aff_file, res_ext, patchshape=None, **kwargs):
    """utilVoteInstances.py:136-251: returns (affinities (C,Z,Y,X), numinst, foreground) or
    None when the result key already exists."""
    numinst = None
    if aff_file.endswith((".hdf", ".zarr")):
        with io_hdflike.open_container(aff_file, "r") as f:
            if "vote_instances" + res_ext in f.keys():
                logger.info("%s vote_instances %s already computed", aff_file, res_ext)
                return None
            if "volumes" in f.keys():
                aff_key = kwargs.get("aff_key")
                if aff_key is None:
                    aff_key = "volumes/pred_affs"
                    kwargs["aff_key"] = aff_key
                ds = f[aff_key]
                shape = ds.shape
                rotate_axes = False
                if patchshape is not None:
                    lin = int(np.prod(patchshape))
                    rotate_axes = shape[-1] == lin and shape[0] != lin
                if len(shape) == 3:
                    if rotate_axes:
                        aff = np.squeeze(np.array(ds[_crop(kwargs, "yx") + (slice(None),)]))
                        aff = np.ascontiguousarray(np.moveaxis(aff, -1, 0))
                    else:
                        aff = np.squeeze(np.array(ds[(slice(None),) + _crop(kwargs, "yx")]))
                    affinities = np.expand_dims(aff, axis=1)
                elif len(shape) == 4:
                    if rotate_axes:
                        aff = np.squeeze(np.array(ds[_crop(kwargs, "zyx") + (slice(None),)]))
                        affinities = np.ascontiguousarray(np.moveaxis(aff, -1, 0))
                    else:
                        affinities = np.squeeze(np.array(ds[(slice(None),) + _crop(kwargs, "zyx")]))
                else:
                    raise RuntimeError("check dimensions of array %s %s" % (aff_file, aff_key))
                if kwargs.get("isbiHack"):
                    affinities = affinities[:, :, ::2, ::2]
            else:
                affinities = np.array(f["images/pred_affs"])
                if affinities.shape[1] != 1:
                    affinities = np.expand_dims(affinities, axis=1)
                kwargs.setdefault("aff_key", "images/pred_affs")
            numinst = maybeLoadNuminst(f, **kwargs)
            foreground, _ = loadFg(f, **dict(kwargs, patchshape=patchshape))
    elif aff_file.endswith("npy"):
        affinities = np.load(aff_file)
        if affinities.shape[1] != 1:
            affinities = np.expand_dims(affinities, axis=1)
        mid = np.prod(patchshape) // 2
        foreground = np.array(affinities[mid]) > getFgThreshold(**kwargs)
        numinst = 1 * foreground
    else:
        logger.info("invalid affinities file, zarr, hdf or npy")
        raise SystemExit(-1)
    if np.min(affinities) < 0 and np.max(affinities) > 1:
        affinities = scipy.special.expit(affinities)
    return affinities, numinst, foreground
