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
    kw = dict(kwargs)
    if step != "consensus":
        # the reference looks at the value rule (and asserts that counted votes are not normalised) for the
        # consensus kernel only (:412-427); the other steps' switches do not depend on it
        kw.update(consensus_norm_prob_product=True)
    P = backend.make_params((1, 1, 1), (1, 1, 1), **kw)
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


def numinst_from_prob(prob, numinst_threshs=None, **_ignored):
    """Per-voxel instance count from the (K, ...) probabilities of "k instances here"
    (utilVoteInstances.py:254-270): class k = 1, 2, .. wherever its probability exceeds its
    threshold (later classes over earlier ones) when thresholds are configured, else the argmax."""
    if numinst_threshs:
        out = np.zeros(prob.shape[1:], dtype=np.uint8)
        for k, th in enumerate(numinst_threshs, start=1):
            out[prob[k] > th] = k
        return out
    return np.argmax(prob, axis=0).astype(np.uint8)


_numinst_from_prob = numinst_from_prob


class PredictionFile:
    """What the `label` task reads from one prediction container (a zarr store or an HDF5 file,
    opened by io_hdflike.open_container): the patch affinities as (C, Z, Y, X), the instance
    count and the foreground mask, with the reference's conventions for where the channel axis
    sits, 2-d data, crops and key defaults (utilVoteInstances.py:136-303).  The functions with
    the reference's names below are views of this object; the streaming reader
    (tiling.ZarrProvider) uses ``dataset`` / ``numinst`` / ``foreground`` and never ``affinities``."""

    def __init__(self, container, patchshape=None, **kwargs):
        self.f, self.kw = container, dict(kwargs)
        self.patchshape = None if patchshape is None else [int(p) for p in patchshape]
        # the prediction step writes under `volumes/`; older files keep `images/pred_affs`
        self.modern = "volumes" in container.keys()
        if self.kw.get("aff_key") is None:
            self.kw["aff_key"] = "volumes/pred_affs" if self.modern else "images/pred_affs"

    # ---- the affinity dataset ------------------------------------------------------------
    @property
    def dataset(self):
        return self.f[self.kw["aff_key"]]

    @property
    def channels_last(self):
        """(.., C) instead of (C, ..): recognised by the channel count (utilVoteInstances.py:170)"""
        if self.patchshape is None:
            return False
        shape, lin = self.dataset.shape, int(np.prod(self.patchshape))
        return shape[-1] == lin and shape[0] != lin

    def _crop(self, axes):
        return tuple(slice(self.kw.get("crop_%s_s" % a, 0), self.kw.get("crop_%s_e" % a, None)) for a in axes)

    def affinities(self):
        """(C, Z, Y, X) values as stored (Z = 1 for 2-d data), cropped; logits are NOT mapped here."""
        if not self.modern:
            aff = np.array(self.dataset)
            return aff if aff.shape[1] == 1 else np.expand_dims(aff, axis=1)
        ds = self.dataset
        nd = len(ds.shape) - 1
        if nd not in (2, 3):
            raise RuntimeError("check dimensions of array %s" % (self.kw["aff_key"],))
        spatial = self._crop("zyx"[-nd:])
        if self.channels_last:
            aff = np.ascontiguousarray(np.moveaxis(np.squeeze(np.array(ds[spatial + (slice(None),)])), -1, 0))
        else:
            aff = np.squeeze(np.array(ds[(slice(None),) + spatial]))
        if nd == 2:
            aff = np.expand_dims(aff, axis=1)
        if self.kw.get("isbiHack"):
            aff = aff[:, :, ::2, ::2]
        return aff

    # ---- instance count and foreground -----------------------------------------------------
    def numinst(self):
        """uint8 per voxel, or None without a numinst_key (utilVoteInstances.py:260-272)."""
        key = self.kw.get("numinst_key")
        if key is None:
            return None
        prob = np.squeeze(np.array(self.f[key]))
        if prob.ndim == 3:                      # 2-d data: (K, Y, X) -> (K, 1, Y, X)
            prob = np.expand_dims(prob, axis=1)
        return numinst_from_prob(prob, **self.kw)

    def foreground(self):
        """(mask, key it came from): the fg_key dataset, else numinst > 0, else the centre channel
        of the affinities -- thresholded; the last two with a leading axis of size 1
        (utilVoteInstances.py:275-303)."""
        th = getFgThreshold(**self.kw)
        if self.kw.get("fg_key") is not None:
            return np.array(self.f[self.kw["fg_key"]]) > th, self.kw["fg_key"]
        if self.kw.get("numinst_key") is not None:
            numinst = numinst_from_prob(np.array(self.f[self.kw["numinst_key"]]), **self.kw)
            return np.expand_dims((numinst > 0).astype(np.float32), axis=0) > th, self.kw["numinst_key"]
        mid = int(np.prod(self.patchshape)) // 2
        return np.expand_dims(np.array(self.dataset[mid]), axis=0) > th, self.kw["aff_key"]

    def load(self):
        """(affinities, numinst, foreground) of loadAffinities"""
        return self.affinities(), self.numinst(), self.foreground()[0]


def maybeLoadNuminst(f, **kwargs):
    """utilVoteInstances.py:260-272."""
    return PredictionFile(f, **kwargs).numinst() if kwargs.get("numinst_key") is not None else None


def loadFg(f, **kwargs):
    """utilVoteInstances.py:275-303 (note the leading axis of size 1 it adds)."""
    return PredictionFile(f, **kwargs).foreground()


def returnFg(affs, numinst, fg, **kwargs):
    """utilVoteInstances.py:306-322: the foreground of a block already in memory."""
    if kwargs.get("fg_key") is not None:
        source = np.squeeze(fg)
    elif kwargs.get("numinst_key") is not None:
        source = numinst > 0
    else:
        source = affs[int(np.prod(kwargs["patchshape"])) // 2]
    return source > getFgThreshold(**kwargs)


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


def loadAffinities(aff_file, res_ext, patchshape=None, **kwargs):
    """utilVoteInstances.py:136-251: returns (affinities (C,Z,Y,X), numinst, foreground) or
    None when the result key already exists."""
    if aff_file.endswith((".hdf", ".zarr")):
        with io_hdflike.open_container(aff_file, "r") as f:
            if "vote_instances" + res_ext in f.keys():
                logger.info("%s vote_instances %s already computed", aff_file, res_ext)
                return None
            affinities, numinst, foreground = PredictionFile(f, patchshape=patchshape, **kwargs).load()
    elif aff_file.endswith("npy"):
        affinities = np.load(aff_file)
        if affinities.shape[1] != 1:
            affinities = np.expand_dims(affinities, axis=1)
        foreground = np.array(affinities[int(np.prod(patchshape)) // 2]) > getFgThreshold(**kwargs)
        numinst = 1 * foreground
    else:
        logger.info("invalid affinities file, zarr, hdf or npy")
        raise SystemExit(-1)
    if np.min(affinities) < 0 and np.max(affinities) > 1:     # logits
        affinities = scipy.special.expit(affinities)
    return affinities, numinst, foreground
