"""Reference-side binding of libppp_mi355x.so -- what a maintainer of PatchPerPix adds to keep
``vote_instances.py`` as it is and only replace the pycuda plug (path B).

It re-implements, on the C ABI of include/ppp_mi355x.h and nothing else from this repository,
  * the device shim ``PatchPerPix/vote_instances/cuda_code.py:5-59``
    (alloc_zero_array / sync / init_cuda / delete_cuda / get_cuda_stream), and
  * the three launchers ``create_consensus_array_cuda`` (consensus_array.py:71-206),
    ``rank_patches_cuda`` (ranked_patches.py:33-74), ``computePatchGraph_cuda``
    (aff_patch_graph.py:113-187), with the reference's argument lists and its
    [NSZ, NSY, NSX, Z, Y, X] consensus array.
Device buffers are torch-ROCm tensors (any owner of device memory would do: the library only
sees pointers).  tests/test_integration_binding.py runs this file against the goldens.
"""
import ctypes
import os

import numpy as np
import torch

_LIB_PATH = os.environ.get("PPP_LIB") or os.path.join(
    os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "patchperpix_amd", "csrc", "libppp_mi355x.so")
_lib = ctypes.CDLL(_LIB_PATH)
_lib.ppp_last_error.restype = ctypes.c_char_p
PPP_F32, PPP_F16 = 0, 1
PPP_CONS_REFERENCE = 1


class Box(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("z0", "y0", "x0", "z1", "y1", "x1")]


class Params(ctypes.Structure):                       # struct ppp_params (include/ppp_mi355x.h)
    _fields_ = [("abi_version", ctypes.c_int32),
                ("Z", ctypes.c_int32), ("Y", ctypes.c_int32), ("X", ctypes.c_int32),
                ("pz", ctypes.c_int32), ("py", ctypes.c_int32), ("px", ctypes.c_int32),
                ("th", ctypes.c_double), ("thi", ctypes.c_double),
                ("bg_rule", ctypes.c_int32), ("value_rule", ctypes.c_int32),
                ("use_overlap", ctypes.c_int32), ("normalise", ctypes.c_int32),
                ("norm_rank", ctypes.c_int32), ("count_pos_neg", ctypes.c_int32),
                ("norm_aff", ctypes.c_int32), ("cons_layout", ctypes.c_int32),
                ("cons_box", Box),
                ("origin_z", ctypes.c_int32), ("origin_y", ctypes.c_int32),
                ("origin_x", ctypes.c_int32), ("ring_z", ctypes.c_int32), ("pred_clean", ctypes.c_int32), ("rank_tile", ctypes.c_int32)]


def _check(rc):
    if rc:
        raise RuntimeError(_lib.ppp_last_error().decode())


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


# ---- cuda_code.py -----------------------------------------------------------------------------
def alloc_zero_array(shape, dtype):                   # was pycuda.driver.managed_zeros
    return torch.zeros(tuple(int(s) for s in np.atleast_1d(shape)),
                       dtype=getattr(torch, np.dtype(dtype).name), device="cuda")


def sync(context):
    torch.cuda.synchronize()


def init_cuda():
    if _lib.ppp_device_count() < 1:
        raise RuntimeError("no HIP device")
    return torch.device("cuda")


def delete_cuda(context):
    torch.cuda.empty_cache()


def get_cuda_stream():
    return torch.cuda.current_stream()


# ---- utilVoteInstances.loadKernelFromFile + setKernelBuildOptions -> run-time parameters -------
def params_from_flags(shape, patchshape, **kw):
    th = float(kw["patch_threshold"])
    P = Params(abi_version=_lib.ppp_abi_version(), Z=shape[0], Y=shape[1], X=shape[2],
               pz=patchshape[0], py=patchshape[1], px=patchshape[2],
               th=th, thi=th if th < 0.5 else 1.0 - th)
    if kw.get("vi_bg_use_inv_th", True):
        P.bg_rule = 2 if th < 0.5 else 0              # -DUSE_LESS_THAN_TH / -DUSE_INV_TH
    elif kw.get("vi_bg_use_half_th", False):
        P.bg_rule = 1
    elif kw.get("vi_bg_use_less_than_th", False):
        P.bg_rule = 2
    else:
        raise RuntimeError("how is bg defined for vote instances?")
    P.use_overlap = int(bool(kw.get("overlapping_inst")))
    P.value_rule = 2 if kw.get("consensus_norm_prob_product", True) else \
        1 if kw.get("consensus_prob_product", True) else 0
    P.normalise = int(kw.get("consensus_norm_aff", True))
    P.norm_rank = int(kw.get("rank_norm_patch_score", True))
    P.count_pos_neg = int(kw.get("rank_int_counter", False))
    P.norm_aff = int(kw.get("patch_graph_norm_aff", True))
    P.cons_layout = PPP_CONS_REFERENCE                # [NSZ, NSY, NSX, Z, Y, X], consensus_array.py:99-106
    P.cons_box = Box(0, 0, 0, *[int(s) for s in shape])
    return P


def _overlap(overlap_mask):
    return torch.as_tensor(np.ascontiguousarray(np.asarray(overlap_mask) != 0).astype(np.uint8), device="cuda")


# ---- the three launchers ------------------------------------------------------------------------
def create_consensus_array_cuda(pred_affs, overlap_mask, patchshape, neighshape, **kwargs):
    """fill (+ count pass) + normalise in ONE call; no JIT, no second -DOUTPUT_CNT launch."""
    P = params_from_flags(pred_affs.shape[1:], patchshape, **kwargs)
    out = alloc_zero_array(tuple(neighshape) + tuple(pred_affs.shape[1:]), np.float32)
    ov = _overlap(overlap_mask) if P.use_overlap else None
    _check(_lib.ppp_consensus(_ptr(pred_affs), PPP_F32, _ptr(ov), _ptr(out), None, ctypes.byref(P), None))
    return out


def rank_patches_cuda(pred_affs, consensus_vote_array, patchshape, neighshape, overlap_mask, **kwargs):
    P = params_from_flags(pred_affs.shape[1:], patchshape, **kwargs)
    scores = alloc_zero_array(pred_affs.shape[1:], np.float32)
    ov = _overlap(overlap_mask) if P.use_overlap else None
    _check(_lib.ppp_rank_patches(_ptr(pred_affs), PPP_F32, _ptr(consensus_vote_array), _ptr(ov),
                                 _ptr(scores), None, ctypes.byref(P), None))
    return scores


def computePatchGraph_cuda(pred_affs, consensus_vote_array, selected_patch_pairsIDs, patchshape,
                           neighshape, **kwargs):
    """One launch for all pairs (the reference loops over 512-pair launches, :137-159)."""
    P = params_from_flags(pred_affs.shape[1:], patchshape, **kwargs)
    pairs = torch.as_tensor(np.ascontiguousarray(selected_patch_pairsIDs, dtype=np.uint32).view(np.int32),
                            device="cuda")
    n = int(pairs.shape[0])
    aff = alloc_zero_array((n,), np.float32)
    _lib.ppp_patch_graph.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                     ctypes.POINTER(Params), ctypes.c_void_p]
    _check(_lib.ppp_patch_graph(_ptr(pred_affs), PPP_F32, _ptr(consensus_vote_array), _ptr(pairs), None,
                                n, _ptr(aff), ctypes.byref(P), None))
    return aff


# ---- the cuda=False stages (int16 votes, integer ranks, all-pairs graph weights) ------------------
# The reference's signatures take Python sets / the lookup table; a maintainer keeps the call sites
# and passes the arrays the sets were made from (pred_affs, foreground): the device kernels form the
# sets themselves (get_patch_sets.py:32-79), so fillLookup / computeFGBGsets are no longer called.
def _ref_plane_index(patchshape):
    """index into the reference's vote array (offset linearised over neighshape,
    utilVoteInstances.py:36-44) of every plane of the device layout (plane 0 = zero offset)"""
    ps = [int(p) for p in patchshape]
    ns1, ns2 = 2 * ps[1], 2 * ps[2]
    idx = [0]
    for dz in range(0, ps[0]):
        for dy in range(-(ps[1] - 1), ps[1]):
            for dx in range(-(ps[2] - 1), ps[2]):
                if (dz, dy, dx) > (0, 0, 0):
                    idx.append(dz * ns1 * ns2 + dy * ns2 + dx)
    return np.array(idx, dtype=np.int64)


def create_consensus_array(pred_affs, foreground, shape, patchshape, neighshape):
    """consensus_array.py:18-68 -> the reference's array: int16 (prod(neighshape), Z, Y, X) on the
    host (the offset lists it also returned only fed its own rank_patches)."""
    P = params_from_flags(shape, patchshape, patch_threshold=create_consensus_array.patch_threshold)
    _lib.ppp_np_vote_planes.restype = ctypes.c_int64
    planes = int(_lib.ppp_np_vote_planes(ctypes.byref(P)))
    fg = torch.as_tensor(np.ascontiguousarray(foreground).astype(np.uint8), device="cuda")
    votes = torch.empty((planes,) + tuple(shape), dtype=torch.int16, device="cuda")
    _check(_lib.ppp_np_consensus(_ptr(pred_affs), PPP_F32, _ptr(fg), _ptr(votes), ctypes.byref(P), None))
    full = np.zeros((int(np.prod(neighshape)),) + tuple(shape), dtype=np.int16)
    full[_ref_plane_index(patchshape)] = votes.cpu().numpy()
    return full, votes


create_consensus_array.patch_threshold = 0.9       # kwargs['patch_threshold'] at the call site


def rank_patches(pred_affs, foreground, votes_dev, all_patches_idx, patchshape):
    """ranked_patches.py:76-105: [(idx, score), ...] sorted like the reference's list"""
    shape = tuple(pred_affs.shape[1:])
    P = params_from_flags(shape, patchshape, patch_threshold=create_consensus_array.patch_threshold)
    fg = torch.as_tensor(np.ascontiguousarray(foreground).astype(np.uint8), device="cuda")
    score = torch.empty(shape, dtype=torch.int32, device="cuda")
    _check(_lib.ppp_np_rank_patches(_ptr(pred_affs), PPP_F32, _ptr(fg), _ptr(votes_dev), _ptr(score),
                                    ctypes.byref(P), None))
    s = score.cpu().numpy()
    ranked = [(np.asarray(idx), int(s[tuple(idx)])) for idx in all_patches_idx]
    return sorted(ranked, key=lambda x: x[1], reverse=True)
