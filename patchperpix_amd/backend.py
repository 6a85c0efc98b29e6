"""ctypes binding of libppp_mi355x.so (include/ppp_mi355x.h).

This module replaces the reference's device shim ``PatchPerPix/vote_instances/cuda_code.py``
(pycuda JIT + managed memory) for this package: device buffers are torch-ROCm tensors
(PyTorch is only the buffer / stream carrier), every computation happens inside the
hand-written HIP kernels of the shared library.  There is NO fallback: if the library is
missing, or no GPU is present, calls raise.
"""
import ctypes
import os

import numpy as np

from . import build as _build

F32, F16 = 0, 1
BG_INV_TH, BG_HALF_TH, BG_LESS_THAN_TH = 0, 1, 2
VAL_COUNT, VAL_PROB_PRODUCT, VAL_NORM_PROB_PRODUCT = 0, 1, 2
CONS_COMPACT, CONS_REFERENCE, CONS_VOXEL_MAJOR = 0, 1, 2
ABI_VERSION = 5
NONE_KEY = 0xFFFFFFFF
NONE_KEY64 = 1 << 62      # PPP_LABEL_NONE_KEY (streaming labels, int64 keys)
PAIR_KEY_FAR = 0x7FFFFFFFFFFFFFFF   # PPP_PAIR_KEY_FAR


class Box(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("z0", "y0", "x0", "z1", "y1", "x1")]

    def shape(self):
        return (self.z1 - self.z0, self.y1 - self.y0, self.x1 - self.x0)


class Params(ctypes.Structure):
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
                ("origin_x", ctypes.c_int32), ("ring_z", ctypes.c_int32),
                # 1: ppp_pred_check found the buffer passed as `pred` clean (S1's short classification);
                # 2 (this layer only; the library reads "not 1"): checked, not clean; 0: not checked
                ("pred_clean", ctypes.c_int32),
                # tile of centres per workgroup of the ranking kernel: 0 = the library's rule,
                # 1 / 2 / 3 = 8 x 8 x 16 / 8 x 16 x 16 / 16 x 8 x 16 (same scores; tiling.assemble times them)
                ("rank_tile", ctypes.c_int32)]

    @property
    def shape(self):
        return (self.Z, self.Y, self.X)

    @property
    def patchshape(self):
        return (self.pz, self.py, self.px)

    def copy(self):
        q = Params()
        ctypes.memmove(ctypes.byref(q), ctypes.byref(self), ctypes.sizeof(Params))
        return q


_LIB = None
_SIGNATURES = {
    # name: (restype, argtypes)
    "ppp_abi_version": (ctypes.c_int, []),
    "ppp_pred_check": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
                                      ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_last_error": (ctypes.c_char_p, []),
    "ppp_consensus_kernel_name": (ctypes.c_char_p, []),
    "ppp_reload_env": (None, []),
    "ppp_np_vote_planes": (ctypes.c_int64, [ctypes.POINTER(Params)]),
    "ppp_np_consensus": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_np_rank_patches": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_np_patch_graph": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_paint_patch_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_uint64, ctypes.c_void_p, ctypes.POINTER(Params),
                                            ctypes.c_void_p]),
    "ppp_counter_calibration": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                               ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "ppp_consensus_writes_voxel_major": (ctypes.c_int, [ctypes.POINTER(Params)]),
    "ppp_device_count": (ctypes.c_int, []),
    "ppp_cons_planes": (ctypes.c_int64, [ctypes.POINTER(Params)]),
    "ppp_cons_elems": (ctypes.c_int64, [ctypes.POINTER(Params)]),
    "ppp_consensus": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_consensus_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_consensus_part": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.POINTER(Params), ctypes.POINTER(Box),
                                          ctypes.c_void_p]),
    "ppp_cons_planes_to_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Box), ctypes.c_void_p,
                                               ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_rank_patches": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.POINTER(Box), ctypes.POINTER(Params),
                                        ctypes.c_void_p]),
    "ppp_rank_workspace_bytes": (ctypes.c_int64, [ctypes.POINTER(Box), ctypes.POINTER(Params)]),
    "ppp_rank_patches_vm": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(Box),
                                           ctypes.c_void_p, ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_patch_graph": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                       ctypes.c_void_p, ctypes.POINTER(Params),
                                       ctypes.c_void_p]),
    "ppp_patch_graph_by_patch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64,
                                                ctypes.c_void_p, ctypes.POINTER(Params),
                                                ctypes.c_void_p]),
    "ppp_patch_graph_by_patch_chunk_small": (ctypes.c_int32, [ctypes.POINTER(Params)]),
    "ppp_patch_graph_by_patch_chunked": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                        ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64,
                                                        ctypes.c_int32, ctypes.c_void_p,
                                                        ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_patch_graph_lcg_words": (ctypes.c_int64, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                   ctypes.POINTER(Params)]),
    "ppp_patch_graph_lcg": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_patch_graph_by_patch_lcg": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64,
                                                    ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_void_p, ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_label_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(Params)]),
    "ppp_label_components": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                            ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.POINTER(Params),
                                            ctypes.c_void_p]),
    "ppp_label_begin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                       ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_label_add": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_int64, ctypes.c_uint64, ctypes.c_void_p,
                                     ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_label_union_edges": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                             ctypes.c_void_p, ctypes.POINTER(Params),
                                             ctypes.c_void_p]),
    "ppp_label_finish": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_patch_pairs_count_subset": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                                    ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                                    ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_patch_pairs_fill_subset": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                                   ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                                   ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                                   ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_patch_pairs_count": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                             ctypes.c_void_p, ctypes.POINTER(Params),
                                             ctypes.c_void_p]),
    "ppp_patch_pairs_fill": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                            ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                            ctypes.c_void_p, ctypes.POINTER(Params),
                                            ctypes.c_void_p]),
    "ppp_pair_sort_keys": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                          ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_paint_instances": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                           ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_cons_to_reference": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_cons_to_voxel_major": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_patch_bits": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_uint64, ctypes.c_double, ctypes.c_void_p,
                                      ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_patch_bits_volume": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_double,
                                             ctypes.c_void_p, ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_pair_group_keys": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                           ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_patch_graph_by_patch_chunk": (ctypes.c_int32, [ctypes.POINTER(Params)]),
    "ppp_cover_workspace_bytes": (ctypes.c_int64, [ctypes.c_int64, ctypes.POINTER(Params)]),
    "ppp_cover_pass": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(Params),
                                      ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]),
    "ppp_cover_pass_voxel_bits": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                                 ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.POINTER(Params), ctypes.c_void_p,
                                                 ctypes.POINTER(ctypes.c_int32)]),
    "ppp_host_mws_sorted": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                             ctypes.c_int64, ctypes.c_void_p]),
    "ppp_rank_order_workspace_bytes": (ctypes.c_int64, [ctypes.POINTER(Params)]),
    "ppp_rank_order": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64),
                                      ctypes.c_void_p, ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_mws_edges_workspace_bytes": (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int64,
                                                       ctypes.POINTER(Params)]),
    "ppp_mws_edges": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                     ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64),
                                     ctypes.c_void_p, ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_thin_workspace_bytes": (ctypes.c_int64, [ctypes.c_int64, ctypes.POINTER(Params)]),
    "ppp_thin_cover": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.POINTER(Params), ctypes.c_void_p,
                                      ctypes.POINTER(ctypes.c_int32)]),
    "ppp_cover_open": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_cover_step": (ctypes.c_int, [ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int32, ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_cover_alive": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Params), ctypes.c_void_p,
                                       ctypes.POINTER(ctypes.c_int32)]),
    "ppp_cover_close": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(Params),
                                       ctypes.c_void_p]),
    "ppp_cover_zone": (ctypes.c_int, [ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                      ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_thin_shard_workspace_bytes": (ctypes.c_int64, [ctypes.POINTER(Params)]),
    "ppp_thin_open": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_thin_step": (ctypes.c_int, [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(Params),
                                     ctypes.c_void_p]),
    "ppp_thin_alive": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Params), ctypes.c_void_p,
                                      ctypes.POINTER(ctypes.c_int32)]),
    "ppp_thin_close": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_thin_zone": (ctypes.c_int, [ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                     ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_synth_pred": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                      ctypes.c_uint32, ctypes.c_float, ctypes.c_float,
                                      ctypes.c_float, ctypes.c_uint64, ctypes.POINTER(Params),
                                      ctypes.c_void_p]),
    "ppp_synth_pred_box": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                          ctypes.c_uint32, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                          ctypes.c_void_p, ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_decode_tail": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                       ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_float,
                                       ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_int, ctypes.POINTER(Params), ctypes.c_void_p]),
    "ppp_host_mws": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]),
    "ppp_host_rank_order": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p]),
    "ppp_host_cover_pass": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                             ctypes.c_double, ctypes.c_void_p,
                                             ctypes.POINTER(ctypes.c_int64),
                                             ctypes.POINTER(ctypes.c_int32)]),
    "ppp_host_cover_pass_marked": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                                    ctypes.c_double, ctypes.c_void_p,
                                                    ctypes.POINTER(ctypes.c_int64),
                                                    ctypes.POINTER(ctypes.c_int32), ctypes.c_void_p]),
    "ppp_host_thin_cover": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                             ctypes.c_void_p]),
    "ppp_host_skeletonize_3d": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "ppp_host_patch_pairs": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                              ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                              ctypes.c_void_p]),
}


def library_path():
    return _build.LIB


def lib():
    """Load libppp_mi355x.so; raise (never fall back) if it has not been built."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(
                "HIP library %s is missing -- build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` "
                "(patchperpix_amd has no CPU fallback)" % path)
        # torch first: it brings its own libamdhip64, and a second HIP runtime loaded BEFORE it
        # (through this library) leaves the process without a visible device
        _torch()
        L = ctypes.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        if L.ppp_abi_version() != ABI_VERSION:
            raise RuntimeError("libppp_mi355x.so ABI %d != %d" % (L.ppp_abi_version(), ABI_VERSION))
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        raise RuntimeError("libppp_mi355x: %s (code %d)" % (lib().ppp_last_error().decode(), rc))


def device_count():
    return lib().ppp_device_count()


# ----------------------------------------------------------------------------------------
# parameters from the reference's kwargs (utilVoteInstances.py:340-449)
# ----------------------------------------------------------------------------------------
def make_params(shape_zyx, patchshape, cons_box=None, cons_layout=CONS_COMPACT,
                origin=(0, 0, 0), **kw):
    """Translate the reference's keyword flags into ppp_params.

    Mirrors loadKernelFromFile's TH/THI substitution and setKernelBuildOptions' -D flags,
    including its defaults (vi_bg_use_inv_th defaults to True when absent) and its error
    for an undefined background rule."""
    th = float(kw["patch_threshold"])
    P = Params()
    P.abi_version = ABI_VERSION
    P.Z, P.Y, P.X = [int(s) for s in shape_zyx]
    P.pz, P.py, P.px = [int(p) for p in patchshape]
    P.th = th
    P.thi = th if th < 0.5 else 1.0 - th
    if kw.get("vi_bg_use_inv_th", True):
        P.bg_rule = BG_LESS_THAN_TH if th < 0.5 else BG_INV_TH
    elif kw.get("vi_bg_use_half_th", False):
        P.bg_rule = BG_HALF_TH
    elif kw.get("vi_bg_use_less_than_th", False):
        P.bg_rule = BG_LESS_THAN_TH
    else:
        raise RuntimeError("how is bg defined for vote instances?")
    P.use_overlap = 1 if kw.get("overlapping_inst", False) else 0
    if kw.get("consensus_norm_prob_product", True):
        P.value_rule = VAL_NORM_PROB_PRODUCT
    elif kw.get("consensus_prob_product", True):
        P.value_rule = VAL_PROB_PRODUCT
    else:
        assert \
            not kw.get("consensus_norm_aff", True) and \
            not kw.get("consensus_interleaved_cnt", True), \
            "no normalizing for accumulate consensus counter available"
        P.value_rule = VAL_COUNT
    P.normalise = 1 if kw.get("consensus_norm_aff", True) else 0
    P.norm_rank = 1 if kw.get("rank_norm_patch_score", True) else 0
    P.count_pos_neg = 1 if kw.get("rank_int_counter", False) else 0
    P.norm_aff = 1 if kw.get("patch_graph_norm_aff", True) else 0
    P.cons_layout = cons_layout
    if cons_box is None:
        cons_box = (0, 0, 0, P.Z, P.Y, P.X)
    P.cons_box = Box(*[int(v) for v in cons_box])
    P.origin_z, P.origin_y, P.origin_x = [int(v) for v in origin]
    return P


def params_from_kwargs(shape_zyx, patchshape, kwargs):
    """make_params for a reference-style kwargs dict (which may carry this package's own
    ``cons_box`` / ``cons_layout`` entries next to the reference's flags)."""
    kw = {k: v for k, v in kwargs.items() if k not in ("cons_box", "cons_layout", "origin")}
    return make_params(shape_zyx, patchshape, cons_box=kwargs.get("cons_box"),
                       cons_layout=kwargs.get("cons_layout", CONS_COMPACT),
                       origin=kwargs.get("origin", (0, 0, 0)), **kw)


# ----------------------------------------------------------------------------------------
# torch plumbing
# ----------------------------------------------------------------------------------------
def _torch():
    import torch
    return torch


def _dev_ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device tensors must be contiguous and on the GPU"
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(_torch().cuda.current_stream().cuda_stream)


# Optional per-kernel timing with HIP events on the stream the kernels are launched on
# (bench.py switches it on): EVENTS = {} collects name -> [(start, stop), ...].
EVENTS = None


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if EVENTS is not None:
            torch = _torch()
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream())

    def __exit__(self, *exc):
        if EVENTS is not None:
            self.b.record(_torch().cuda.current_stream())
            EVENTS.setdefault(self.name, []).append((self.a, self.b))


HOST_TIMES = None   # name -> [seconds, ...] wall time of host stages (bench.py switches on)


class host_timer:
    """Wall-clock timer for a host stage (synchronises the device first when enabled, so a
    stage is not charged for kernels still running from the previous one)."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if HOST_TIMES is not None:
            import time
            _torch().cuda.synchronize()
            self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        if HOST_TIMES is not None:
            import time
            _torch().cuda.synchronize()
            HOST_TIMES.setdefault(self.name, []).append(time.perf_counter() - self.t0)


NOTES = {}
LAST_PLAN = None    # the memory plan of the last to_instance_seg call that made one (vote_instances.py)


def note(key, value):
    """Record a workload statistic (number of selected patches, pairs, ...) or a name."""
    NOTES[key] = value if isinstance(value, str) else int(value)


def note_add(key, value):
    NOTES[key] = NOTES.get(key, 0) + int(value)


def event_times_ms():
    """name -> list of elapsed ms (call after a synchronize)."""
    return {k: [a.elapsed_time(b) for a, b in v] for k, v in (EVENTS or {}).items()}


def pred_dtype_code(t):
    torch = _torch()
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float16:
        return F16
    raise TypeError("pred must be float32 or float16, got %s" % t.dtype)


def counter_calibration(src, n_read, dst, n_write):
    """ppp_counter_calibration: read n_read elements of `src` (float16 / float32 device tensor),
    write n_write floats to `dst` (float32 device tensor).  Returns the true (read, write) bytes."""
    check(lib().ppp_counter_calibration(_dev_ptr(src), pred_dtype_code(src), int(n_read), _dev_ptr(dst),
                                        int(n_write), _stream()))
    return int(n_read) * src.element_size(), int(n_write) * 4


_HOST_ALLOCATOR = None


def tune_host_allocator(cli=False):
    """Host allocator of a long-running service: the pipeline's host arrays (masks, instance map:
    a few MB each at 140^3, several dozen per call) would otherwise be mmap'ed, first-touched and
    unmapped again on every call by glibc (~5 ms per step of page faults at 140^3).  Same effect
    as MALLOC_TRIM_THRESHOLD_ / MALLOC_MMAP_THRESHOLD_ in the environment.

    The thresholds are process-wide and keep freed host memory from going back to the OS, so they
    are set only for a process that IS one of the drivers (`cli=True`: run_ppp / vote_instances /
    stitch_patch_graph run as a program, bench.py) -- unless PPP_MALLOPT=0 (or the older
    PPP_BENCH_MALLOPT=0) -- or, for an application that calls main() as a library function, when
    it opts in with PPP_MALLOPT=1.  Returns what was done (each mallopt call reported on its own:
    many glibc builds refuse an M_MMAP_THRESHOLD above 32 MiB)."""
    global _HOST_ALLOCATOR
    if _HOST_ALLOCATOR is not None:
        return _HOST_ALLOCATOR
    want = os.environ.get("PPP_MALLOPT", os.environ.get("PPP_BENCH_MALLOPT"))
    if want == "0" or (not cli and want != "1"):
        if cli or want == "0":
            _HOST_ALLOCATOR = "off"
        return _HOST_ALLOCATOR or "untouched (library call; PPP_MALLOPT=1 opts in)"
    try:
        libc = ctypes.CDLL("libc.so.6")
        trim = bool(libc.mallopt(-1, 1 << 30))             # M_TRIM_THRESHOLD
        mmap = bool(libc.mallopt(-3, 1 << 30))             # M_MMAP_THRESHOLD
        if not mmap:
            mmap32 = bool(libc.mallopt(-3, 32 << 20))      # the largest value older glibc accepts
        _HOST_ALLOCATOR = "trim threshold 1 GiB: %s; mmap threshold 1 GiB: %s%s" % (
            "set" if trim else "refused", "set" if mmap else "refused",
            "" if mmap else (", 32 MiB: %s" % ("set" if mmap32 else "refused")))
    except OSError:
        _HOST_ALLOCATOR = "no libc"
    return _HOST_ALLOCATOR


def reload_env():
    """The library reads its PPP_* development switches once; after changing one in a running
    process (tests that compare kernel variants) this makes it look again."""
    lib().ppp_reload_env()


def to_device_pred(pred, device="cuda", keep_f16=True):
    """Host ndarray / tensor -> contiguous device tensor (f16 stays f16: widening is exact
    and happens in registers)."""
    torch = _torch()
    if isinstance(pred, np.ndarray):
        if pred.dtype not in (np.float32, np.float16):
            pred = pred.astype(np.float32)
        pred = torch.from_numpy(np.ascontiguousarray(pred))
    if pred.dtype not in (torch.float32, torch.float16):
        pred = pred.float()
    if pred.dtype == torch.float16 and not keep_f16:
        pred = pred.float()
    return pred.to(device).contiguous()


# ----------------------------------------------------------------------------------------
# device stages
# ----------------------------------------------------------------------------------------
_CHECK_FLAG = {}


def pred_check(pred, P):
    """ppp_pred_check over the whole (contiguous) prediction tensor: 1 when every value lies in [0, 1]
    and none sits between the two class tests of P (with the shipped rule: none equals the threshold),
    2 otherwise -- the value of ppp_params.pred_clean for calls on THIS tensor while it is unchanged.
    One streaming read (196 GB at 512^3 / 9^3: 50 ms per volume) and one device -> host copy of the
    flag.  PPP_S1_CLEAN=0: never asked (returns 2)."""
    torch = _torch()
    if os.environ.get("PPP_S1_CLEAN", "1") == "0" or not torch.is_tensor(pred) or not pred.is_cuda \
            or not pred.is_contiguous():
        return 2
    key = str(pred.device)
    if key not in _CHECK_FLAG:
        _CHECK_FLAG[key] = torch.zeros((1,), dtype=torch.int32, device=pred.device)
    flag = _CHECK_FLAG[key]
    with _timed("pred_check"):
        check(lib().ppp_pred_check(_dev_ptr(pred), pred_dtype_code(pred), int(pred.numel()), _dev_ptr(flag),
                                   ctypes.byref(P), _stream()))
    bad = int(flag.item())
    note("pred_unclean_bits", bad)
    return 1 if bad == 0 else 2


def with_pred_clean(pred, P):
    """P with pred_clean decided (a copy when it was not): callers that pass the same tensor to many
    S1 launches (tiling.assemble: once per frame) decide once; a direct call decides per call."""
    if P.pred_clean != 0:
        return P
    Q = P.copy()
    Q.pred_clean = pred_check(pred, P)
    return Q


def consensus(pred, overlap, P, want_count=False, out=None, open_rows=False):
    """S1.  Returns cons (and count) as device float32 tensors shaped
    [planes, bz, by, bx] (compact), [bz, by, bx, W] (voxel-major) or [NSZ, NSY, NSX, Z, Y, X]
    (reference layout).  out: a flat float32 device tensor to carve the result from (the tiled
    path keeps ONE buffer for all its tiles: tens of GB allocated and freed per tile fragment the
    caching allocator until a tile that fits on paper no longer does)."""
    torch = _torch()
    L = lib()
    P = with_pred_clean(pred, P)
    if P.cons_layout == CONS_REFERENCE:
        shape = (2 * P.pz if P.pz > 1 else 1, 2 * P.py, 2 * P.px, P.Z, P.Y, P.X)
    elif P.cons_layout == CONS_VOXEL_MAJOR:
        shape = P.cons_box.shape() + (int(L.ppp_cons_planes(ctypes.byref(P))),)
    else:
        shape = (int(L.ppp_cons_planes(ctypes.byref(P))),) + P.cons_box.shape()
    n_el = int(np.prod(shape))
    if out is not None and out.numel() >= n_el:
        cons = out[:n_el].view(shape)
    else:
        cons = _big_empty(shape, pred.device)
    cnt = torch.empty(shape, dtype=torch.float32, device=pred.device) if want_count else None
    note_add("s1_base_voxels", int(np.prod(P.cons_box.shape())))
    note_add("s1_output_bytes", 4 * n_el)
    with _timed("consensus"):
        if open_rows and P.cons_layout == CONS_VOXEL_MAJOR and cnt is None:
            # rows for the library's own consumers: no zeroing of entries they never read
            check(L.ppp_consensus_rows(_dev_ptr(pred), pred_dtype_code(pred), _dev_ptr(overlap),
                                       _dev_ptr(cons), ctypes.byref(P), _stream()))
        else:
            check(L.ppp_consensus(_dev_ptr(pred), pred_dtype_code(pred), _dev_ptr(overlap),
                                  _dev_ptr(cons), _dev_ptr(cnt), ctypes.byref(P), _stream()))
    note("s1_kernel", L.ppp_consensus_kernel_name().decode())
    return (cons, cnt) if want_count else cons


def _big_empty(shape, device):
    """float32 device buffer of tens of GB.  A block of the right size is normally waiting in
    torch's cache from the previous slab / step (a fresh hipMalloc costs ~25 ms per GB); only
    when the allocator really runs dry are the cached blocks handed back first."""
    torch = _torch()
    try:
        return torch.empty(shape, dtype=torch.float32, device=device)
    except torch.OutOfMemoryError:
        torch.cuda.empty_cache()
        return torch.empty(shape, dtype=torch.float32, device=device)


def rank_patches(pred, cons, overlap, P, score_box=None, out=None):
    """S2.  Returns the (Z, Y, X) float32 score volume on the device.  With a VOXEL_MAJOR
    consensus (P.cons_layout) the row-stationary kernel runs (ppp_rank_patches_vm)."""
    torch = _torch()
    if out is None:
        out = torch.zeros(P.shape, dtype=torch.float32, device=pred.device)
    box = None if score_box is None else ctypes.byref(Box(*[int(v) for v in score_box]))
    if P.cons_layout == CONS_VOXEL_MAJOR:
        nbytes = int(lib().ppp_rank_workspace_bytes(box, ctypes.byref(P)))
        check(min(nbytes, 0))
        if nbytes == 0:
            raise RuntimeError("libppp_mi355x: no voxel-major ranking kernel for this configuration")
        work = torch.empty(nbytes, dtype=torch.uint8, device=pred.device)
        # voxels whose consensus row the launch reads: the score box grown by the patch radius
        sb = (0, 0, 0) + tuple(P.shape) if score_box is None else tuple(int(v) for v in score_box)
        cb = P.cons_box
        rad = (P.pz // 2, P.py // 2, P.px // 2)
        lo3 = [max(sb[a] - rad[a], (cb.z0, cb.y0, cb.x0)[a]) for a in range(3)]
        hi3 = [min(sb[3 + a] + rad[a], (cb.z1, cb.y1, cb.x1)[a]) for a in range(3)]
        note_add("s2_base_voxels", int(np.prod([max(0, h - l) for l, h in zip(lo3, hi3)])))
        with _timed("rank_patches"):
            check(lib().ppp_rank_patches_vm(_dev_ptr(pred), pred_dtype_code(pred), _dev_ptr(cons),
                                            _dev_ptr(overlap), _dev_ptr(out), box, _dev_ptr(work),
                                            ctypes.byref(P), _stream()))
        return out
    with _timed("rank_patches"):
        check(lib().ppp_rank_patches(_dev_ptr(pred), pred_dtype_code(pred), _dev_ptr(cons),
                                     _dev_ptr(overlap), _dev_ptr(out), box, ctypes.byref(P),
                                     _stream()))
    return out


def rank_vm_available(P):
    """True when ppp_rank_patches_vm handles this configuration (cubic 3/5/7/9 patches, float
    accumulation) and PPP_RANK does not ask for the gather kernel."""
    if os.environ.get("PPP_RANK", "vm") != "vm":
        return False
    Pv = P.copy()
    Pv.cons_layout = CONS_VOXEL_MAJOR
    # (the question is about the configuration, not a tile; with ring_z set -- "from a RING of rows?", which
    # only the workgroup-per-tile kernel reads -- the box is as thick as a ring may hold)
    Pv.cons_box = Box(0, 0, 0, P.Z, P.Y, P.X)
    if P.ring_z and P.ring_z < P.Z:
        # (a ring holds fewer slices than the volume: ask about centres whose rows fit into it)
        Pv.cons_box = Box(0, 0, 0, P.ring_z, P.Y, P.X)
        sb = Box(0, 0, 0, max(1, P.ring_z - P.pz // 2), P.Y, P.X)
        return int(lib().ppp_rank_workspace_bytes(ctypes.byref(sb), ctypes.byref(Pv))) > 0
    return int(lib().ppp_rank_workspace_bytes(None, ctypes.byref(Pv))) > 0


def pair_order(pairs, P):
    """Processing order for ppp_patch_graph: rows grouped by patch offset d = B - A, and by
    position of A inside a group.  pairs: device int32 [N, 6]; returns device int32 [N]."""
    torch = _torch()
    n = int(pairs.shape[0])
    keys = torch.empty((n,), dtype=torch.int64, device=pairs.device)
    check(lib().ppp_pair_sort_keys(_dev_ptr(pairs), n, _dev_ptr(keys), ctypes.byref(P), _stream()))
    return torch.argsort(keys).to(torch.int32)


def patch_graph_prepare(pred, pairs, Pv, ahead=False):
    """The part of S5 (patch_graph_by_patch) that needs no consensus: the rows grouped by patch A
    (inside a group by patch offset, so neighbouring lanes do similar work), the plan of the thinning
    masks and their buffers.  (`ahead` is accepted and ignored: rounds 4-5 ran the masks of tile t + 1
    on a side stream beside tile t's per-patch kernel; with a wave per pair the masks of a 512^3 step
    take 0.14 s and running them beside that kernel only slowed it -- 6.24 s in line, 6.70 s beside,
    DESIGN.md section 4 -- so the side stream is gone.)
    Pv: parameters of the FRAME `pred` and the rows live in (consensus box and ring do not matter)."""
    import types
    torch = _torch()
    n = int(pairs.shape[0])
    job = types.SimpleNamespace(n=n, n_live=0, plan=None, bufs=[None, None], masks_done=set())
    job.aff = torch.zeros((n,), dtype=torch.float32, device=pred.device)
    if n == 0:
        return job
    with host_timer("s5e_group_sort"):
        keys = torch.empty((n,), dtype=torch.int64, device=pairs.device)
        check(lib().ppp_pair_group_keys(_dev_ptr(pairs), n, _dev_ptr(keys), ctypes.byref(Pv), _stream()))
        keys, order = torch.sort(keys)
        # rows that cannot share a stored consensus offset keep the zero aff was created with
        n_live = int(torch.searchsorted(keys, torch.tensor([PAIR_KEY_FAR], dtype=torch.int64,
                                                           device=keys.device)).item())
        note_add("s5_rows_dispatched", n_live)
        job.n_live = n_live
        if n_live == 0:
            return job
        keys, order = keys[:n_live], order[:n_live]
    dkey = keys & 0x1FFFF                         # patch offset B - A
    keys >>= 18                                   # linear index of patch A
    _, counts = torch.unique_consecutive(keys, return_counts=True)
    del keys
    zero = torch.zeros((1,), dtype=torch.int64, device=pred.device)
    job.group_start = torch.cat([zero, torch.cumsum(counts, 0)])
    chunk = int(lib().ppp_patch_graph_by_patch_chunk(ctypes.byref(Pv)))
    if chunk <= 0:
        raise RuntimeError("libppp_mi355x: no per-patch kernel for this patch shape")
    # few rows per patch (thinned covers): one- or two-wave workgroups
    small = int(lib().ppp_patch_graph_by_patch_chunk_small(ctypes.byref(Pv)))
    mode = os.environ.get("PPP_PA_CHUNK", "auto")
    if mode == "small" or (mode == "auto" and n_live <= 1.5 * small * int(counts.shape[0])):
        chunk = small
    job.chunk = chunk
    job.chunk_offsets = torch.cat([zero, torch.cumsum((counts + chunk - 1) // chunk, 0)])
    job.n_groups = int(counts.shape[0])
    job.order32 = order.to(torch.int32)
    del order
    # thinning masks beforehand: the groups are cut into batches whose masks fit the budget (one
    # buffer, filled and read batch after batch; a second buffer when a batch's masks are made
    # beside the per-patch kernel of the batch before)
    # The plan's temporaries (about eight int64 per dispatched row) and the mask buffers are not
    # part of the tile planner's budget: the mask budget is capped by what is free right now, and
    # running out of memory anywhere in here falls back to the generator inside the kernel.
    oom = getattr(torch, "OutOfMemoryError", None) or torch.cuda.OutOfMemoryError
    plan = drops = None
    try:
        plan = _lcg_plan(dkey, job.group_start, Pv)
        if plan is not None:
            drops = torch.empty((plan["buffer_words"],), dtype=torch.int64, device=pred.device)
    except oom:
        plan = drops = None                       # the kernel runs the generator itself
        torch.cuda.empty_cache()
    del dkey
    job.plan = plan
    job.cuts = plan["group_cuts"] if plan is not None else [0, job.n_groups]
    co_host = job.chunk_offsets[torch.tensor(job.cuts, dtype=torch.int64, device=pred.device)].cpu().tolist()
    job.co_host = dict(zip(job.cuts, co_host))    # blocks before each cut: ONE copy for all batches
    job.bufs = [drops, drops]
    return job


def _pa_masks(job, pred, pairs, Pv, b):
    """the thinning masks of batch b into the buffer (filled and read batch after batch)"""
    n_b = len(job.cuts) - 1
    if job.plan is None or b >= n_b or b in job.masks_done:
        return
    lo, hi = job.plan["pos_cuts"][b], job.plan["pos_cuts"][b + 1]
    if hi > lo:
        with _timed("patch_graph_lcg"):
            check(lib().ppp_patch_graph_lcg(
                _dev_ptr(pred), pred_dtype_code(pred), _dev_ptr(pairs), _dev_ptr(job.order32),
                _dev_ptr(job.plan["pos"][lo:hi]), hi - lo, _dev_ptr(job.plan["drop_off"]),
                _dev_ptr(job.bufs[b % 2]), ctypes.byref(Pv), _stream()))
    job.masks_done.add(b)


def patch_graph_by_patch(pred, cons_vm, pairs, Pv, job=None):
    """S5 with one workgroup per patch A (ppp_patch_graph_by_patch).  job: what patch_graph_prepare
    made for these rows (None: made here)."""
    if job is None:
        job = patch_graph_prepare(pred, pairs, Pv)
    aff = job.aff
    if job.n == 0 or job.n_live == 0:
        return aff
    plan, cuts = job.plan, job.cuts
    n_b = len(cuts) - 1
    for b in range(n_b):
        g0, g1 = cuts[b], cuts[b + 1]
        if plan is not None:
            _pa_masks(job, pred, pairs, Pv, b)
        co = job.chunk_offsets[g0:g1 + 1]
        n_blocks = int(job.co_host[g1] - job.co_host[g0])
        with _timed("patch_graph"):
            check(lib().ppp_patch_graph_by_patch_lcg(
                _dev_ptr(pred), pred_dtype_code(pred), _dev_ptr(cons_vm), _dev_ptr(pairs),
                _dev_ptr(job.order32), _dev_ptr(job.group_start[g0:g1 + 1].contiguous()),
                _dev_ptr((co - co[0]).contiguous()), g1 - g0, n_blocks, job.chunk, _dev_ptr(aff),
                _dev_ptr(plan["drop_off"]) if plan is not None else None,
                _dev_ptr(job.bufs[b % 2]) if plan is not None else None, ctypes.byref(Pv), _stream()))
    return aff


def lcg_words(dz, dy, dx, P):
    """uint64 words of precomputed thinning masks for a pair with patch offset (dz, dy, dx)
    (ppp_patch_graph_lcg_words as array arithmetic, for the patch widths whose kernel reads masks --
    the library returns 0 for the others; any integer arrays or scalars)."""
    nz, ny, nx = P.pz - abs(dz), P.py - abs(dy), P.px - abs(dx)
    rpc = 64 // P.px
    nch = (P.px + rpc - 1) // rpc
    inter = (nz > 0) & (ny > 0) & (nx > 0)
    return nz * ny * nx * nz * nch * inter


def _lcg_plan(dkey, group_start, Pv):
    """Which dispatched pair rows get their thinning decisions made beforehand (ppp_patch_graph_lcg):
    those whose windows intersect.  The groups (patches A) are cut into batches whose masks fit
    PPP_PA_LCG_BYTES (default 4 GiB -- a tile of the thinned 512^3 / 9^3 cover needs 1.9 GB; 0 = the
    per-patch kernel runs the generator itself; so it does for patch widths whose kernel does not
    read masks: ppp_patch_graph_lcg_words is 0 then), at most PPP_PA_LCG_BATCHES (64) of them: rows
    of later groups keep the generator.  dkey: offset code of the dispatched rows
    (ppp_pair_group_keys), group_start: int64 [groups + 1].  Returns None or a dict:
      group_cuts  [batches + 1] group indices, pos_cuts [batches + 1] cuts of `pos`,
      pos         int64 positions of the served rows, batch after batch, sorted by offset inside
      drop_off    int64 per dispatched row: word offset inside its batch's masks, -1 = none
      buffer_words  words of the largest batch"""
    torch = _torch()
    budget = int(os.environ.get("PPP_PA_LCG_BYTES", str(4 << 30)))
    if dkey.is_cuda:
        # never more than a quarter of what is free (+ what torch's allocator holds unused), after
        # the plan's own temporaries: ~8 int64 per dispatched row
        free = torch.cuda.mem_get_info()[0] + torch.cuda.memory_reserved() - torch.cuda.memory_allocated()
        budget = min(budget, max(0, int(0.25 * (free - 64 * int(dkey.shape[0])))))
    budget //= 8
    max_batches = int(os.environ.get("PPP_PA_LCG_BATCHES", "64"))
    if budget <= 0 or int(lib().ppp_patch_graph_lcg_words(0, 0, 0, ctypes.byref(Pv))) <= 0:
        return None
    wx, wy = 4 * Pv.px + 1, 4 * Pv.py + 1
    dx = dkey % wx - 2 * Pv.px
    dy = (dkey // wx) % wy - 2 * Pv.py
    dz = dkey // (wx * wy) - 2 * Pv.pz
    words = lcg_words(dz, dy, dx, Pv)
    del dx, dy, dz
    ends = torch.cumsum(words, 0)
    total = int(ends[-1].item())
    if total == 0:
        return None
    n_groups = int(group_start.shape[0]) - 1
    zero = torch.zeros((1,), dtype=torch.int64, device=dkey.device)
    ends0 = torch.cat([zero, ends])                       # words before row i
    g_end = ends0[group_start[1:]]                        # words up to the end of group g
    g_words = g_end - ends0[group_start[:-1]]
    # windows of (budget - largest group) words: a batch = the groups that END in one window, so
    # it holds at most a window plus the part of its first group before the window
    window = budget - int(g_words.max().item())
    if window <= 0:
        return None
    batch_of_group = torch.clamp(g_end - 1, min=0) // window
    n_batches = int(batch_of_group[-1].item()) + 1
    n_served_batches = min(n_batches, max_batches)
    # (a group larger than the window leaves windows in which no group ends: number the batches
    # by the windows that do hold a group end, so that none is empty and PPP_PA_LCG_BATCHES counts
    # batches with work)
    _, batch_of_group = torch.unique_consecutive(batch_of_group, return_inverse=True)
    n_batches = int(batch_of_group[-1].item()) + 1
    n_served_batches = min(n_batches, max_batches)
    firsts = torch.searchsorted(batch_of_group, torch.arange(n_served_batches + 1, device=dkey.device))
    group_cuts = [int(v) for v in firsts.tolist()]
    if n_served_batches < n_batches:                      # the rest: one more launch, no masks
        group_cuts_all = group_cuts + [n_groups]
    else:
        group_cuts[-1] = n_groups
        group_cuts_all = group_cuts
    row_cuts = group_start[torch.tensor(group_cuts, device=dkey.device)]
    base = ends0[row_cuts]                                # words before each served batch (+ end)
    n_rows = int(dkey.shape[0])
    batch_of_row = torch.clamp(torch.searchsorted(row_cuts, torch.arange(n_rows, device=dkey.device),
                                                  right=True) - 1, max=n_served_batches)
    served = (words > 0) & (batch_of_row < n_served_batches)
    drop_off = torch.where(served, ends - words - base[torch.clamp(batch_of_row, max=n_served_batches - 1)],
                           torch.full_like(ends, -1))
    pos = torch.nonzero(served).reshape(-1)
    # batch after batch; inside a batch the lanes of a wave = rows of (nearly) the same offset
    pos = pos[torch.argsort(batch_of_row[pos] * (1 << 17) + dkey[pos])]
    pos_cuts = torch.searchsorted(batch_of_row[pos].contiguous(),
                                  torch.arange(n_served_batches + 1, device=dkey.device)).tolist()
    buffer_words = int((base[1:] - base[:-1]).max().item())
    note_add("s5_rows_lcg_beforehand", int(pos.shape[0]))
    note_add("s5_lcg_batches", n_served_batches)
    return dict(group_cuts=group_cuts_all, pos_cuts=[int(v) for v in pos_cuts] + [int(pos.shape[0])] *
                (len(group_cuts_all) - len(group_cuts)), pos=pos.contiguous(),
                drop_off=drop_off.contiguous(), buffer_words=max(buffer_words, 1))


def patch_graph_auto(pred, cons_compact, pairs, P, job=None):
    """S5 from a COMPACT consensus: re-layout to voxel-major, then the workgroup-per-patch
    kernel (consensus rows staged once per patch in LDS; 0.3 TB instead of 5.4 TB of HBM reads
    on the 140^3 benchmark).  PPP_PATCH_GRAPH=pairs selects the pair-per-lane gather kernel
    with offset-grouped lanes (same bits), which is also what patch shapes without a per-patch
    specialisation use."""
    if P.cons_layout == CONS_VOXEL_MAJOR:
        vm, Pv = cons_compact, P          # already re-laid out (kept from the ranking stage)
    elif getattr(cons_compact, "_ppp_vm", None) is not None:
        vm, Pv = cons_compact._ppp_vm     # the ranking stage of the stage pipeline left it here
        cons_compact._ppp_vm = None
    else:
        vm, Pv = cons_to_voxel_major(cons_compact, P)
    if os.environ.get("PPP_PATCH_GRAPH", "patch") != "pairs" and \
            int(lib().ppp_patch_graph_by_patch_chunk(ctypes.byref(Pv))) > 0 and \
            max(P.pz, P.py) <= P.px:
        return patch_graph_by_patch(pred, vm, pairs, Pv, job=job)
    return patch_graph(pred, vm, pairs, Pv, order=pair_order(pairs, Pv))


def device_patch_pairs(sorted_zyx, P, max_ps_dist=2, include_single=True):
    """Pair rows on the device from the x-sorted selected list (device int32 [n, 3]).
    Returns device int32 [rows, 6] (uint32 bit patterns) or None when there are no rows."""
    torch = _torch()
    n = int(sorted_zyx.shape[0])
    counts = torch.zeros((max(n, 1),), dtype=torch.int64, device=sorted_zyx.device)
    with _timed("patch_pairs"):
        check(lib().ppp_patch_pairs_count(_dev_ptr(sorted_zyx), n, int(max_ps_dist),
                                          _dev_ptr(counts), ctypes.byref(P), _stream()))
        ends = torch.cumsum(counts, 0)
        n_rows = int(ends[n - 1].item()) if n else 0
        total = n_rows + (n if include_single else 0)
        if total == 0:
            return None
        offsets = (ends - counts).contiguous()
        rows = torch.empty((total, 6), dtype=torch.int32, device=sorted_zyx.device)
        check(lib().ppp_patch_pairs_fill(_dev_ptr(sorted_zyx), n, int(max_ps_dist),
                                         _dev_ptr(offsets), n_rows, 1 if include_single else 0,
                                         _dev_ptr(rows), ctypes.byref(P), _stream()))
    return rows


def pair_counts_subset(sorted_zyx, subset, P, max_ps_dist=2):
    """Number of partners j > i of the patches listed in `subset` (device int64 indices into the
    x-sorted list).  Returns device int64 [n], zero outside the subset."""
    torch = _torch()
    n = int(sorted_zyx.shape[0])
    counts = torch.zeros((max(n, 1),), dtype=torch.int64, device=sorted_zyx.device)
    m = int(subset.numel())
    if n and m:
        with _timed("patch_pairs"):
            check(lib().ppp_patch_pairs_count_subset(_dev_ptr(sorted_zyx), n, int(max_ps_dist),
                                                     _dev_ptr(subset), m, _dev_ptr(counts),
                                                     ctypes.byref(P), _stream()))
    return counts[:n]


def pairs_subset(sorted_zyx, subset, counts, goffsets, n_rows_total, P, max_ps_dist=2,
                 include_single=True):
    """Pair rows of the patches in `subset` (device int64 [m], ascending) written compactly, in
    the canonical order, with their GLOBAL row ids: rows int32 [r (+ m), 6], ids int64 [r (+ m)].
    counts / goffsets: device int64 [n] (partners per patch, exclusive scan over all patches)."""
    torch = _torch()
    n, m = int(sorted_zyx.shape[0]), int(subset.numel())
    if m == 0:
        return None, None
    local_counts = counts[subset]
    ends = torch.cumsum(local_counts, 0)
    n_local = int(ends[-1].item())
    total = n_local + (m if include_single else 0)
    if total == 0:
        return None, None
    local_off = (ends - local_counts).contiguous()
    rows = torch.empty((total, 6), dtype=torch.int32, device=sorted_zyx.device)
    gid = torch.empty((total,), dtype=torch.int64, device=sorted_zyx.device)
    with _timed("patch_pairs"):
        check(lib().ppp_patch_pairs_fill_subset(
            _dev_ptr(sorted_zyx), n, int(max_ps_dist), _dev_ptr(subset.contiguous()), m,
            _dev_ptr(local_off), _dev_ptr(goffsets.contiguous()), n_local, int(n_rows_total),
            1 if include_single else 0, _dev_ptr(rows), _dev_ptr(gid), ctypes.byref(P), _stream()))
    return rows, gid


class LabelState:
    """Streaming union-find over the selected patches (ppp_label_begin / _add / _finish).
    The workspace is a torch tensor, so the node entries of its four volumes (parent u32,
    has-positive-edge u32, firstpos u64, key u64, all indexed by linear voxel index) can be
    read and written with torch indexing when ranks merge their forests."""

    def __init__(self, nodes, P):
        torch = _torch()
        self.torch, self.P, self.nodes = torch, P, nodes.contiguous()
        self.n = int(nodes.shape[0])
        V = int(P.Z) * int(P.Y) * int(P.X)
        self.V = V
        nbytes = int(lib().ppp_label_workspace_bytes(ctypes.byref(P)))
        self.work = torch.empty((nbytes,), dtype=torch.uint8, device=nodes.device)
        self.parent = self.work[:4 * V].view(torch.int32)
        self.haspos = self.work[4 * V:8 * V].view(torch.int32)
        self.firstpos = self.work[8 * V:16 * V].view(torch.int64)
        n64 = nodes.to(torch.int64)
        self.lin = (n64[:, 0] * int(P.Y) + n64[:, 1]) * int(P.X) + n64[:, 2]
        with _timed("label_components"):
            check(lib().ppp_label_begin(_dev_ptr(self.nodes), self.n, _dev_ptr(self.work),
                                        ctypes.byref(P), _stream()))

    def add(self, rows, aff, gid=None, first_id=0):
        n = int(rows.shape[0])
        if n == 0:
            return
        with _timed("label_components"):
            check(lib().ppp_label_add(_dev_ptr(rows), _dev_ptr(aff), _dev_ptr(gid), int(first_id), n,
                                      _dev_ptr(self.work), ctypes.byref(self.P), _stream()))

    def export(self):
        """(parent, firstpos, haspos) of the nodes: int64 [n] linear voxel index of the node's
        parent, int64 [n], int32 [n]."""
        return (self.parent[self.lin].to(self.torch.int64) & 0xFFFFFFFF,
                self.firstpos[self.lin].clone(), self.haspos[self.lin].clone())

    def merge(self, parents, firstpos, haspos):
        """parents: int64 [k, n] forests of k ranks (this rank's included or not); firstpos /
        haspos: the element-wise MIN / MAX over all ranks."""
        for row in parents.reshape(-1, self.n):
            with _timed("label_components"):
                check(lib().ppp_label_union_edges(_dev_ptr(self.lin.contiguous()),
                                                  _dev_ptr(row.contiguous()), self.n,
                                                  _dev_ptr(self.work), ctypes.byref(self.P), _stream()))
        self.firstpos[self.lin] = firstpos
        self.haspos[self.lin] = haspos.to(self.torch.int32)

    def finish(self):
        """int64 [n]: order key of the node's component, NONE_KEY64 = not in a component."""
        keys = self.torch.empty((self.n,), dtype=self.torch.int64, device=self.nodes.device)
        if self.n:
            with _timed("label_components"):
                check(lib().ppp_label_finish(_dev_ptr(self.nodes), self.n, _dev_ptr(keys),
                                             _dev_ptr(self.work), ctypes.byref(self.P), _stream()))
        return keys


def patch_graph(pred, cons, pairs, P, order=None):
    """S5.  pairs: device uint32-as-int32 [N, 6]; order: optional device int32 permutation
    (see pair_order); returns float32 [N]."""
    torch = _torch()
    n = int(pairs.shape[0])
    aff = torch.zeros((n,), dtype=torch.float32, device=pred.device)
    with _timed("patch_graph"):
        check(lib().ppp_patch_graph(_dev_ptr(pred), pred_dtype_code(pred), _dev_ptr(cons),
                                    _dev_ptr(pairs), _dev_ptr(order), n, _dev_ptr(aff),
                                    ctypes.byref(P), _stream()))
    return aff


def label_components(pairs, aff, nodes, P):
    """S6 (components).  nodes: device int32 [K, 3]; returns int64 [K] order keys
    (NONE_KEY = not in a component)."""
    torch = _torch()
    n = 0 if pairs is None else int(pairs.shape[0])
    k = int(nodes.shape[0])
    keys = torch.empty((k,), dtype=torch.int32, device=nodes.device)
    nbytes = int(lib().ppp_label_workspace_bytes(ctypes.byref(P)))
    work = torch.empty((nbytes,), dtype=torch.uint8, device=nodes.device)
    with _timed("label_components"):
        check(lib().ppp_label_components(_dev_ptr(pairs), _dev_ptr(aff), n, _dev_ptr(nodes), k,
                                         _dev_ptr(keys), _dev_ptr(work), ctypes.byref(P),
                                         _stream()))
    return keys.to(torch.int64) & 0xFFFFFFFF


def paint_instances(pred, nodes, labels, instances, P):
    """S6 (paint).  nodes int32 [K, 3], labels int32 [K], instances int32 (Z, Y, X) in place."""
    with _timed("paint_instances"):
        check(lib().ppp_paint_instances(_dev_ptr(pred), pred_dtype_code(pred), _dev_ptr(nodes),
                                        _dev_ptr(labels), int(nodes.shape[0]),
                                        _dev_ptr(instances), ctypes.byref(P), _stream()))
    return instances


# ---- the reference's NumPy-semantics stages (cuda=False) -------------------------------------------
def np_consensus(pred, foreground, P):
    """create_consensus_array (consensus_array.py:18-68) on the device: int16 votes
    [planes, Z, Y, X] (plane 0 = zero offset, plane q = COMPACT plane q - 1)."""
    torch = _torch()
    planes = int(lib().ppp_np_vote_planes(ctypes.byref(P)))
    votes = torch.empty((planes, P.Z, P.Y, P.X), dtype=torch.int16, device=pred.device)
    with _timed("np_consensus"):
        check(lib().ppp_np_consensus(_dev_ptr(pred), pred_dtype_code(pred), _dev_ptr(foreground), _dev_ptr(votes),
                                     ctypes.byref(P), _stream()))
    return votes


def np_rank_patches(pred, foreground, votes, P):
    """rank_patches (ranked_patches.py:76-105) on the device: int32 scores (Z, Y, X)."""
    torch = _torch()
    score = torch.empty((P.Z, P.Y, P.X), dtype=torch.int32, device=pred.device)
    with _timed("np_rank_patches"):
        check(lib().ppp_np_rank_patches(_dev_ptr(pred), pred_dtype_code(pred), _dev_ptr(foreground), _dev_ptr(votes),
                                        _dev_ptr(score), ctypes.byref(P), _stream()))
    return score


def np_patch_graph(pred, mask, votes, rows, P):
    """computePatchGraph's NumPy branch (aff_patch_graph.py:209-282) for candidate rows int32 [n, 6]:
    (weight int64 [n], count int32 [n])."""
    torch = _torch()
    n = int(rows.shape[0])
    weight = torch.zeros((n,), dtype=torch.int64, device=pred.device)
    count = torch.zeros((n,), dtype=torch.int32, device=pred.device)
    if n:
        with _timed("np_patch_graph"):
            check(lib().ppp_np_patch_graph(_dev_ptr(pred), pred_dtype_code(pred), _dev_ptr(mask), _dev_ptr(votes),
                                           _dev_ptr(rows), n, _dev_ptr(weight), _dev_ptr(count), ctypes.byref(P),
                                           _stream()))
    return weight, count


def paint_patch_rows(rows, nodes, labels, instances, P):
    """S6 (paint) from a patch table: rows float16 / float32 [K, C] (row k = pred[:, node k]),
    nodes int32 [K, 3], labels int32 [K], instances int32 (Z, Y, X) in place."""
    assert rows.is_contiguous() and rows.shape[0] == nodes.shape[0]
    with _timed("paint_instances"):
        check(lib().ppp_paint_patch_rows(_dev_ptr(rows), pred_dtype_code(rows), _dev_ptr(nodes),
                                         _dev_ptr(labels), int(nodes.shape[0]),
                                         _dev_ptr(instances), ctypes.byref(P), _stream()))
    return instances


def direct_voxel_major(P):
    """S1 can write the voxel-major layout itself for these parameters."""
    Pv = P.copy()
    Pv.cons_layout = CONS_VOXEL_MAJOR
    return os.environ.get("PPP_S1_DIRECT_VM", "1") != "0" and \
        lib().ppp_consensus_writes_voxel_major(ctypes.byref(Pv)) == 1


def consensus_voxel_major(pred, overlap, P, out=None, open_rows=False):
    """S1 straight into the symmetric voxel-major layout when the library can do that for these
    parameters (ppp_consensus_writes_voxel_major), else COMPACT + ppp_cons_to_voxel_major.
    Returns (tensor [bz, by, bx, W], params with cons_layout = VOXEL_MAJOR).  out: see consensus."""
    Pv = P.copy()
    Pv.cons_layout = CONS_VOXEL_MAJOR
    if direct_voxel_major(P):
        return consensus(pred, overlap, Pv, out=out, open_rows=open_rows), Pv
    Pc = P.copy()
    Pc.cons_layout = CONS_COMPACT
    cons = consensus(pred, overlap, Pc)
    return cons_to_voxel_major(cons, Pc)


def consensus_part(pred, overlap, P, part, out):
    """S1 for the base voxels of `part` (z0, y0, x0, z1, y1, x1; inside P.cons_box) written into
    `out`, a buffer indexed by the whole P.cons_box (COMPACT planes or open VOXEL_MAJOR rows)."""
    b = Box(*[int(v) for v in part])
    note_add("s1_base_voxels", int(np.prod(b.shape())))
    P = with_pred_clean(pred, P)
    with _timed("consensus"):
        check(lib().ppp_consensus_part(_dev_ptr(pred), pred_dtype_code(pred), _dev_ptr(overlap), _dev_ptr(out),
                                       ctypes.byref(P), ctypes.byref(b), _stream()))
    note("s1_kernel", lib().ppp_consensus_kernel_name().decode())


def cons_planes_to_rows(planes, planes_box, P, out=None):
    """VOXEL_MAJOR rows of P.cons_box cut from COMPACT planes indexed by `planes_box`
    (ppp_cons_planes_to_rows).  Returns (tensor [bz, by, bx, W], params with VOXEL_MAJOR)."""
    Pv = P.copy()
    Pv.cons_layout = CONS_VOXEL_MAJOR
    W = (2 * P.pz - 1) * (2 * P.py - 1) * (2 * P.px - 1)
    shape = Pv.cons_box.shape() + (W,)
    n_el = int(np.prod(shape))
    rows = out[:n_el].view(shape) if out is not None and out.numel() >= n_el else _big_empty(shape, planes.device)
    b = Box(*[int(v) for v in planes_box])
    with _timed("cons_planes_to_rows"):
        check(lib().ppp_cons_planes_to_rows(_dev_ptr(planes), ctypes.byref(b), _dev_ptr(rows), ctypes.byref(Pv),
                                            _stream()))
    return rows, Pv


def cons_to_reference(cons_compact, P):
    torch = _torch()
    shape = (2 * P.pz if P.pz > 1 else 1, 2 * P.py, 2 * P.px, P.Z, P.Y, P.X)
    ref = torch.empty(shape, dtype=torch.float32, device=cons_compact.device)
    check(lib().ppp_cons_to_reference(_dev_ptr(cons_compact), _dev_ptr(ref), ctypes.byref(P),
                                      _stream()))
    return ref


def cons_to_voxel_major(cons_compact, P):
    """Re-layout for ppp_patch_graph.  Returns (tensor [bz, by, bx, W], params with
    cons_layout = VOXEL_MAJOR)."""
    torch = _torch()
    W = (2 * P.pz - 1) * (2 * P.py - 1) * (2 * P.px - 1)
    vm = _big_empty(P.cons_box.shape() + (W,), cons_compact.device)
    Pc = P.copy()
    Pc.cons_layout = CONS_COMPACT
    with _timed("cons_to_voxel_major"):
        check(lib().ppp_cons_to_voxel_major(_dev_ptr(cons_compact), _dev_ptr(vm), ctypes.byref(Pc),
                                            _stream()))
    Pv = P.copy()
    Pv.cons_layout = CONS_VOXEL_MAJOR
    return vm, Pv


def patch_bits(pred, centres, thresh, P, scratch=None):
    """Bit r of row k = (pred[r][centre k] > float32(thresh)).  centres int32 [n, 3] on the
    device; returns int32 [n, ceil(C/32)] on the device.  scratch: a flat int32 device buffer the
    dense path may carve its per-voxel table and the result from (the caller's idle consensus
    pool: the two are 23 GB at 512^3 / 9^3)."""
    torch = _torch()
    n = int(centres.shape[0])
    words = (P.pz * P.py * P.px + 31) // 32
    V = int(P.Z) * int(P.Y) * int(P.X)
    dense = n * 16 >= V and os.environ.get("PPP_PATCH_BITS", "auto") != "sparse"
    vol = None
    if dense:
        # many centres (the cover candidates): one coalesced pass over the prediction for all
        # voxels, then a row gather -- instead of one cache line per (centre, channel)
        if scratch is not None and scratch.numel() >= V * words:
            vol = scratch[:V * words].view(V, words)
            scratch = scratch[V * words:]
        else:
            scratch = None
            try:
                vol = torch.empty((V, words), dtype=torch.int32, device=pred.device)
            except RuntimeError:          # no room for the per-voxel table: per-centre gathers
                dense = False
    if dense:
        with _timed("patch_bits"):
            check(lib().ppp_patch_bits_volume(_dev_ptr(pred), pred_dtype_code(pred), float(thresh),
                                              _dev_ptr(vol), ctypes.byref(P), _stream()))
            c = centres.to(torch.int64)
            lin = (c[:, 0] * int(P.Y) + c[:, 1]) * int(P.X) + c[:, 2]
            del c
            if scratch is not None and scratch.numel() >= n * words:
                bits = torch.index_select(vol, 0, lin, out=scratch[:n * words].view(n, words))
            else:
                bits = vol[lin]
        del vol
        return bits
    bits = torch.empty((n, words), dtype=torch.int32, device=pred.device)
    with _timed("patch_bits"):
        check(lib().ppp_patch_bits(_dev_ptr(pred), pred_dtype_code(pred), _dev_ptr(centres), n,
                                   float(thresh), _dev_ptr(bits), ctypes.byref(P), _stream()))
    return bits


def cover_pass_device(mask, bits, lin, state, pix_th, P, bits_first_voxel=None):
    """One pass of the greedy cover on the device (foreground_cover.py:111-180 without the stop
    rule, see ppp_cover_pass).  mask uint8 (Z,Y,X) is cleared in place; state int32 [n]
    (0 = takes part; ends 1 selected / 2 not) is updated in place.  bits: int32 [n, words] in
    list order, or -- bits_first_voxel given -- a table with a row per voxel whose first row
    belongs to that linear voxel index (ppp_cover_pass_voxel_bits).
    Returns (cleared int32 [n], rounds)."""
    torch = _torch()
    n = int(state.numel())
    cleared = torch.empty(n, dtype=torch.int32, device=mask.device)
    nbytes = int(lib().ppp_cover_workspace_bytes(n, ctypes.byref(P)))
    check(min(nbytes, 0))
    work = torch.empty(nbytes, dtype=torch.uint8, device=mask.device)
    rounds = ctypes.c_int32(0)
    with _timed("cover"):
        if bits_first_voxel is None:
            check(lib().ppp_cover_pass(_dev_ptr(mask), _dev_ptr(bits), _dev_ptr(lin), n, int(pix_th),
                                       _dev_ptr(state), _dev_ptr(cleared), _dev_ptr(work),
                                       ctypes.byref(P), _stream(), ctypes.byref(rounds)))
        else:
            check(lib().ppp_cover_pass_voxel_bits(_dev_ptr(mask), _dev_ptr(bits), int(bits_first_voxel),
                                                  _dev_ptr(lin), n, int(pix_th), _dev_ptr(state),
                                                  _dev_ptr(cleared), _dev_ptr(work), ctypes.byref(P),
                                                  _stream(), ctypes.byref(rounds)))
    return cleared, int(rounds.value)


def thin_cover_device(mask, bits, lin, P):
    """Set-cover thinning on the device (ppp_thin_cover; foreground_cover.py:183-256).
    mask uint8 (Z,Y,X) device tensor = mask_to_cover (not modified), bits int32 [n, words] /
    lin int64 [n]: the selected patches in list order.  Returns keep, bool [n] device tensor."""
    torch = _torch()
    n = int(lin.numel())
    keep = torch.zeros(max(n, 1), dtype=torch.uint8, device=mask.device)
    if n == 0:
        return keep[:0].bool()
    nbytes = int(lib().ppp_thin_workspace_bytes(n, ctypes.byref(P)))
    check(min(nbytes, 0))
    work = torch.empty(nbytes, dtype=torch.uint8, device=mask.device)
    rounds = ctypes.c_int32(0)
    with _timed("thin_cover"):
        check(lib().ppp_thin_cover(_dev_ptr(mask), _dev_ptr(bits.contiguous()), _dev_ptr(lin.contiguous()),
                                   n, _dev_ptr(keep), _dev_ptr(work), ctypes.byref(P), _stream(),
                                   ctypes.byref(rounds)))
    note("thin_rounds", rounds.value)
    return keep[:n].bool()


class CoverShard:
    """One rank's share of the greedy cover rounds (ppp_cover_open / _step / _zone / _close):
    local mask uint8 (Zl, Y, X) device tensor (own slices + halo), the own ranked patches (local
    linear indices, global ranks, local bit table), local params with origin_z = first slice."""
    COUNT, FILTER, SELECT = 0, 1, 2

    def __init__(self, mask, lin_local, rank_id, bits, P, global_z):
        torch = _torch()
        self.torch, self.P, self.mask = torch, P, mask
        self.lin, self.rank_id, self.bits = lin_local.contiguous(), rank_id.contiguous(), bits
        self.n = int(lin_local.numel())
        self.gz = int(global_z)
        nbytes = int(lib().ppp_cover_workspace_bytes(self.n, ctypes.byref(P)))
        self.work = torch.empty((nbytes,), dtype=torch.uint8, device=mask.device)
        self.state = self.cleared = None

    def open(self, state):
        """state int32 [n]: 0 takes part, 1 selected in an earlier pass, 2 never."""
        self.state = state.contiguous()
        self.cleared = self.torch.empty(max(self.n, 1), dtype=self.torch.int32, device=self.mask.device)
        with _timed("cover"):
            check(lib().ppp_cover_open(_dev_ptr(self.mask), _dev_ptr(self.lin), _dev_ptr(self.rank_id),
                                       self.n, _dev_ptr(self.state), _dev_ptr(self.cleared),
                                       _dev_ptr(self.work), ctypes.byref(self.P), _stream()))

    def step(self, what, pix_th=0):
        with _timed("cover"):
            check(lib().ppp_cover_step(int(what), _dev_ptr(self.bits), int(pix_th), _dev_ptr(self.state),
                                       _dev_ptr(self.cleared), _dev_ptr(self.work), self.gz,
                                       ctypes.byref(self.P), _stream()))

    def alive(self):
        a = ctypes.c_int32(0)
        check(lib().ppp_cover_alive(_dev_ptr(self.work), ctypes.byref(self.P), _stream(), ctypes.byref(a)))
        return a.value != 0

    def zone(self, imp, z_lo, z_hi, own, rank=None, mask=None, clean=None):
        with _timed("cover"):
            check(lib().ppp_cover_zone(1 if imp else 0, _dev_ptr(self.work), int(z_lo), int(z_hi),
                                       int(own[0]), int(own[1]), _dev_ptr(rank), _dev_ptr(mask),
                                       _dev_ptr(clean), ctypes.byref(self.P), _stream()))

    def close(self):
        with _timed("cover"):
            check(lib().ppp_cover_close(_dev_ptr(self.mask), _dev_ptr(self.work), ctypes.byref(self.P),
                                        _stream()))


class ThinShard:
    """One rank's share of the set-cover thinning rounds (ppp_thin_open / _step / _zone / _close; round 6):
    local mask uint8 (Zl, Y, X) device tensor (own slices + halo), the own selected patches (local linear
    indices, positions in the global selected list, local bit table), local params with origin_z = first
    slice.  After the rounds: state (1 = kept), count (voxels covered when kept), cleared (interior ones)."""
    COUNT, FILTER, SELECT = 0, 1, 2

    def __init__(self, mask, lin_local, index_global, bits, P, global_z):
        torch = _torch()
        self.P, self.mask, self.bits = P, mask, bits
        self.lin, self.index = lin_local.contiguous(), index_global.to(torch.int32).contiguous()
        self.n = int(lin_local.numel())
        self.gz = int(global_z)
        nbytes = int(lib().ppp_thin_shard_workspace_bytes(ctypes.byref(P)))
        check(min(nbytes, 0))
        dev = mask.device
        self.work = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        self.state = torch.empty(max(self.n, 1), dtype=torch.int32, device=dev)
        self.count = torch.empty(max(self.n, 1), dtype=torch.int32, device=dev)
        self.cleared = torch.empty(max(self.n, 1), dtype=torch.int32, device=dev)
        with _timed("thin_cover"):
            check(lib().ppp_thin_open(_dev_ptr(self.mask), _dev_ptr(self.lin), _dev_ptr(self.index), self.n,
                                      _dev_ptr(self.state), _dev_ptr(self.count), _dev_ptr(self.cleared),
                                      _dev_ptr(self.work), ctypes.byref(self.P), _stream()))

    def step(self, what):
        with _timed("thin_cover"):
            check(lib().ppp_thin_step(int(what), _dev_ptr(self.bits), _dev_ptr(self.state), _dev_ptr(self.count),
                                      _dev_ptr(self.cleared), _dev_ptr(self.work), self.gz, ctypes.byref(self.P),
                                      _stream()))

    def alive(self):
        a = ctypes.c_int32(0)
        check(lib().ppp_thin_alive(_dev_ptr(self.work), ctypes.byref(self.P), _stream(), ctypes.byref(a)))
        return a.value != 0

    def zone(self, imp, z_lo, z_hi, own, key=None, mask=None, clean=None):
        with _timed("thin_cover"):
            check(lib().ppp_thin_zone(1 if imp else 0, _dev_ptr(self.work), int(z_lo), int(z_hi), int(own[0]),
                                      int(own[1]), _dev_ptr(key), _dev_ptr(mask), _dev_ptr(clean),
                                      ctypes.byref(self.P), _stream()))

    def close(self):
        with _timed("thin_cover"):
            check(lib().ppp_thin_close(_dev_ptr(self.mask), _dev_ptr(self.work), ctypes.byref(self.P), _stream()))


def synth_pred(labels, P, seed=0, hi=0.95, lo=0.05, noise=0.04, f16=True, voxel_offset=0):
    """Procedural prediction volume on the device (bench / tests).  voxel_offset: linear index
    of local voxel 0 in the global volume when `labels` is a z-slab of a larger volume."""
    torch = _torch()
    C = P.pz * P.py * P.px
    pred = torch.empty((C,) + P.shape, dtype=torch.float16 if f16 else torch.float32,
                       device=labels.device)
    check(lib().ppp_synth_pred(_dev_ptr(labels), _dev_ptr(pred), F16 if f16 else F32,
                               int(seed) & 0xFFFFFFFF, hi, lo, noise, int(voxel_offset),
                               ctypes.byref(P), _stream()))
    return pred


def decode_tail(x, w1, b1, w2, b2, w3, b3, dst, pred, patchshape):
    """ppp_decode_tail: x float32 [n, 64, 4, 4, 4] (decoder features), dst int64 [n] voxel indices,
    pred (C, ...) float16 / float32 device block -- written in place at pred[:, dst]."""
    torch = _torch()
    n = int(x.shape[0])
    if n == 0:
        return pred
    P = Params()
    P.abi_version = ABI_VERSION
    vol = [int(v) for v in pred.shape[1:]]
    while len(vol) < 3:
        vol = [1] + vol
    P.Z, P.Y, P.X = vol
    P.pz, P.py, P.px = [int(p) for p in patchshape]
    P.th = P.thi = 0.5
    P.bg_rule, P.value_rule = BG_LESS_THAN_TH, VAL_COUNT
    P.cons_box = Box(0, 0, 0, P.Z, P.Y, P.X)
    f32 = lambda t: t.detach().to(torch.float32).contiguous()
    x, w1, w2, w3 = f32(x), f32(w1).reshape(-1), f32(w2).reshape(-1), f32(w3).reshape(-1)
    check(lib().ppp_decode_tail(_dev_ptr(x), n, int(x.shape[1]), int(x.shape[2]), _dev_ptr(w1), float(b1),
                                _dev_ptr(w2), float(b2), _dev_ptr(w3), float(b3),
                                _dev_ptr(dst.to(torch.int64).contiguous()), _dev_ptr(pred),
                                pred_dtype_code(pred), ctypes.byref(P), _stream()))
    return pred


def synth_pred_box(labels, label_box, box, gshape, patchshape, flags, seed=0, hi=0.95, lo=0.05,
                   noise=0.04, f16=True):
    """ppp_synth_pred for a box (z0, z1, y0, y1, x0, x1) of a volume of shape gshape.  labels:
    int32 device tensor over label_box (z0, z1, y0, y1, x0, x1), the box grown by the patch radius
    (clipped).  Returns (C, bz, by, bx)."""
    torch = _torch()
    z0, z1, y0, y1, x0, x1 = [int(v) for v in box]
    P = make_params((z1 - z0, y1 - y0, x1 - x0), patchshape, origin=(z0, y0, x0), **flags)
    C = P.pz * P.py * P.px
    pred = torch.empty((C,) + P.shape, dtype=torch.float16 if f16 else torch.float32, device=labels.device)
    lb = np.ascontiguousarray([label_box[0], label_box[2], label_box[4], label_box[1], label_box[3],
                               label_box[5]], dtype=np.int32)
    gd = np.ascontiguousarray(gshape, dtype=np.int32)
    check(lib().ppp_synth_pred_box(_dev_ptr(labels.contiguous()), _np_ptr(lb), _dev_ptr(pred),
                                   F16 if f16 else F32, int(seed) & 0xFFFFFFFF, hi, lo, noise,
                                   _np_ptr(gd), ctypes.byref(P), _stream()))
    return pred


# ----------------------------------------------------------------------------------------
# host stages (NumPy arrays in, NumPy arrays out)
# ----------------------------------------------------------------------------------------
def _np_ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _i32(seq):
    return np.ascontiguousarray(np.array(seq, dtype=np.int32))


def host_mws(pairs, aff, shape):
    """Mutex watershed on the patch graph (ppp_host_mws).  pairs uint32 [n, 6], aff float32 [n]
    (host arrays).  Returns (nodes int32 [k, 3], labels int64 [k], n_labels): the nodes that
    ended up in a component and their instance ids; n_labels = number of ids issued."""
    pairs = np.ascontiguousarray(np.asarray(pairs).reshape(-1, 6), dtype=np.uint32)
    aff = np.ascontiguousarray(aff, dtype=np.float32)
    n = len(aff)
    cap = min(2 * n, int(np.prod(shape))) + 1
    nodes = np.empty((cap, 3), dtype=np.int32)
    labels = np.empty((cap,), dtype=np.int32)
    n_labels = ctypes.c_int64(0)
    k = int(lib().ppp_host_mws(_np_ptr(pairs), _np_ptr(aff), n, _np_ptr(_i32(shape)), _np_ptr(nodes),
                               _np_ptr(labels), cap, ctypes.byref(n_labels)))
    if k < 0:
        raise RuntimeError("libppp_mi355x: ppp_host_mws: %s" %
                           ("node capacity too small" if k == -1 else "coordinate outside the volume"))
    keep = labels[:k] > 0
    return nodes[:k][keep], labels[:k][keep].astype(np.int64), int(n_labels.value)


def mws_labels_device(rows, aff, nodes, P):
    """Mutex watershed of the patch graph (graph_mws.py:7-85) with everything but the inherently
    sequential loop on the device: ppp_mws_edges filters the rows with aff != 0, puts them in
    networkx's edge order and sorts them stably by |aff| (descending); the edge list (8 bytes per
    edge) crosses to the host once and ppp_host_mws_sorted walks it.
    rows int32 [N, 6] / aff float32 [N] device tensors without repeated node pairs, nodes int32
    [K, 3] device.  Returns (labels int32 [K] device tensor, 0 = no component; ids issued)."""
    torch = _torch()
    n, k = int(rows.shape[0]), int(nodes.shape[0])
    labels = np.zeros((k,), dtype=np.int32)
    if n == 0 or k == 0:
        return torch.from_numpy(labels).to(nodes.device), 0
    nbytes = int(lib().ppp_mws_edges_workspace_bytes(n, k, ctypes.byref(P)))
    check(min(nbytes, 0))
    work = torch.empty(nbytes, dtype=torch.uint8, device=rows.device)
    eu = torch.empty((n,), dtype=torch.int32, device=rows.device)
    ev = torch.empty((n,), dtype=torch.int32, device=rows.device)
    n_edges = ctypes.c_int64(0)
    with host_timer("s6a_mws_edges"):
        with _timed("mws_edges"):
            check(lib().ppp_mws_edges(_dev_ptr(rows), _dev_ptr(aff), n, _dev_ptr(nodes.contiguous()), k,
                                      _dev_ptr(eu), _dev_ptr(ev), ctypes.byref(n_edges), _dev_ptr(work),
                                      ctypes.byref(P), _stream()))
        ne = int(n_edges.value)
        eu_h = eu[:ne].cpu().numpy()
        ev_h = ev[:ne].cpu().numpy()
    del work, eu, ev
    with host_timer("s6b_mws_loop"):
        issued = int(lib().ppp_host_mws_sorted(_np_ptr(eu_h), _np_ptr(ev_h), ne, k, _np_ptr(labels)))
    note("mws_edges", ne)
    return torch.from_numpy(labels).to(nodes.device), issued


def host_rank_order(score, foreground, patchshape):
    score = np.ascontiguousarray(score, dtype=np.float32)
    fg = np.ascontiguousarray(foreground).astype(np.uint8)
    out = np.empty(score.size, dtype=np.int64)
    vol, ps = _i32(score.shape), _i32(patchshape)
    n = lib().ppp_host_rank_order(_np_ptr(score), _np_ptr(fg), _np_ptr(vol), _np_ptr(ps),
                                  _np_ptr(out))
    return out[:n].copy()


def padded_mask(mask):
    """uint8 0/1 copy of a (Z,Y,X) mask whose buffer has 8 spare bytes at the end, as the native
    cover functions require; returns (volume view, owner)."""
    flat = np.zeros(mask.size + 8, dtype=np.uint8)
    flat[:mask.size] = (np.asarray(mask).reshape(-1) != 0)
    return flat[:mask.size].reshape(mask.shape), flat


def rank_order_device(score, foreground, patchshape, to_host=True):
    """all_patches + rank_patches_by_score on the device (ppp_rank_order): interior foreground
    voxels in raster order, stably sorted by score descending (vote_instances.py:276,286-287,
    ranked_patches.py:21-30).  score: device float32 (Z,Y,X); foreground: host bool.
    Returns (lin int64, scores float32), NumPy arrays or (to_host=False) device tensors."""
    torch = _torch()
    Z, Y, X = [int(v) for v in score.shape]
    P = Params()
    P.abi_version = ABI_VERSION
    P.Z, P.Y, P.X = Z, Y, X
    P.pz, P.py, P.px = [int(p) for p in patchshape]
    P.th = P.thi = 0.5
    P.bg_rule, P.value_rule = BG_LESS_THAN_TH, VAL_COUNT
    P.cons_box = Box(0, 0, 0, Z, Y, X)
    if torch.is_tensor(foreground):
        fg = foreground.to(score.device)
        fg = (fg if fg.dtype == torch.uint8 else (fg != 0).to(torch.uint8)).contiguous().reshape(-1)
    else:
        fg = torch.from_numpy(np.ascontiguousarray(np.asarray(foreground) != 0).astype(np.uint8)
                              ).to(score.device).reshape(-1)
    V = Z * Y * X
    nbytes = int(lib().ppp_rank_order_workspace_bytes(ctypes.byref(P)))
    check(min(nbytes, 0))
    work = torch.empty(nbytes, dtype=torch.uint8, device=score.device)
    # the ranked list has at most one entry per interior voxel
    cap = max(1, (Z - 2 * (P.pz // 2)) * (Y - 2 * (P.py // 2)) * (X - 2 * (P.px // 2)))
    lin = torch.empty((min(cap, V),), dtype=torch.int64, device=score.device)
    sc = torch.empty((min(cap, V),), dtype=torch.float32, device=score.device)
    count = ctypes.c_int64(0)
    with _timed("rank_order"):
        check(lib().ppp_rank_order(_dev_ptr(score.contiguous()), _dev_ptr(fg), _dev_ptr(lin), _dev_ptr(sc),
                                   ctypes.byref(count), _dev_ptr(work), ctypes.byref(P), _stream()))
    n = int(count.value)
    lin, sc = lin[:n], sc[:n]
    if to_host:
        return lin.cpu().numpy(), sc.cpu().numpy()
    return lin, sc


def host_cover_pass(mask_running, overlap, patchshape, ranked_lin, ranked_score, bits, pix_th,
                    score_threshold, selected, remaining, marked=None):
    """In place on mask_running (uint8), selected (uint8) and -- mark_close_neighboorhood --
    marked (uint8 volume); returns (new `remaining`, stopped-by-score-threshold)."""
    vol, ps = _i32(mask_running.shape), _i32(patchshape)
    rem = ctypes.c_int64(int(remaining))
    stopped = ctypes.c_int32(0)
    thr = float("nan") if score_threshold is None else float(score_threshold)
    lib().ppp_host_cover_pass_marked(_np_ptr(mask_running), _np_ptr(overlap), _np_ptr(vol), _np_ptr(ps),
                                     _np_ptr(ranked_lin), _np_ptr(ranked_score), _np_ptr(bits),
                                     len(ranked_lin), int(pix_th), thr, _np_ptr(selected),
                                     ctypes.byref(rem), ctypes.byref(stopped),
                                     None if marked is None else _np_ptr(marked))
    return rem.value, bool(stopped.value)


def host_thin_cover(mask, patchshape, sel_lin, bits):
    vol, ps = _i32(mask.shape), _i32(patchshape)
    keep = np.zeros(len(sel_lin), dtype=np.uint8)
    lib().ppp_host_thin_cover(_np_ptr(mask), _np_ptr(vol), _np_ptr(ps), _np_ptr(sel_lin),
                              _np_ptr(bits), len(sel_lin), _np_ptr(keep))
    return keep.astype(bool)


def host_skeletonize_3d(mask):
    """3-d thinning of a (Z, Y, X) (or (Y, X)) mask: bool array of the skeleton
    (ppp_host_skeletonize_3d; Lee / Kashyap / Chu 1994, what skimage's skeletonize_3d implements)."""
    m = np.ascontiguousarray(np.asarray(mask) != 0).astype(np.uint8)
    shape = m.shape
    m3 = m.reshape((1,) * (3 - m.ndim) + shape)
    out = np.empty_like(m3)
    if lib().ppp_host_skeletonize_3d(_np_ptr(m3), _np_ptr(_i32(m3.shape)), _np_ptr(out)) < 0:
        raise RuntimeError("libppp_mi355x: ppp_host_skeletonize_3d: bad arguments")
    return out.reshape(shape).astype(bool)


def host_patch_pairs(sel_zyx, patchshape, max_ps_dist=2, include_single=True):
    sel = np.ascontiguousarray(np.asarray(sel_zyx).reshape(-1, 3), dtype=np.int32)
    ps = _i32(patchshape)
    n = len(sel)
    sorted_zyx = np.empty((n, 3), dtype=np.int32)
    rows = lib().ppp_host_patch_pairs(_np_ptr(sel), n, _np_ptr(ps), int(max_ps_dist),
                                      1 if include_single else 0, _np_ptr(sorted_zyx), None)
    if rows == 0:
        return sorted_zyx, None
    pairs = np.empty((rows, 6), dtype=np.uint32)
    lib().ppp_host_patch_pairs(_np_ptr(sel), n, _np_ptr(ps), int(max_ps_dist),
                               1 if include_single else 0, _np_ptr(sorted_zyx), _np_ptr(pairs))
    return sorted_zyx, pairs
