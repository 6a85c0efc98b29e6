#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

This script only works in the development container, where the upstream tree is
mounted read-only at /root/reference.  Nothing from that tree is copied: the
reference's Python modules are imported in place (with stub modules for the
packages this image lacks) and its device kernels are compiled *where they lie*
(their text is produced by the reference's own ``loadKernelFromFile`` templating) as
plain C++ for the host, through a tiny CUDA-keyword shim (SURVEY.md section 8c,
"recipe B", and Appendix B).  Outputs are DATA: seeded synthetic inputs and the
reference's stage-by-stage results, written as small ``.npz`` files that the tests
load on machines where /root/reference does not exist.

Thread order: the shim's launcher runs the CUDA threads serially in global raster
order (z, y, x).  Any serial order is a legal CUDA execution; raster order is the
canonical one the oracle (oracle/) and the HIP kernels reproduce bit-for-bit.

Pair order: the reference's patch-pair list comes out of a Python ``set``
(vote_instances/aff_patch_graph.py:57,86).  The canonical order used for the
golden S5/S6 vectors is: pairs sorted by (i, j) where i < j index the x-sorted
selected list (same orientation as the reference), self pairs appended after; the
reference accepts this list through its ``selected_patch_pairs`` injection point
(vote_instances/vote_instances.py:400-406).  The reference's own set order is
stored too (``pairs_ref_order``) and is compared as a set.

Usage:  python tests/golden/gen_golden.py [case ...]
"""
import ctypes
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
import threading
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
REF_VI = os.path.join(REF, "PatchPerPix", "vote_instances")
sys.path.insert(0, REPO)

from patchperpix_amd import synth  # noqa: E402

# ----------------------------------------------------------------------------------
# 1. stub modules for packages the image lacks (import-time only, never called on
#    the cuda=True path with the flags used here)
# ----------------------------------------------------------------------------------


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _missing(*a, **k):
    raise RuntimeError("stubbed function called")


def install_stubs():
    _stub("h5py", File=_missing)
    _stub("zarr", open=_missing)
    sk = _stub("skimage")
    sk.io = _stub("skimage.io")
    sk.morphology = _stub("skimage.morphology", skeletonize_3d=_missing,
                          binary_dilation=_missing, ball=_missing)
    sk.draw = _stub("skimage.draw", line=_missing)
    pc = _stub("pycuda")
    pc.compiler = _stub("pycuda.compiler")
    if not hasattr(np, "product"):
        np.product = np.prod


# ----------------------------------------------------------------------------------
# 2. fake ``cuda_code`` module: the reference's device shim (vote_instances/
#    cuda_code.py:5-59) re-implemented on the host CPU
# ----------------------------------------------------------------------------------
SHIM = r"""
#include <cstdint>
#include <cmath>
#include <cstdlib>
#include <utility>
#include <type_traits>
#define __global__
#define __device__
struct dim3s { unsigned x,y,z; };
static dim3s blockIdx, blockDim, threadIdx;
template<typename T, typename U> static inline T atomicAdd(T* p, U v){ T o=*p; *p = o + (T)v; return o; }
static inline unsigned max(int a, unsigned b){ return (unsigned)a > b ? (unsigned)a : b; }
"""

LAUNCH_COMMON = r"""
template<class T> static typename std::enable_if<std::is_pointer<T>::value,T>::type cv(void*p){return (T)p;}
template<class T> static typename std::enable_if<!std::is_pointer<T>::value,T>::type cv(void*p){return *(T*)p;}
template<class... A, size_t... I> static void call(void(*f)(A...), void**a, std::index_sequence<I...>){
  f(cv<typename std::decay<A>::type>(a[I])...); }
// run every CUDA thread once, serially, in GLOBAL raster order (z, y, x)
template<class... A> static void run(void(*f)(A...), void**a, const unsigned*g, const unsigned*b){
  blockDim={b[0],b[1],b[2]};
  const unsigned nx=g[0]*b[0], ny=g[1]*b[1], nz=g[2]*b[2];
  for(unsigned z=0;z<nz;z++)for(unsigned y=0;y<ny;y++)for(unsigned x=0;x<nx;x++){
    blockIdx={x/b[0],y/b[1],z/b[2]}; threadIdx={x%b[0],y%b[1],z%b[2]};
    call(f,a,std::index_sequence_for<A...>{}); } }
"""

_BUILD_DIR = tempfile.mkdtemp(prefix="ppp_golden_")
_SO_CACHE = {}


class _Kernel:
    def __init__(self, lib, name):
        self.fn = getattr(lib, "launch_" + name)
        self.fn.restype = None

    def __call__(self, *args, block=None, grid=None):
        keep = []
        ptrs = (ctypes.c_void_p * len(args))()
        for i, a in enumerate(args):
            if isinstance(a, np.ndarray):
                assert a.flags["C_CONTIGUOUS"]
                ptrs[i] = a.ctypes.data
            else:  # numpy scalar, passed by address
                box = np.array([a])
                keep.append(box)
                ptrs[i] = box.ctypes.data
        g = (ctypes.c_uint * 3)(*[int(v) for v in grid])
        b = (ctypes.c_uint * 3)(*[int(v) for v in block])
        self.fn(ptrs, g, b)


class _Module:
    def __init__(self, lib):
        self.lib = lib

    def get_function(self, name):
        return _Kernel(self.lib, name)


def make_kernel(code, options=None):
    options = list(options or [])
    names = re.findall(r"__global__\s+void\s+(\w+)", code)
    src = SHIM + code + LAUNCH_COMMON
    for n in names:
        src += ('extern "C" void launch_%s(void**a,const unsigned*g,const unsigned*b)'
                "{ run(%s,a,g,b); }\n" % (n, n))
    key = hashlib.sha1((src + " ".join(options)).encode()).hexdigest()
    if key not in _SO_CACHE:
        cpp = os.path.join(_BUILD_DIR, key + ".cpp")
        so = os.path.join(_BUILD_DIR, key + ".so")
        with open(cpp, "w") as f:
            f.write(src)
        subprocess.check_call(["g++", "-O2", "-std=c++14", "-ffp-contract=off",
                               "-shared", "-fPIC", "-x", "c++", cpp, "-o", so]
                              + options)
        _SO_CACHE[key] = ctypes.CDLL(so)
    return _Module(_SO_CACHE[key])


class _Owner(np.ndarray):
    def free(self):
        pass


def alloc_zero_array(shape, dtype):
    if np.isscalar(shape):
        shape = (int(shape),)
    owner = np.ndarray.__new__(_Owner, shape, dtype)
    owner[...] = 0
    return owner.view(np.ndarray)


def install_fake_cuda_code():
    _stub("cuda_code", make_kernel=make_kernel, alloc_zero_array=alloc_zero_array,
          sync=lambda ctx: None, init_cuda=lambda: None,
          delete_cuda=lambda ctx: None, get_cuda_stream=lambda: None)


# ----------------------------------------------------------------------------------
# 3. stage-wise driver around the reference's own stage functions
# ----------------------------------------------------------------------------------
FLYLIGHT = dict(  # experiments/flylight/setups/setup01/default.toml:114-169 (+ [model])
    patch_threshold=0.5, fc_threshold=0.5, cuda=True, blockwise=False,
    select_patches_for_sparse_data=True, includeSinglePatchCCS=True,
    removeIntersection=False, mws=False, skipThinCover=True,
    consensus_interleaved_cnt=False, consensus_norm_prob_product=True,
    consensus_prob_product=True, consensus_norm_aff=True,
    vi_bg_use_inv_th=False, vi_bg_use_half_th=False, vi_bg_use_less_than_th=True,
    rank_norm_patch_score=True, rank_int_counter=False, patch_graph_norm_aff=True,
    flip_cons_arr_axes=False, pad_with_ps=False, overlapping_inst=True,
)
FIXED = dict(debug=False, isbiHack=False, skipLookup=True, skipConsensus=False,
             skipRanking=False, graphToInst=False, save_no_intermediates=True,
             termAfterThinCover=False, sample=1.0, result_folder=_BUILD_DIR,
             context=None, affinities="x.zarr", num_parallel_samples=1,
             num_parallel_blocks=1, return_intermediates=False)


def positive_planes(cons_ref, patchshape):
    """Compact the reference-layout consensus (NSZ,NSY,NSX,Z,Y,X) to the planes with a
    lexicographically positive offset; assert every other plane is exactly zero."""
    pz, py, px = patchshape
    used = np.zeros(cons_ref.shape[:3], dtype=bool)
    planes = []
    for dz in range(0, pz):
        for dy in range(-(py - 1), py):
            for dx in range(-(px - 1), px):
                if (dz, dy, dx) <= (0, 0, 0):
                    continue
                o = (dz + pz - 1, dy + py - 1, dx + px - 1)
                used[o] = True
                planes.append(cons_ref[o])
    assert not np.any(cons_ref[~used]), "reference wrote an unexpected plane"
    return np.stack(planes, axis=0)


def canonical_pairs(ref_pairs_arr, selected_sorted, n_self):
    """(i, j)-sorted pair rows + the self pairs in list order."""
    n = ref_pairs_arr.shape[0] - n_self
    idx = {tuple(int(v) for v in c): i for i, c in enumerate(selected_sorted)}
    rows = []
    for r in ref_pairs_arr[:n]:
        i, j = idx[tuple(int(v) for v in r[:3])], idx[tuple(int(v) for v in r[3:])]
        assert i < j
        rows.append((i, j))
    rows.sort()
    out = np.zeros_like(ref_pairs_arr)
    for k, (i, j) in enumerate(rows):
        out[k, :3] = selected_sorted[i]
        out[k, 3:] = selected_sorted[j]
    out[n:] = ref_pairs_arr[n:]
    return out


def run_reference(case, flags):
    import vote_instances as vi  # the reference module, imported in place
    import consensus_array as ca
    import ranked_patches as rp
    import foreground_cover as fc
    import aff_patch_graph as apg
    import graph_to_labeling as g2l

    patchshape = np.array(case["patchshape"])
    kw = dict(FLYLIGHT)
    kw.update(FIXED)
    kw.update(flags)
    kw["mutex"] = threading.Lock()
    pred = np.ascontiguousarray(case["pred"].astype(np.float32))
    fg = case["foreground"].copy()
    numinst = case["numinst"].copy()
    out = {}

    # ---- end-to-end through the reference's own orchestration (set pair order)
    res = vi.to_instance_seg(pred.copy(), fg.copy(), fg.copy(), numinst.copy(),
                             patchshape.copy(), **kw)
    out["e2e_instances_ref_order"] = np.asarray(res[0]) if res[0] is not None \
        else np.zeros(fg.shape, np.uint16)

    # ---- stage by stage (same calls as vote_instances.py:193-452)
    rad = np.array([p // 2 for p in patchshape])
    radslice = tuple(slice(rad[i], fg.shape[i] - rad[i]) for i in range(3))
    overlap_mask = 1 * (numinst > 1)
    mask_to_cover = fg.copy()
    mask_to_cover[overlap_mask > 0] = 0
    out["early_out"] = np.array(0)
    if np.count_nonzero(mask_to_cover[radslice]) == 0:
        out["early_out"] = np.array(1)
        return out
    neighshape = patchshape.copy()
    if neighshape[0] > 1:
        neighshape *= 2
    else:
        neighshape[1:] *= 2
    all_patches = np.transpose(np.where(fg))
    all_patches = [p for p in all_patches
                   if np.all(p >= rad) and np.all(p < fg.shape - rad)]
    if len(all_patches) == 0:
        out["early_out"] = np.array(2)
        return out

    tmp = alloc_zero_array(pred.shape, np.float32)
    tmp[:] = pred
    pred_m = tmp
    cons = ca.create_consensus_array_cuda(pred_m, overlap_mask, patchshape,
                                          neighshape, **kw)
    if kw.get("flip_cons_arr_axes"):
        cons_std = np.ascontiguousarray(np.moveaxis(cons, (0, 1, 2), (3, 4, 5)))
    else:
        cons_std = cons
    out["cons_pos"] = positive_planes(cons_std, [int(p) for p in patchshape])

    scores = rp.rank_patches_cuda(pred_m, cons, patchshape, neighshape,
                                  overlap_mask, **kw)
    out["scores"] = np.array(scores)
    ranked = rp.rank_patches_by_score(all_patches, scores)
    out["ranked_coords"] = np.array([r[0] for r in ranked], dtype=np.int32)
    out["ranked_scores"] = np.array([r[1] for r in ranked], dtype=np.float32)

    sel, nsel = fc.computeForegroundCover(
        overlap_mask, mask_to_cover, patchshape, ranked, radslice, pred_m, rad,
        None, np.array(scores), silent=True, **kw)
    out["cover_coords"] = np.array([s[0] for s in sel], dtype=np.int32).reshape(-1, 3)
    if not kw["skipThinCover"] and nsel > 0:
        sel, nsel = fc.thinOutForegroundCover(mask_to_cover, sel, radslice, pred_m,
                                              rad, patchshape, **kw)
        out["thin_coords"] = np.array([s[0] for s in sel],
                                      dtype=np.int32).reshape(-1, 3)

    sel = list(sel)
    pairs_ref = apg.computeAndStorePatchPairs(sel, patchshape, **kw)  # sorts sel by x
    out["selected_sorted"] = np.array([s[0] for s in sel], dtype=np.int32).reshape(-1, 3)
    if pairs_ref is None:
        out["early_out"] = np.array(3)
        return out
    n_self = len(sel) if kw["includeSinglePatchCCS"] else 0
    out["pairs_ref_order"] = np.array(pairs_ref)
    pairs = canonical_pairs(np.array(pairs_ref), out["selected_sorted"], n_self)
    out["pairs"] = pairs

    tmpp = alloc_zero_array(pairs.shape, np.uint32)
    tmpp[:] = pairs
    kw_ri = dict(kw)
    kw_ri["return_intermediates"] = True
    aff = apg.computePatchGraph_cuda(pred_m, cons, tmpp, patchshape, neighshape,
                                     **kw_ri)
    out["aff"] = np.array(aff)

    graph = apg.setAffgraph(aff, tmpp)
    instances = (0 * fg).astype(np.uint16)
    inst, fgo = g2l.affGraphToInstances(graph, pred_m, patchshape, rad, None, None,
                                        instances, fg, **kw)
    out["instances"] = np.asarray(inst)
    out["foreground_out"] = np.asarray(fgo)
    return out


# ----------------------------------------------------------------------------------
# 4. the cases
# ----------------------------------------------------------------------------------
CASES = {
    # name: (shape, patchshape, synth kwargs, flag overrides)
    "c3d_p3_blobs": ((12, 12, 12), (3, 3, 3), dict(kind="two_blobs", seed=1), {}),
    "c3d_p3_cells": ((14, 14, 14), (3, 3, 3), dict(kind="cells", seed=2, cell=[5, 5, 5]),
                     {}),
    "c3d_p3_overlap": ((14, 14, 14), (3, 3, 3),
                       dict(kind="cells", seed=3, cell=[6, 6, 6], overlap_frac=0.03), {}),
    "c3d_p3_thin_mws": ((12, 12, 12), (3, 3, 3), dict(kind="cells", seed=4, cell=[5, 5, 5]),
                        dict(skipThinCover=False, mws=True)),
    "c3d_p3_nosparse": ((12, 12, 12), (3, 3, 3), dict(kind="cells", seed=9, cell=[5, 5, 5]),
                        dict(select_patches_for_sparse_data=False,
                             includeSinglePatchCCS=False)),
    "c2d_p5_blobs": ((1, 24, 24), (1, 5, 5), dict(kind="two_blobs", seed=5), {}),
    "c2d_p5_th09_inv": ((1, 28, 28), (1, 5, 5),
                        dict(kind="cells", seed=6, cell=[1, 9, 9], noise=0.3),
                        dict(patch_threshold=0.9, vi_bg_use_inv_th=True,
                             vi_bg_use_less_than_th=False, overlapping_inst=False,
                             consensus_interleaved_cnt=True, flip_cons_arr_axes=True,
                             skipThinCover=False)),
    "c2d_p5_th09_half": ((1, 28, 28), (1, 5, 5),
                         dict(kind="cells", seed=7, cell=[1, 9, 9], noise=0.3),
                         dict(patch_threshold=0.9, vi_bg_use_half_th=True,
                              vi_bg_use_less_than_th=False)),
    "c2d_p5_th07_lt": ((1, 28, 28), (1, 5, 5),
                       dict(kind="cells", seed=8, cell=[1, 9, 9], noise=0.25),
                       dict(patch_threshold=0.7, rank_int_counter=True)),
    "c2d_p5_rawcount": ((1, 24, 24), (1, 5, 5), dict(kind="cells", seed=10, cell=[1, 8, 8]),
                        dict(consensus_norm_prob_product=False,
                             consensus_prob_product=False, consensus_norm_aff=False,
                             consensus_interleaved_cnt=False,
                             rank_norm_patch_score=False, patch_graph_norm_aff=False)),
    "c2d_p5_probprod": ((1, 24, 24), (1, 5, 5), dict(kind="cells", seed=11, cell=[1, 8, 8]),
                        dict(consensus_norm_prob_product=False,
                             consensus_prob_product=True)),
    "c3d_p5_cells": ((16, 16, 16), (5, 5, 5), dict(kind="cells", seed=12, cell=[7, 7, 7]),
                     {}),
    "c3d_p7_cells": ((16, 18, 20), (7, 7, 7), dict(kind="cells", seed=13, cell=[9, 9, 9]),
                     {}),
    # the patch shapes of BASELINE configs [2]/[3] (9^3) and [0] (wormbodies 2-d 25x25)
    "c3d_p9_cells": ((14, 15, 16), (9, 9, 9), dict(kind="cells", seed=14, cell=[7, 8, 8]), {}),
    "c2d_p25_cells": ((1, 40, 44), (1, 25, 25), dict(kind="cells", seed=15, cell=[1, 14, 14]),
                      {}),
    # the shipped flylight flags (default.toml:134,141: thinning + mutex watershed) at p = 5 / 7
    "c3d_p5_thin_mws": ((16, 16, 16), (5, 5, 5), dict(kind="cells", seed=16, cell=[7, 7, 7]),
                        dict(skipThinCover=False, mws=True)),
    "c3d_p7_thin_mws": ((16, 18, 20), (7, 7, 7), dict(kind="cells", seed=17, cell=[9, 9, 9]),
                        dict(skipThinCover=False, mws=True)),
    # graph_to_labeling.py:57-115: a channel per instance / instances packed into channels
    # without overlap (components above 2000 voxels are placed, the others go to channel 0)
    "c3d_p3_per_channel": ((12, 13, 14), (3, 3, 3), dict(kind="cells", seed=18, cell=[5, 5, 5]),
                           dict(one_instance_per_channel=True)),
    "c3d_p3_packed_channels": ((22, 30, 32), (3, 3, 3),
                               dict(kind="cells", seed=19, cell=[11, 15, 16], overlap_frac=0.02),
                               dict(no_overlap_per_channel=True, skipThinCover=False)),
    # the two optional branches of the greedy cover (foreground_cover.py:53-85, 141-143, 162-168)
    "c2d_p5_mark": ((1, 28, 30), (1, 5, 5), dict(kind="cells", seed=24, cell=[1, 9, 9]),
                    dict(mark_close_neighboorhood=True)),
    "c3d_p3_mark_nosparse": ((8, 13, 14), (3, 3, 3), dict(kind="cells", seed=26, cell=[4, 6, 6]),
                             dict(mark_close_neighboorhood=True, select_patches_for_sparse_data=False)),
    "c3d_p3_near_overlap": ((12, 13, 14), (3, 3, 3),
                            dict(kind="cells", seed=25, cell=[6, 6, 6], overlap_frac=0.04),
                            dict(select_patches_overlap_neighborhood=True)),
    "c3d_empty": ((10, 10, 10), (3, 3, 3), dict(kind="empty", seed=0), {}),
    "c3d_single_patch": ((3, 3, 3), (3, 3, 3), dict(kind="cells", seed=1, cell=[9, 9, 9]),
                         {}),
}
# cases whose consensus array is too big to commit: keep a SHA-256 of the float bits
HASH_ONLY_CONS = {"c3d_p5_cells", "c3d_p7_cells", "c3d_p9_cells", "c2d_p25_cells",
                  "c3d_p5_thin_mws", "c3d_p7_thin_mws", "c3d_p3_packed_channels",
                  "c3d_p3_near_overlap", "c3d_p3_mark_nosparse"}
# cases that are ALSO run through the reference's NumPy path (cuda=False; int16 +-1 votes,
# SURVEY 8c "recipe A").  Different arithmetic from the kernels: only the final instance map
# is stored, to document that both semantics agree on well separated instances.
CPU_PATH_CASES = {"c2d_p5_blobs", "c3d_p3_blobs"}


def run_reference_cpu_path(case, flags):
    import vote_instances as vi
    kw = dict(FLYLIGHT)
    kw.update(FIXED)
    kw.update(flags)
    kw.update(cuda=False, skipLookup=False, mutex=threading.Lock())
    patchshape = np.array(case["patchshape"])
    fg = case["foreground"].copy()
    inst, _ = vi.to_instance_seg(case["pred"].astype(np.float32), fg, fg.copy(),
                                 case["numinst"].copy(), patchshape, **kw)
    return np.asarray(inst)


def main(argv):
    if not os.path.isdir(REF_VI):
        sys.exit("reference tree not found at %s (development container only)" % REF)
    install_stubs()
    install_fake_cuda_code()
    sys.path.insert(0, REF_VI)
    import logging
    logging.basicConfig(level=logging.WARNING)
    names = argv or list(CASES)
    for name in names:
        shape, ps, skw, flags = CASES[name]
        case = synth.make_case(shape, ps, **skw)
        if name == "c3d_single_patch":
            case["foreground"][:] = True
            case["numinst"][:] = 1
        case["patchshape"] = list(ps)
        out = run_reference(case, flags)
        if name in HASH_ONLY_CONS and "cons_pos" in out:
            c = np.ascontiguousarray(out.pop("cons_pos"))
            out["cons_pos_sha256"] = np.array(hashlib.sha256(c.tobytes()).hexdigest())
            out["cons_pos_sum"] = np.array(c.astype(np.float64).sum())
        if name in CPU_PATH_CASES:
            out["instances_cpu_path"] = run_reference_cpu_path(case, flags)
        kw = dict(FLYLIGHT)
        kw.update(flags)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            pred_f16=case["pred"].astype(np.float16),
            foreground=case["foreground"], numinst=case["numinst"],
            patchshape=np.array(ps), flags=np.array(json.dumps(kw)), **out)
        ninst = len(np.unique(out.get("instances", np.zeros(1)))) - 1
        print("%-20s early_out=%s pairs=%s instances=%s" % (
            name, int(out["early_out"]),
            out.get("pairs", np.zeros((0, 6))).shape[0], ninst))


if __name__ == "__main__":
    main(sys.argv[1:])
