#!/usr/bin/env python3
"""Golden vectors of the reference's NumPy path (``cuda=False``; SURVEY 8(a) row a11, SURVEY 8(c)
"recipe A"), stage by stage.  DEVELOPMENT CONTAINER ONLY: imports the reference's Python modules in
place from /root/reference (with the stub modules of gen_golden.py) and calls ITS functions:

  utilVoteInstances.fillLookup / computeFGBGsets     (:19-92)
  consensus_array.create_consensus_array              (:18-68)   int16 +-1 votes
  ranked_patches.rank_patches                         (:76-105)  integer sign counts
  foreground_cover.computeForegroundCover / thinOutForegroundCover
  aff_patch_graph.computeAndStorePatchPairs           (sorts the selection by x)
  aff_patch_graph.computePatchGraph (NumPy branch)    (:190-282) weights over all pixel pairs
  graph_to_labeling.affGraphToInstances

with ``removeIntersection=False, sample=1.0`` (the other settings draw from Python's unseeded
``random``) and checks that the stage-wise run equals ``vote_instances.to_instance_seg(cuda=False)``
end to end.  Writes tests/golden/np_<case>.npz (DATA: seeded inputs + the reference's outputs).

  python tests/golden/gen_golden_numpy_path.py [case ...]
"""
import json
import os
import sys
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (stubs, fake cuda_code, synthetic cases, flag sets)
from patchperpix_amd import synth  # noqa: E402

CASES = {
    # name: (shape, patchshape, synth kwargs, flag overrides)
    "c2d_p5_blobs": ((1, 24, 24), (1, 5, 5), dict(kind="two_blobs", seed=5), {}),
    "c3d_p3_blobs": ((10, 10, 10), (3, 3, 3), dict(kind="two_blobs", seed=1), {}),
    # touching cells, thinning + mutex watershed, overlap voxels
    "c3d_p3_cells_thin_mws": ((10, 11, 12), (3, 3, 3),
                              dict(kind="cells", seed=21, cell=[5, 5, 5], overlap_frac=0.03),
                              dict(skipThinCover=False, mws=True)),
    # the reference's own default thresholds (vote_instances.py:82-83): 0.9 / background < 0.1
    "c2d_p5_th09": ((1, 22, 26), (1, 5, 5), dict(kind="cells", seed=22, cell=[1, 9, 9], noise=0.3),
                    dict(patch_threshold=0.9, skipThinCover=False)),
    # a threshold below 0.5: a pixel can be in the foreground AND the background set
    "c2d_p3_th04": ((1, 16, 18), (1, 3, 3), dict(kind="cells", seed=23, cell=[1, 6, 6], noise=0.45),
                    dict(patch_threshold=0.4, includeSinglePatchCCS=False)),
    # BASELINE config [0]'s patch shape, (1, 25, 25), with the reference's own default thresholds
    # (vote_instances.py:82-83, 488) on a crop its Python path finishes: 48 x 52 pixels, 672 centres
    "c2d_p25_crop": ((1, 48, 52), (1, 25, 25), dict(kind="cells", seed=24, cell=[1, 13, 14], noise=0.3),
                     dict(patch_threshold=0.9, skipThinCover=False)),
}


def run(case, flags):
    import vote_instances as vi
    import utilVoteInstances as util
    import consensus_array as ca
    import ranked_patches as rp
    import foreground_cover as fc
    import aff_patch_graph as apg
    import graph_to_labeling as g2l

    patchshape = np.array(case["patchshape"])
    kw = dict(gg.FLYLIGHT)
    kw.update(gg.FIXED)
    kw.update(flags)
    kw.update(cuda=False, skipLookup=False, removeIntersection=False, sample=1.0,
              mutex=threading.Lock())
    pred = np.ascontiguousarray(case["pred"].astype(np.float32))
    fg = case["foreground"].copy()
    numinst = case["numinst"].copy()
    out = {}
    inst_e2e, _ = vi.to_instance_seg(pred.copy(), fg.copy(), fg.copy(), numinst.copy(), patchshape.copy(), **kw)
    out["instances"] = np.asarray(inst_e2e)

    rad = np.array([p // 2 for p in patchshape])
    radslice = tuple(slice(rad[i], fg.shape[i] - rad[i]) for i in range(3))
    overlap_mask = 1 * (numinst > 1)
    mask_to_cover = fg.copy()
    mask_to_cover[overlap_mask > 0] = 0
    neighshape = patchshape.copy()
    if neighshape[0] > 1:
        neighshape *= 2
    else:
        neighshape[1:] *= 2
    every = np.transpose(np.where(fg))
    lookup = util.fillLookup(fg, patchshape, neighshape, every)
    all_patches = [p for p in every if np.all(p >= rad) and np.all(p < fg.shape - rad)]
    fgs, bgs = util.computeFGBGsets(fg, all_patches, pred, patchshape, rad, **kw)
    cons, obff, obfb = ca.create_consensus_array(fgs, bgs, fg.shape, patchshape, neighshape, lookup)
    assert cons.dtype == np.int16
    nz = np.nonzero(cons)
    out["cons_index"] = np.stack(nz, axis=1).astype(np.int32)        # (L, z, y, x) of the non-zero votes
    out["cons_value"] = cons[nz].astype(np.int16)
    out["cons_shape"] = np.array(cons.shape)
    ranked = rp.rank_patches(obff, obfb, all_patches, cons)
    out["ranked_coords"] = np.array([r[0] for r in ranked], dtype=np.int32).reshape(-1, 3)
    out["ranked_scores"] = np.array([int(r[1]) for r in ranked], dtype=np.int64)
    sel, nsel = fc.computeForegroundCover(overlap_mask, mask_to_cover, patchshape, ranked, radslice, pred,
                                          rad, None, None, silent=True, **kw)
    out["cover_coords"] = np.array([s[0] for s in sel], dtype=np.int32).reshape(-1, 3)
    if not kw["skipThinCover"] and nsel > 0:
        sel, nsel = fc.thinOutForegroundCover(mask_to_cover, sel, radslice, pred, rad, patchshape, **kw)
        out["thin_coords"] = np.array([s[0] for s in sel], dtype=np.int32).reshape(-1, 3)
    sel = list(sel)
    pairs_ref = apg.computeAndStorePatchPairs(sel, patchshape, **kw)          # sorts sel by x
    out["selected_sorted"] = np.array([s[0] for s in sel], dtype=np.int32).reshape(-1, 3)
    out["has_pairs"] = np.array(0 if pairs_ref is None else 1)
    if pairs_ref is None:
        return out
    graph = apg.computePatchGraph(sel, nsel, pairs_ref, pred, mask_to_cover, patchshape, neighshape, rad,
                                  overlap_mask, lookup, cons, **kw)
    edges = [(u, v, d["aff"]) for u, v, d in graph.edges(data=True)]
    out["edge_rows"] = np.array([list(u) + list(v) for u, v, _ in edges], dtype=np.int32).reshape(-1, 6)
    out["edge_weight"] = np.array([int(w) for _, _, w in edges], dtype=np.int64)
    out["node_order"] = np.array(list(graph.nodes()), dtype=np.int32).reshape(-1, 3)
    instances = (0 * fg).astype(np.uint16)
    inst, _ = g2l.affGraphToInstances(graph, pred, patchshape, rad, None, None, instances, fg, **kw)
    assert np.array_equal(np.asarray(inst), out["instances"]), "stage-wise run differs from to_instance_seg"
    return out, kw


def main(argv):
    gg.install_stubs()
    gg.install_fake_cuda_code()
    sys.path.insert(0, gg.REF_VI)
    import logging
    logging.basicConfig(level=logging.WARNING)
    for name in argv or list(CASES):
        shape, ps, skw, flags = CASES[name]
        case = synth.make_case(shape, ps, **skw)
        case["patchshape"] = list(ps)
        res = run(case, flags)
        out, kw = res if isinstance(res, tuple) else (res, None)
        kwj = dict(gg.FLYLIGHT)
        kwj.update(flags)
        kwj.update(cuda=False, removeIntersection=False, sample=1.0)
        np.savez_compressed(os.path.join(HERE, "np_" + name + ".npz"),
                            pred_f16=case["pred"].astype(np.float16), foreground=case["foreground"],
                            numinst=case["numinst"], patchshape=np.array(ps),
                            flags=np.array(json.dumps({k: v for k, v in kwj.items()
                                                       if isinstance(v, (bool, int, float, str))})), **out)
        print("%-24s votes=%d ranked=%d selected=%d edges=%d instances=%d" % (
            name, len(out["cons_value"]), len(out["ranked_coords"]), len(out["selected_sorted"]),
            len(out.get("edge_weight", [])), len(np.unique(out["instances"])) - 1))


if __name__ == "__main__":
    main(sys.argv[1:])
