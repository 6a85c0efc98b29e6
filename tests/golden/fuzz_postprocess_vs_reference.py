#!/usr/bin/env python3
"""remove_small_components / relabel of patchperpix_amd.postprocess against the reference's own functions
(PatchPerPix/util/postprocess.py:24-52, imported in place; import-time stubs for colorcet, skimage, zarr, h5py,
nrrd -- none of them is called) on random label volumes.  Development container only.

  python tests/golden/fuzz_postprocess_vs_reference.py [--trials 300]
"""
import argparse
import contextlib
import importlib.util
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = "/root/reference/PatchPerPix/util/postprocess.py"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    for name in ("colorcet", "skimage", "skimage.morphology", "zarr", "h5py", "nrrd"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["skimage.morphology"].skeletonize_3d = None
    spec = importlib.util.spec_from_file_location("ppp_ref_postprocess", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    from patchperpix_amd import postprocess as pp
    rng = np.random.default_rng(args.seed)
    bad = 0
    for trial in range(args.trials):
        nd = int(rng.integers(2, 4))
        shape = tuple(int(rng.integers(1, 14)) for _ in range(nd))
        nlab = int(rng.integers(1, 40))
        dtype = rng.choice([np.uint16, np.uint32, np.int64])
        a = (rng.integers(0, nlab + 1, size=shape) * (rng.random(shape) < rng.uniform(0.1, 1.0))).astype(dtype)
        size = int(rng.integers(0, 30))
        start = None if rng.integers(0, 2) else int(rng.integers(1, 50))
        with contextlib.redirect_stdout(io.StringIO()):       # (the reference prints its labels)
            want = ref.remove_small_components(a.copy(), size)
            want_rl = ref.relabel(want.copy(), start)
        got = pp.remove_small_components(a.copy(), size)
        got_rl = pp.relabel(got.copy(), start)
        ok = np.array_equal(got, want) and np.array_equal(got_rl, want_rl)
        if not ok:
            bad += 1
            print("trial", trial, shape, nlab, dtype.__name__, size, start, "DIFFER")
    print("%d trials, %d failures" % (args.trials, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
