#!/usr/bin/env python3
"""loadAffinities (utilVoteInstances.py:136-251): what the reference makes of a prediction container -- channels
first or last, 2-d (three axes) or 3-d arrays, crops, the ISBI hack, logits -> logistic function, the numinst /
foreground companions, the "already computed" early return, the 'images/' layout -- against the package's reader.
The reference's function is imported in place; its `zarr.open` is pointed at in-memory containers holding exactly the
arrays this package reads from a real store written by minizarr.  Development container only.

  python tests/golden/fuzz_load_affinities_vs_reference.py [--trials 300]
"""
import argparse
import os
import sys
import tempfile
import traceback

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
from patchperpix_amd import minizarr  # noqa: E402


class Container:
    """zarr-group look-alike over {path: array}: keys() = the top-level names, item access by path"""

    def __init__(self, arrays):
        self.arrays = arrays

    def keys(self):
        return sorted(set(k.split("/")[0] for k in self.arrays))

    def __contains__(self, k):
        return k in self.arrays or any(a.startswith(k + "/") for a in self.arrays)

    def __getitem__(self, k):
        return self.arrays[k]


def same(a, b):
    if a is None or b is None:
        return a is None and b is None
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a, b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    gg.install_stubs()
    gg.install_fake_cuda_code()
    sys.path.insert(0, gg.REF_VI)
    import logging
    logging.basicConfig(level=logging.CRITICAL)
    import utilVoteInstances as ref
    from patchperpix_amd.vote_instances import utilVoteInstances as mine
    rng = np.random.default_rng(args.seed)
    stores = {}
    ref.zarr.open = lambda path, mode="r": stores[path]
    bad = both_raised = 0
    for trial in range(args.trials):
        two_d = bool(rng.integers(0, 2))
        p = int(rng.choice([3, 5]))
        ps = np.array([1, p, p] if two_d else [p, p, p])
        C = int(np.prod(ps))
        sp = (int(rng.integers(3, 9)), int(rng.integers(3, 9))) if two_d else tuple(int(rng.integers(2, 7)) for _ in range(3))
        three_axes = two_d and bool(rng.integers(0, 2))
        full = sp if three_axes else ((1,) + sp if two_d else sp)
        last = bool(rng.integers(0, 4) == 0)          # channels last (the ISBI layout)
        vals = rng.random((C,) + full).astype(np.float32)
        if rng.integers(0, 3) == 0:                  # logits
            vals = (vals * 8 - 4).astype(np.float32)
        affs = np.moveaxis(vals, 0, -1).copy() if last else vals
        dtype = rng.choice([np.float16, np.float32])
        arrays = {"volumes/pred_affs": affs.astype(dtype)}
        kw = dict(gg.FLYLIGHT)
        kw.update(gg.FIXED)
        for k in ("fg_key", "numinst_key", "numinst_threshs", "fg_thresh_vi"):
            kw.pop(k, None)
        kw.update(aff_key="volumes/pred_affs" if rng.integers(0, 2) else None, patch_threshold=float(rng.choice([0.5, 0.9])))
        if rng.integers(0, 2):
            nch = int(rng.choice([2, 3]))
            arrays["volumes/pred_numinst"] = rng.random((nch,) + full).astype(np.float16)
            kw["numinst_key"] = "volumes/pred_numinst"
            if rng.integers(0, 2):
                kw["numinst_threshs"] = [float(v) for v in rng.random(nch - 1)]
        if rng.integers(0, 3) == 0:
            arrays["volumes/pred_fgbg"] = rng.random((1,) + full).astype(np.float32)
            kw["fg_key"] = "volumes/pred_fgbg"
        if rng.integers(0, 4) == 0:
            for ax, n in zip("zyx"[-len(full):] if not three_axes else "yx", full):
                if rng.integers(0, 2) and n > 3:
                    kw["crop_%s_s" % ax] = int(rng.integers(0, 2))
                    kw["crop_%s_e" % ax] = int(n - rng.integers(0, 2))
        if rng.integers(0, 10) == 0 and not three_axes:
            kw["isbiHack"] = True
        res_ext = ""
        if rng.integers(0, 12) == 0:
            arrays["vote_instances"] = np.zeros(full, dtype=np.uint16)
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "s.zarr")
            g = minizarr.open(path, "w")
            for k, v in arrays.items():
                g.create_dataset(k, data=v, chunks=tuple(min(4, s) if i else s for i, s in enumerate(v.shape)))
            stores[path] = Container(arrays)
            want = got = err = gerr = None
            try:
                want = ref.loadAffinities(path, res_ext, patchshape=ps.copy(), **dict(kw))
            except BaseException as e:      # noqa: BLE001  (the reference exits on some inputs)
                err = e
            try:
                got = mine.loadAffinities(path, res_ext, patchshape=ps.copy(), **dict(kw))
            except BaseException as e:      # noqa: BLE001
                gerr = e
            stores.pop(path)
        status = []
        if err is not None or gerr is not None:
            if err is not None and gerr is not None:
                both_raised += 1
                continue
            status.append("reference %s, package %s" % ("raised %r" % (err,) if err is not None else "returned", "raised %r" % (gerr,) if gerr is not None else "returned"))
        elif (want is None) != (got is None):
            status.append("early return")
        elif want is not None:
            for name, a, b in zip(("affinities", "numinst", "foreground"), want, got):
                if not same(a, b):
                    status.append("%s %s vs %s" % (name, None if a is None else np.asarray(a).shape, None if b is None else np.asarray(b).shape))
        if status:
            bad += 1
            if bad <= 12:
                print("trial", trial, "2d" if two_d else "3d", "three axes" if three_axes else "", "last" if last else "first", affs.shape,
                      {k: kw.get(k) for k in ("aff_key", "numinst_key", "numinst_threshs", "fg_key", "isbiHack", "crop_z_s", "crop_z_e", "crop_y_s", "crop_y_e", "crop_x_s", "crop_x_e") if kw.get(k) is not None},
                      "DIFFER", status, flush=True)
    print("%d trials, %d failures (%d where both raise)" % (args.trials, bad, both_raised))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
