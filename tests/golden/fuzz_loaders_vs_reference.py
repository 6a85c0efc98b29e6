#!/usr/bin/env python3
"""The loaders' small print -- getFgThreshold / maybeLoadNuminst / loadFg / returnFg / getResKey / the kernel
build options (utilVoteInstances.py:254-330, 340-400) -- of this package's mirror against the reference's own
functions (imported in place with gen_golden.py's stubs) on random containers and flags.  Development container only.

  python tests/golden/fuzz_loaders_vs_reference.py [--trials 400]
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402


def same(a, b):
    if a is None or b is None:
        return a is None and b is None
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    gg.install_stubs()
    gg.install_fake_cuda_code()
    sys.path.insert(0, gg.REF_VI)
    import logging
    logging.basicConfig(level=logging.ERROR)
    import utilVoteInstances as ref
    from patchperpix_amd.vote_instances import utilVoteInstances as mine
    rng = np.random.default_rng(args.seed)
    bad = raised = 0
    for trial in range(args.trials):
        two_d = bool(rng.integers(0, 3) == 0)
        p = int(rng.choice([3, 5, 7]))
        ps = np.array([1, p, p] if two_d else [p, p, p])
        shape = (1 if two_d else int(rng.integers(2, 7)), int(rng.integers(2, 9)), int(rng.integers(2, 9)))
        C = int(np.prod(ps))
        nch = int(rng.choice([1, 2, 3, 4]))
        prob = rng.random((nch,) + shape).astype(rng.choice([np.float16, np.float32]))
        f = {"volumes/pred_affs": rng.random((C,) + shape).astype(np.float16),
             "volumes/pred_numinst": prob,
             "volumes/pred_fgbg": rng.random((1,) + shape).astype(np.float32)}
        kw = dict(gg.FLYLIGHT)
        kw.update(gg.FIXED)
        kw.pop("fg_key", None), kw.pop("numinst_key", None), kw.pop("numinst_threshs", None), kw.pop("fg_thresh_vi", None)
        kw.update(aff_key="volumes/pred_affs", patch_threshold=float(rng.choice([0.5, 0.9])), patchshape=ps,
                  mws=bool(rng.integers(0, 2)), select_patches_for_sparse_data=bool(rng.integers(0, 2)))
        mode = str(rng.choice(["fg", "numinst", "affs"]))
        if mode == "fg":
            kw["fg_key"] = "volumes/pred_fgbg"
        if mode == "numinst" or rng.integers(0, 2):
            kw["numinst_key"] = "volumes/pred_numinst"
        if rng.integers(0, 2):
            kw["fg_thresh_vi"] = float(rng.choice([-1, 0.3, 0.7]))
        if "numinst_key" in kw and nch >= 2 and rng.integers(0, 2):
            kw["numinst_threshs"] = [float(v) for v in rng.random(int(rng.integers(1, nch)))]
        kw["skipThinCover"] = bool(rng.integers(0, 2))
        status = []
        try:
            if ref.getFgThreshold(**kw) != mine.getFgThreshold(**kw):
                status.append("getFgThreshold")
            if not same(ref.maybeLoadNuminst(f, **kw), mine.maybeLoadNuminst(f, **kw)):
                status.append("maybeLoadNuminst")
            a, ka = ref.loadFg(f, **kw)
            b, kb = mine.loadFg(f, **kw)
            if ka != kb or not same(a, b):
                status.append("loadFg")
            ni = ref.maybeLoadNuminst(f, **kw)
            ra = ref.returnFg(f["volumes/pred_affs"], ni, f["volumes/pred_fgbg"], **kw)
            rb = mine.returnFg(f["volumes/pred_affs"], ni, f["volumes/pred_fgbg"], **kw)
            if not same(ra, rb):
                status.append("returnFg")
            if ref.getResKey(**kw) != mine.getResKey(**kw):
                status.append("getResKey")
        except Exception as e:      # noqa: BLE001
            # both sides must fail the same way
            raised += 1
            try:
                mine.loadFg(f, **kw)
                mine.maybeLoadNuminst(f, **kw)
                status.append("REFERENCE RAISED %r, mirror did not" % (e,))
            except Exception:       # noqa: BLE001
                pass
        if status:
            bad += 1
            print("trial", trial, mode, {k: kw.get(k) for k in ("fg_key", "numinst_key", "numinst_threshs", "fg_thresh_vi", "patch_threshold")}, prob.shape, "DIFFER", status)
    print("%d trials, %d failures (%d where the reference raised and the mirror raised too)" % (args.trials, bad, raised))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
