#!/usr/bin/env python3
"""Random zarr stores (chunk shapes incl. split channels, codecs, probabilities or logits) streamed tile by
tile through tiling.ZarrProvider -- decode on a worker thread, pinned buffers, asynchronous upload -- under
random tile plans, against the assembly of the resident tensor as ONE tile (development aid; the fixed cases
are tests/test_cli_gpu.py::test_streamed_prediction_equals_loaded_prediction and tests/test_minizarr.py).

  python tools/fuzz_streaming.py [--trials 30] [--seed 1]
"""
import argparse
import os
import sys
import tempfile
import time
import traceback

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch
    from patchperpix_amd import minizarr, synth, tiling
    from patchperpix_amd import flags as F
    rng = np.random.default_rng(args.seed)
    bad = 0
    t0 = time.time()
    for trial in range(args.trials):
        p = int(rng.choice([3, 5, 7], p=[0.3, 0.45, 0.25]))
        ps = (p, p, p)
        C = p ** 3
        shape = tuple(int(rng.integers(2 * p + 2, 2 * p + 34)) for _ in range(3))
        flagset = str(rng.choice(["shipped", "cc"]))
        logits = bool(rng.integers(0, 3) == 0)
        chunks = (int(rng.choice([C, C, max(1, C // 2 + 1), max(1, C // 3)])),) + tuple(int(rng.integers(3, 17)) for _ in range(3))
        comp = str(rng.choice(["default", "gzip", "none"]))
        compressor = {"default": "default", "gzip": {"id": "gzip", "level": 1}, "none": None}[comp]
        n_slabs = int(rng.integers(1, 5))
        yx = (int(rng.integers(1, 3)), int(rng.integers(1, 3)))
        async_h2d = str(rng.integers(0, 2))
        seed = int(rng.integers(1, 100000))
        cell = int(rng.integers(max(4, p), 2 * p + 4))
        desc = "trial %d shape %s p %d %s logits %d chunks %s %s slabs %d yx %s async %s seed %d" % (
            trial, shape, p, flagset, logits, chunks, comp, n_slabs, yx, async_h2d, seed)
        try:
            c = synth.make_case(shape, ps, seed=seed, cell=[cell] * 3, overlap_frac=0.02)
            prob = np.clip(c["pred"], 1e-3, 1 - 1e-3)
            data = (np.log(prob / (1 - prob)) if logits else c["pred"]).astype(np.float16)
            kw = dict(F.FLAG_SETS[flagset], _instances_dtype=np.uint32)
            fg, ni = c["foreground"], c["numinst"]
            with tempfile.TemporaryDirectory() as tmp:
                g = minizarr.open(os.path.join(tmp, "s.zarr"), "w")
                a = g.create("volumes/pred_affs", shape=data.shape, chunks=chunks, dtype=np.float16, compressor=compressor)
                a[...] = data
                arr = minizarr.open(os.path.join(tmp, "s.zarr"), "r")["volumes/pred_affs"]
                # the resident tensor: what the provider hands out for the whole volume, as one tile
                whole = tiling.ZarrProvider(arr, expit=logits).pred_box((0, shape[0], 0, shape[1], 0, shape[2]))
                want, _ = tiling.assemble(whole, 0, shape, fg.copy(), fg.copy(), ni.copy(), list(ps), tiling.plan_slabs(shape[0], 1), **kw)
                del whole
                os.environ["PPP_ASYNC_H2D"] = async_h2d
                prov = tiling.ZarrProvider(arr, expit=logits)
                got, _ = tiling.assemble(prov, 0, shape, fg.copy(), fg.copy(), ni.copy(), list(ps), tiling.plan_slabs(shape[0], n_slabs),
                                         _yx_tiles=yx, **kw)
                os.environ.pop("PPP_ASYNC_H2D", None)
            ok = np.array_equal(want, got)
            print(desc, "instances %d prefetched %d uploaded ahead %d:" % (int(want.max()), prov.boxes_prefetched, prov.boxes_uploaded_ahead),
                  "ok" if ok else "DIFFER (%d voxels)" % int(np.count_nonzero(want != got)), flush=True)
            bad += not ok
        except Exception as e:       # noqa: BLE001
            print(desc, "EXCEPTION %r" % (e,), flush=True)
            traceback.print_exc()
            bad += 1
            os.environ.pop("PPP_ASYNC_H2D", None)
    print("%d trials, %d failures, %.0f s" % (args.trials, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
