#!/usr/bin/env python3
"""Random volumes / block sizes / flags through the BLOCKWISE driver with the reference's per-block semantics
(patchperpix_amd.blockwise): the real kernels against the same driver served by the CPU oracle (development
aid; under tests/ because it calls the oracle; the fixed cases -- goldens of the reference's own driver -- are
tests/test_blockwise.py).  Compared: every block's and every inter-block group's stored pair rows and affinities
(bit patterns), the set of stored groups, the instance map.

  python tests/fuzz_blockwise_gpu.py [--trials 25] [--seed 1]
"""
import argparse
import json
import os
import sys
import tempfile
import time
import traceback

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def stored(store_path):
    from patchperpix_amd import minizarr
    out = {}
    res = minizarr.open(store_path, "r")
    if "volumes" not in res or "blocks" not in res["volumes"]:
        return out
    blocks = res["volumes/blocks"]

    def walk(g, prefix):
        for k in sorted(g.keys()):
            item = g[k]
            if hasattr(item, "keys"):
                walk(item, prefix + "/" + k)
            else:
                out[prefix + "/" + k] = np.asarray(item[...])
    walk(blocks, "")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=25)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import test_blockwise as tb
    from patchperpix_amd import blockwise, minizarr, synth
    from patchperpix_amd.vote_instances import vote_instances as vi
    base = json.loads(str(np.load(os.path.join(REPO, "tests", "golden", "bw_p5_thin_mws.npz"))["flags"]))
    rng = np.random.default_rng(args.seed)
    real_do_block, real_label = blockwise._do_block, blockwise.label_graph
    real_write = vi.write_result
    bad = 0
    t0 = time.time()
    for trial in range(args.trials):
        p = int(rng.choice([3, 5, 7], p=[0.4, 0.45, 0.15]))
        ps = [p, p, p]
        shape = tuple(int(rng.integers(2 * p + 2, 2 * p + 20)) for _ in range(3))
        chunk = [int(rng.integers(max(p + 1, 6), max(p + 2, s))) for s in shape]
        kw = dict(base, patchshape=ps, chunksize=chunk, mws=bool(rng.integers(0, 2)), skipThinCover=bool(rng.integers(0, 2)),
                  overlapping_inst=False, includeSinglePatchCCS=bool(rng.integers(0, 3) != 0))
        seed = int(rng.integers(1, 100000))
        cell = [int(rng.integers(p, 2 * p + 3))] * 3
        desc = "trial %d shape %s p %d chunk %s mws %d thin %d single %d seed %d cell %d" % (
            trial, shape, p, chunk, kw["mws"], not kw["skipThinCover"], kw["includeSinglePatchCCS"], seed, cell[0])
        try:
            c = synth.make_case(shape, tuple(ps), seed=seed, cell=cell, overlap_frac=0.0)
            with tempfile.TemporaryDirectory() as tmp:
                pred_file = os.path.join(tmp, "sample.zarr")
                g = minizarr.open(pred_file, "w")
                pred16 = c["pred"].astype(np.float16)
                g.create_dataset("volumes/pred_affs", data=pred16, chunks=(pred16.shape[0], 8, 8, 8))
                vi.write_result = lambda fn, ds: None
                blockwise._do_block, blockwise.label_graph = tb.oracle_do_block, tb.cpu_label_graph
                want = blockwise.main(pred_file, result_folder=os.path.join(tmp, "cpu"), **kw)
                blockwise._do_block, blockwise.label_graph = real_do_block, real_label
                got = blockwise.main(pred_file, result_folder=os.path.join(tmp, "gpu"), **kw)
                a, b = stored(os.path.join(tmp, "cpu", "sample.zarr")), stored(os.path.join(tmp, "gpu", "sample.zarr"))
            status = []
            if sorted(a) != sorted(b):
                status.append("GROUPS DIFFER (%d vs %d)" % (len(a), len(b)))
            else:
                for k in a:
                    va, vb = a[k], b[k]
                    same = va.shape == vb.shape and np.array_equal(va.view(np.uint32) if va.dtype == np.float32 else va,
                                                                   vb.view(np.uint32) if vb.dtype == np.float32 else vb)
                    if not same:
                        status.append("DATASET %s" % k)
            if (want is None) != (got is None) or (want is not None and not np.array_equal(want, got)):
                status.append("INSTANCES")
            n_inst = 0 if want is None else int(want.max())
            print(desc, "datasets %d instances %d:" % (len(a), n_inst), "ok" if not status else "DIFFER " + "; ".join(status[:6]), flush=True)
            bad += bool(status)
        except Exception as e:      # noqa: BLE001
            print(desc, "EXCEPTION %r" % (e,), flush=True)
            traceback.print_exc()
            bad += 1
        finally:
            blockwise._do_block, blockwise.label_graph = real_do_block, real_label
            vi.write_result = real_write
    print("%d trials, %d failures, %.0f s" % (args.trials, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
