#!/usr/bin/env python3
"""clean_mask / get_offsets of the reference's stitch_patch_graph.py (:46-57, :425-440; imported in place with the
stubs of gen_golden_blockwise.py) against this package's (vote_instances/stitch_patch_graph.py, blockwise.py) on
random masks and shapes.  Development container only.

  python tests/golden/fuzz_stitch_helpers_vs_reference.py [--trials 400]
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
import gen_golden_blockwise as gb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    gg.install_stubs()
    gg.install_fake_cuda_code()
    gb.install_blockwise_stubs({})
    sys.path.insert(0, gg.REF_VI)
    import logging
    logging.basicConfig(level=logging.ERROR)
    import stitch_patch_graph as ref
    from patchperpix_amd import blockwise
    from patchperpix_amd.vote_instances import stitch_patch_graph as mine
    rng = np.random.default_rng(args.seed)
    bad = 0
    for trial in range(args.trials):
        nd = int(rng.integers(2, 4))
        shape = tuple(int(rng.integers(1, 13)) for _ in range(nd))
        mask = rng.random(shape) < rng.uniform(0.05, 0.8)
        structure = np.ones([3] * nd) if rng.integers(0, 2) else None
        size = int(rng.integers(0, 12))
        a = ref.clean_mask(mask.copy(), structure, size)
        b = mine.clean_mask(mask.copy(), structure, size)
        status = []
        if a.shape != b.shape or not np.array_equal(np.asarray(a, dtype=bool), np.asarray(b, dtype=bool)):
            status.append("clean_mask")
        total = [int(rng.integers(1, 40)) for _ in range(3)]
        chunk = [int(rng.integers(1, 20)) for _ in range(3)]
        oa = [tuple(int(v) for v in o) for o in ref.get_offsets(np.array(total), chunk)]
        ob = [tuple(int(v) for v in o) for o in blockwise.get_offsets(np.array(total), chunk)]
        if oa != ob:
            status.append("get_offsets")
        if status:
            bad += 1
            print("trial", trial, shape, size, total, chunk, "DIFFER", status)
    print("%d trials, %d failures" % (args.trials, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
