#!/usr/bin/env python3
"""Random small cases of the reference's NumPy path (``cuda=False`` semantics: int16 votes, integer ranks,
all-pairs graph weights) computed on the device against oracle/ppp_oracle_np.py (development aid; under tests/
because it calls the oracle; the fixed cases are tests/test_numpy_semantics.py).

  python tests/fuzz_numpy_path_gpu.py [--trials 40] [--seed 1]
"""
import argparse
import os
import sys
import time
import traceback

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch
    from oracle import ppp_oracle_np as onp
    from patchperpix_amd import synth
    from patchperpix_amd.flags import FLYLIGHT
    from patchperpix_amd.vote_instances import numpy_semantics as ns
    rng = np.random.default_rng(args.seed)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()      # noqa: E731
    bad = 0
    t0 = time.time()
    for trial in range(args.trials):
        if rng.integers(0, 3) == 0:
            p = int(rng.choice([3, 5, 7]))
            ps = (1, p, p)
            shape = (1, int(rng.integers(p + 3, p + 16)), int(rng.integers(p + 3, p + 16)))
        else:
            ps = tuple(int(v) for v in rng.choice([3, 3, 5], size=3))
            shape = tuple(int(rng.integers(q + 2, q + 7)) for q in ps)
        th = float(rng.choice([0.5, 0.45, 0.7, 0.9]))
        seed = int(rng.integers(1, 100000))
        step = int(rng.integers(2, 9))
        desc = "trial %d shape %s ps %s th %.2f seed %d step %d" % (trial, shape, ps, th, seed, step)
        try:
            r2 = np.random.default_rng(seed)
            case = synth.make_case(shape, list(ps), seed=seed, cell=[max(1, min(5, s)) for s in shape], overlap_frac=0.03)
            pred = (case["pred"] + r2.uniform(-0.3, 0.3, size=case["pred"].shape)).astype(np.float16).astype(np.float32)
            fg = case["foreground"].astype(bool)
            overlap = 1 * (case["numinst"] > 1)
            mask = fg.copy()
            mask[overlap > 0] = 0
            kw = dict(FLYLIGHT, patch_threshold=th, cuda=False, removeIntersection=False)
            votes_o = onp.consensus(pred, fg, list(ps), th)
            cs, sc = onp.rank(pred, fg, votes_o, list(ps), th)
            rc, rs = onp.ranked(cs, sc)
            pd, fd = dev(pred), dev(fg.astype(np.uint8))
            votes = ns.create_consensus_array(pd, fd, list(ps), **kw)
            status = []
            if not np.array_equal(votes.cpu().numpy(), votes_o):
                status.append("VOTES")
            ranked, _ = ns.rank_patches(pd, fd, votes, fg, list(ps), **kw)
            if not (np.array_equal(ranked.coords, rc) and np.array_equal(ranked.scores.astype(np.int64), rs)):
                status.append("RANKS")
            sel = rc[::step]
            if len(sel):
                sel = sel[np.argsort(sel[:, 2], kind="stable")]
                rows_o, w_o = onp.patch_graph(pred, mask, overlap, votes_o, sel, list(ps), th, include_single=True)
                rows, w = ns.computePatchGraph(sel, pd, mask, overlap, votes, list(ps), **kw)
                if not (np.array_equal(rows, rows_o) and np.array_equal(w, w_o)):
                    status.append("GRAPH")
            print(desc, "centres", len(rc), ":", "ok" if not status else "DIFFER " + " ".join(status), flush=True)
            bad += bool(status)
        except Exception as e:      # noqa: BLE001
            print(desc, "EXCEPTION %r" % (e,), flush=True)
            traceback.print_exc()
            bad += 1
    print("%d trials, %d failures, %.0f s" % (args.trials, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
