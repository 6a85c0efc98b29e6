#!/usr/bin/env python3
"""Random small cases through the library's NATIVE HOST stages (ppp_host_*: stable ranking order, the
sequential greedy cover with its pixel-threshold passes, set-cover thinning, pair enumeration, mutex
watershed) against the oracle's Python restatement (development aid, CPU only; under tests/ because it
calls the oracle; the fixed cases -- goldens of the reference -- are tests/test_abi_and_host.py).

  python tests/fuzz_host_stages.py [--trials 100] [--seed 1]
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
    ap.add_argument("--trials", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    from oracle import ppp_oracle as orc
    from patchperpix_amd import backend, synth
    from patchperpix_amd.flags import FLYLIGHT
    from test_abi_and_host import _bits
    rng = np.random.default_rng(args.seed)
    bad = 0
    t0 = time.time()
    for trial in range(args.trials):
        if rng.integers(0, 4) == 0:
            p = int(rng.choice([3, 5, 7]))
            ps = [1, p, p]
            shape = (1, int(rng.integers(p + 3, p + 30)), int(rng.integers(p + 3, p + 30)))
        else:
            ps = [int(v) for v in rng.choice([3, 3, 5], size=3)]
            shape = tuple(int(rng.integers(q + 2, q + 12)) for q in ps)
        flags = dict(select_patches_for_sparse_data=bool(rng.integers(0, 2)), skipThinCover=bool(rng.integers(0, 2)),
                     includeSinglePatchCCS=bool(rng.integers(0, 2)), mws=bool(rng.integers(0, 2)),
                     overlapping_inst=bool(rng.integers(0, 2)), fc_threshold=float(rng.choice([0.5, 0.7])))
        seed = int(rng.integers(1, 100000))
        desc = "trial %d shape %s ps %s seed %d %s" % (trial, shape, ps, seed, {k: v for k, v in flags.items()})
        try:
            c = synth.make_case(shape, tuple(ps), seed=seed, cell=[max(1, min(int(rng.integers(3, 8)), s)) for s in shape],
                                overlap_frac=float(rng.choice([0.0, 0.03])), noise=float(rng.choice([0.0, 0.3])))
            kw = dict(FLYLIGHT, **flags)
            pred = c["pred"].astype(np.float32)
            fg = c["foreground"]
            ref = orc.to_instance_seg(pred, fg, fg.copy(), c["numinst"], ps, **kw)
            if "ranked_coords" not in ref:
                print(desc, ": early out", flush=True)
                continue
            status = []
            rad = [q // 2 for q in ps]
            lin = backend.host_rank_order(ref["scores"], fg, ps)
            coords = np.stack(np.unravel_index(lin, shape), axis=1)
            if not np.array_equal(coords, ref["ranked_coords"]):
                status.append("RANK ORDER")
            overlap = (c["numinst"] > 1)
            mask = fg.copy()
            mask[overlap] = 0
            running, _owner = backend.padded_mask(mask)
            radslice = tuple(slice(rad[i], shape[i] - rad[i]) for i in range(3))
            remaining = int(np.count_nonzero(running[radslice]))
            bits = _bits(pred, coords, kw["fc_threshold"])
            selected = np.zeros(len(lin), dtype=np.uint8)
            pix_ths = [0] if kw["select_patches_for_sparse_data"] else [t for t in [500, 100, 50, 10, 0] if t < int(np.prod(ps) / 2)]
            ov8 = overlap.astype(np.uint8)
            for t in pix_ths:
                remaining, _ = backend.host_cover_pass(running, ov8, ps, lin, np.ascontiguousarray(ref["ranked_scores"]), bits, t, None,
                                                       selected, remaining, marked=None)
                if remaining < 1:
                    break
            cover = coords[selected.astype(bool)]
            cover_lin, cover_bits = lin[selected.astype(bool)], bits[selected.astype(bool)]
            if not np.array_equal(cover, ref["cover_coords"]):
                status.append("COVER (%d vs %d)" % (len(cover), len(ref["cover_coords"])))
            sel = ref["cover_coords"]
            if "thin_coords" in ref and not status:
                keep = backend.host_thin_cover(mask.astype(np.uint8), ps, np.ascontiguousarray(cover_lin), np.ascontiguousarray(cover_bits))
                sel = cover[keep]
                if not np.array_equal(sel, ref["thin_coords"]):
                    status.append("THINNING (%d vs %d)" % (len(sel), len(ref["thin_coords"])))
                sel = ref["thin_coords"]
            sorted_zyx, pairs = backend.host_patch_pairs(sel, ps, include_single=kw["includeSinglePatchCCS"])
            if not np.array_equal(sorted_zyx, ref["selected_sorted"]):
                status.append("SORTED SELECTION")
            if ("pairs" in ref) != (pairs is not None) or (pairs is not None and not np.array_equal(pairs, ref["pairs"])):
                status.append("PAIRS")
            if kw["mws"] and "pairs" in ref:
                nodes, labels, n_labels = backend.host_mws(ref["pairs"], ref["aff"], shape)
                ccs = orc.mutex_watershed(ref["pairs"], ref["aff"])
                want = {}
                for k, cc in enumerate(ccs):
                    for n in cc:
                        want[tuple(int(v) for v in n)] = k + 1
                got = {tuple(int(v) for v in n): int(l) for n, l in zip(nodes, labels)}
                if got != want:
                    status.append("MUTEX WATERSHED (%d vs %d nodes)" % (len(got), len(want)))
            print(desc, "ranked %d cover %d pairs %d:" % (len(lin), len(ref["cover_coords"]), 0 if "pairs" not in ref else len(ref["pairs"])),
                  "ok" if not status else "DIFFER " + "; ".join(status), flush=True)
            bad += bool(status)
        except Exception as e:      # noqa: BLE001
            print(desc, "EXCEPTION %r" % (e,), flush=True)
            traceback.print_exc()
            bad += 1
    print("%d trials, %d failures, %.0f s" % (args.trials, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
