#!/usr/bin/env python3
"""Random small cases through THE REFERENCE and the oracle, stage by stage (development container only: the
reference is imported in place from /root/reference with the machinery of gen_golden.py -- its kernels compiled
where they lie for the host, its host stages as they are).  The committed goldens pin the oracle on ~30 hand-picked
cases; this runs the same comparison on random shapes, patch shapes (cubic, anisotropic, 2-d), thresholds,
background rules, value / normalisation / ranking switches, cover / thinning / labelling options.

Compared bit for bit: consensus (positive planes), scores, ranking order, cover, thinning, x-sorted selection,
pair rows (canonical order; the reference's set order as a set), pair affinities, instance map.

  python tests/golden/fuzz_oracle_vs_reference.py [--trials 60] [--seed 1]
"""
import argparse
import json
import os
import sys
import time
import traceback

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
from patchperpix_amd import synth  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def draw(rng):
    kind = str(rng.choice(["cubic", "aniso", "2d"]))
    if kind == "cubic":
        p = int(rng.choice([3, 3, 5]))
        ps = (p, p, p)
        shape = tuple(int(rng.integers(p + 1, p + 9 if p == 3 else p + 6)) for _ in range(3))
    elif kind == "aniso":
        ps = tuple(int(v) for v in rng.choice([3, 3, 5], size=3))
        shape = tuple(int(rng.integers(q + 1, q + 7)) for q in ps)
    else:
        p = int(rng.choice([3, 5, 7]))
        ps = (1, p, p)
        shape = (1, int(rng.integers(p + 2, p + 20)), int(rng.integers(p + 2, p + 20)))
    th = float(rng.choice([0.5, 0.5, 0.6, 0.8, 0.9]))
    bg = str(rng.choice(["less", "inv", "half"]))
    flags = dict(patch_threshold=th, fc_threshold=float(rng.choice([0.5, 0.7])),
                 vi_bg_use_less_than_th=bg == "less", vi_bg_use_inv_th=bg == "inv", vi_bg_use_half_th=bg == "half",
                 consensus_norm_aff=bool(rng.integers(0, 4) != 0), rank_norm_patch_score=bool(rng.integers(0, 4) != 0),
                 rank_int_counter=bool(rng.integers(0, 5) == 0), patch_graph_norm_aff=bool(rng.integers(0, 4) != 0),
                 overlapping_inst=bool(rng.integers(0, 2)), includeSinglePatchCCS=bool(rng.integers(0, 3) != 0),
                 select_patches_for_sparse_data=bool(rng.integers(0, 3) != 0), skipThinCover=bool(rng.integers(0, 2)),
                 mws=bool(rng.integers(0, 2)))
    value = str(rng.choice(["norm_prob", "norm_prob", "prob", "count"]))
    flags.update(consensus_norm_prob_product=value == "norm_prob", consensus_prob_product=value in ("norm_prob", "prob"))
    if value == "count":
        flags["consensus_norm_aff"] = False
    # the rarer switches (each in a few of the goldens): interleaved count, flipped consensus axes, the two optional
    # branches of the cover, one instance per channel / no overlap per channel
    if rng.integers(0, 4) == 0:
        flags["consensus_interleaved_cnt"] = True
    if rng.integers(0, 6) == 0:
        flags["flip_cons_arr_axes"] = True
    if rng.integers(0, 5) == 0:
        flags["mark_close_neighboorhood"] = True
    if rng.integers(0, 5) == 0:
        flags["select_patches_overlap_neighborhood"] = True
    r = rng.integers(0, 8)
    if r == 0:
        flags["one_instance_per_channel"] = True
    elif r == 1:
        flags["no_overlap_per_channel"] = True
    cell = [1 if q == 1 else int(rng.integers(3, 8)) for q in ps]
    return dict(shape=list(shape), ps=list(ps), seed=int(rng.integers(1, 100000)), cell=cell,
                overlap=float(rng.choice([0.0, 0.03])), noise=float(rng.choice([0.0, 0.25])), flags=flags)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    if not os.path.isdir(gg.REF_VI):
        sys.exit("reference tree not found (development container only)")
    gg.install_stubs()
    gg.install_fake_cuda_code()
    sys.path.insert(0, gg.REF_VI)
    import logging
    logging.basicConfig(level=logging.ERROR)
    from oracle import ppp_oracle as orc
    rng = np.random.default_rng(args.seed)
    bad = 0
    t0 = time.time()
    for trial in range(args.trials):
        cfg = draw(rng)
        ps = cfg["ps"]
        skw = dict(seed=cfg["seed"], cell=cfg["cell"], overlap_frac=cfg["overlap"])
        if cfg["noise"]:
            skw["noise"] = cfg["noise"]
        case = synth.make_case(tuple(cfg["shape"]), tuple(ps), **skw)
        case["patchshape"] = list(ps)
        # (the goldens hold float16-exact values; so do these)
        case["pred"] = case["pred"].astype(np.float16).astype(np.float32)
        t1 = time.time()
        try:
            ref = gg.run_reference(case, cfg["flags"])
        except Exception as e:       # noqa: BLE001
            print("trial %d %s: REFERENCE RAISED %r" % (trial, json.dumps(cfg), e), flush=True)
            continue
        t_ref = time.time() - t1
        status = []
        try:
            kw = dict(gg.FLYLIGHT)
            kw.update(cfg["flags"])
            out = orc.to_instance_seg(case["pred"], case["foreground"], case["foreground"].copy(), case["numinst"], ps, **kw)
            early = int(ref["early_out"])
            if early in (1, 2):
                if "cons" in out or out["instances"].any():
                    status.append("EARLY OUT")
            else:
                if not np.array_equal(bits(orc.positive_planes(out["cons"], ps)), bits(ref["cons_pos"])):
                    status.append("CONSENSUS")
                if not np.array_equal(bits(out["scores"]), bits(ref["scores"])):
                    status.append("SCORES")
                if not np.array_equal(out["ranked_coords"], ref["ranked_coords"]):
                    status.append("RANK ORDER")
                if not np.array_equal(out["cover_coords"], ref["cover_coords"]):
                    status.append("COVER")
                if "thin_coords" in ref and not np.array_equal(out.get("thin_coords"), ref["thin_coords"]):
                    status.append("THINNING")
                if not np.array_equal(out["selected_sorted"], ref["selected_sorted"]):
                    status.append("SELECTION")
                if early == 3:
                    if "pairs" in out:
                        status.append("PAIRS where the reference has none")
                else:
                    if not np.array_equal(out["pairs"], ref["pairs"]):
                        status.append("PAIRS")
                    elif not np.array_equal(bits(out["aff"]), bits(ref["aff"])):
                        status.append("AFFINITIES")
                    if not np.array_equal(out["instances"], ref["instances"]):
                        status.append("INSTANCES")
        except Exception as e:       # noqa: BLE001
            status.append("ORACLE RAISED %r" % (e,))
            traceback.print_exc()
        n_inst = int(np.max(ref["instances"])) if "instances" in ref else 0
        print("trial %d %s early %d instances %d reference %.1f s: %s" % (trial, json.dumps(cfg), int(ref["early_out"]), n_inst, t_ref,
                                                                           "ok" if not status else "DIFFER " + "; ".join(status)), flush=True)
        bad += bool(status)
    print("%d trials, %d failures, %.0f s" % (args.trials, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
