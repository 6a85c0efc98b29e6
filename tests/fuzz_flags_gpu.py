#!/usr/bin/env python3
"""Random shapes, patch shapes and FLAGS: the HIP pipeline against the CPU oracle (development aid; lives
under tests/ because it calls the oracle; the fixed cases are tests/test_gpu_parity.py).

Every trial draws a volume (3-d, or one slice with 2-d patches), a patch shape (cubic, anisotropic, 2-d up
to 25 wide), thresholds, the background rule, the value / normalisation / ranking switches, cover and
labelling options and a prediction (float16-exact or perturbed in float32, some values pinned to the
threshold, to 0 and to 1), runs ``vote_instances.to_instance_seg`` on the GPU and the oracle on the
host, and compares pair rows, pair affinities (bit patterns) and the instance map.

  python tests/fuzz_flags_gpu.py [--trials 40] [--seed 1]
"""
import argparse
import json
import os
import sys
import time
import traceback

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def draw(rng):
    kind = str(rng.choice(["cubic", "cubic", "aniso", "2d"]))
    if kind == "cubic":
        p = int(rng.choice([3, 5, 7, 9], p=[0.2, 0.35, 0.3, 0.15]))
        ps = (p, p, p)
    elif kind == "aniso":
        ps = tuple(int(v) for v in rng.choice([3, 5, 7, 9], size=3, p=[0.4, 0.3, 0.2, 0.1]))
    else:
        p = int(rng.choice([5, 9, 11, 25]))
        ps = (1, p, p)
    C = int(np.prod(ps))
    # the oracle visits C^2 pixel pairs per foreground voxel: bound the volume by that
    budget = 2e9
    vmax = max(int(budget / (C * C)), 2 * int(np.prod([q + 1 for q in ps])))
    long_x = bool(rng.integers(0, 4) == 0)      # lines of more than one 64-voxel run (the packed S1 kernel's flattened runs)
    while True:
        shape = tuple(1 if q == 1 else int(rng.integers(q + 1, q + 28)) for q in ps)
        if kind == "2d":
            shape = (1, int(rng.integers(ps[1] + 4, 90)), int(rng.integers(ps[2] + 4, 90)))
        if long_x:
            shape = shape[:2] + (int(rng.integers(60, 200)),)
        if int(np.prod(shape)) <= vmax:
            break
        long_x = long_x and rng.integers(0, 4) != 0
    th = float(rng.choice([0.5, 0.5, 0.5, 0.6, 0.8, 0.9]))
    bg = str(rng.choice(["less", "inv", "half"]))
    flags = dict(patch_threshold=th, fc_threshold=float(rng.choice([0.5, 0.5, 0.7])),
                 vi_bg_use_less_than_th=bg == "less", vi_bg_use_inv_th=bg == "inv", vi_bg_use_half_th=bg == "half",
                 consensus_norm_aff=bool(rng.integers(0, 4) != 0), rank_norm_patch_score=bool(rng.integers(0, 4) != 0),
                 rank_int_counter=bool(rng.integers(0, 5) == 0), patch_graph_norm_aff=bool(rng.integers(0, 4) != 0),
                 overlapping_inst=bool(rng.integers(0, 2)), includeSinglePatchCCS=bool(rng.integers(0, 3) != 0),
                 select_patches_for_sparse_data=bool(rng.integers(0, 3) != 0), skipThinCover=bool(rng.integers(0, 2)),
                 mws=bool(rng.integers(0, 2)))
    value = str(rng.choice(["norm_prob", "norm_prob", "prob", "count"]))
    flags.update(consensus_norm_prob_product=value == "norm_prob", consensus_prob_product=value in ("norm_prob", "prob"))
    if value == "count":
        flags["consensus_norm_aff"] = False       # (the reference asserts: no normalising of counted votes)
    cell = [1 if q == 1 else int(rng.integers(max(3, q // 2 + 1), 2 * q + 4)) for q in ps]
    return dict(shape=list(shape), ps=list(ps), seed=int(rng.integers(1, 100000)), cell=cell,
                overlap=float(rng.choice([0.0, 0.02, 0.05])), noise=float(rng.choice([0.0, 0.2, 0.35])),
                perturb=str(rng.choice(["f16", "f32", "pinned"])), half_input=bool(rng.integers(0, 2)), flags=flags)


def make_pred(cfg, synth):
    rng = np.random.default_rng(cfg["seed"])
    kw = dict(seed=cfg["seed"], cell=cfg["cell"], overlap_frac=cfg["overlap"])
    if cfg["noise"]:
        kw["noise"] = cfg["noise"]
    c = synth.make_case(tuple(cfg["shape"]), tuple(cfg["ps"]), **kw)
    pred = c["pred"].astype(np.float32)
    if cfg["perturb"] == "f32":
        pred = (pred * rng.uniform(0.97, 1.0, size=pred.shape)).astype(np.float32)
    elif cfg["perturb"] == "pinned":
        th = np.float32(cfg["flags"]["patch_threshold"])
        r = rng.uniform(size=pred.shape)
        pred = pred.copy()
        pred[r < 0.02] = th
        pred[(r >= 0.02) & (r < 0.04)] = 0.0
        pred[(r >= 0.04) & (r < 0.06)] = 1.0
        pred[(r >= 0.06) & (r < 0.07)] = np.float32(1.0) - th
    if cfg.get("half_input") and cfg["perturb"] != "f32":
        # float16-exact values: the oracle sees their widening
        pred = pred.astype(np.float16).astype(np.float32)
    return c, pred


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cfg")
    args = ap.parse_args()
    from oracle import ppp_oracle as orc
    from patchperpix_amd import synth
    from patchperpix_amd.flags import FLYLIGHT
    from patchperpix_amd.vote_instances import vote_instances as vi
    rng = np.random.default_rng(args.seed)
    bad = unsupported = 0
    t0 = time.time()
    for trial in range(1 if args.cfg else args.trials):
        cfg = json.loads(args.cfg) if args.cfg else draw(rng)
        ps = cfg["ps"]
        kw = dict(FLYLIGHT, **cfg["flags"])
        c, pred = make_pred(cfg, synth)
        t1 = time.time()
        try:
            ref = orc.to_instance_seg(pred, c["foreground"], c["foreground"].copy(), c["numinst"], ps, **kw)
        except Exception as e:       # noqa: BLE001
            print("trial %d %s: ORACLE REFUSED %r" % (trial, json.dumps(cfg), e), flush=True)
            unsupported += 1
            continue
        t_or = time.time() - t1
        status = "ok"
        try:
            # (half of the float16-exact cases enter as float16 arrays: the __half instantiations of the kernels)
            as16 = bool(cfg.get("half_input")) and cfg["perturb"] != "f32"
            pred_in = pred.astype(np.float16) if as16 else pred
            inst, _ = vi.to_instance_seg(pred_in.copy(), c["foreground"].copy(), c["foreground"].copy(), c["numinst"].copy(), ps, **kw)
            if not np.array_equal(inst, ref["instances"]):
                d = np.argwhere(inst != ref["instances"])
                status = "INSTANCES DIFFER: %d voxels" % len(d)
            inter = vi.to_instance_seg(pred_in.copy(), c["foreground"].copy(), c["foreground"].copy(), c["numinst"].copy(), ps,
                                       **dict(kw, return_intermediates=True))
            if "pairs" in ref and ref["pairs"] is not None and len(ref["pairs"]):
                if inter[0] is None or not np.array_equal(inter[0], ref["pairs"]):
                    status += " | PAIR ROWS DIFFER (%s vs %d)" % (None if inter[0] is None else len(inter[0]), len(ref["pairs"]))
                elif not np.array_equal(np.asarray(inter[1]).view(np.uint32), ref["aff"].view(np.uint32)):
                    n = int(np.count_nonzero(np.asarray(inter[1]).view(np.uint32) != ref["aff"].view(np.uint32)))
                    status += " | AFFINITIES DIFFER in %d of %d rows" % (n, len(ref["aff"]))
            elif inter[0] is not None:
                status += " | PAIRS where the oracle has none"
        except NotImplementedError as e:
            status = "refused: %s" % e
            unsupported += 1
        except Exception as e:       # noqa: BLE001
            status = "EXCEPTION %r" % (e,)
            traceback.print_exc()
        n_inst = int(ref["instances"].max())
        print("trial %d %s instances %d oracle %.1f s: %s" % (trial, json.dumps(cfg), n_inst, t_or, status), flush=True)
        bad += not (status == "ok" or status.startswith("refused"))
    print("%d trials, %d failures, %d refused, %.0f s" % (args.trials, bad, unsupported, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
