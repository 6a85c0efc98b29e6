#!/usr/bin/env python3
"""Random tiny cases through the reference's NumPy path (``cuda=False``: fillLookup / computeFGBGsets /
create_consensus_array / rank_patches / the cover / computePatchGraph's NumPy branch / affGraphToInstances,
called by gen_golden_numpy_path.run) and oracle/ppp_oracle_np.py, stage by stage (development container only;
the committed np_*.npz goldens are six hand-picked cases of the same comparison).

  python tests/golden/fuzz_oracle_np_vs_reference.py [--trials 25] [--seed 1]
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
import gen_golden_numpy_path as gnp  # noqa: E402
from patchperpix_amd import synth  # noqa: E402


def dense_votes(out, ps, shape, onp):
    """the reference's sparse (L_ref, z, y, x) -> value list as the oracle's dense plane layout
    (tests/test_numpy_semantics.py::NpGolden.votes)"""
    ns1, ns2 = 2 * ps[1], 2 * ps[2]
    wy, wx = 2 * ps[1] - 1, 2 * ps[2] - 1
    dense = np.zeros((onp.n_planes(ps),) + tuple(shape), dtype=np.int16)
    idx, val = out["cons_index"], out["cons_value"]
    if len(idx) == 0:
        return dense
    L = idx[:, 0].astype(np.int64)
    dx = L % ns2
    m = L // ns2 + (dx > ps[2] - 1)
    dx = np.where(dx > ps[2] - 1, dx - ns2, dx)
    dy = m % ns1
    dz = m // ns1 + (dy > ps[1] - 1)
    dy = np.where(dy > ps[1] - 1, dy - ns1, dy)
    q = (dz * wy + dy) * wx + dx
    dense[q, idx[:, 1], idx[:, 2], idx[:, 3]] = val
    return dense


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=25)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    gg.install_stubs()
    gg.install_fake_cuda_code()
    sys.path.insert(0, gg.REF_VI)
    import logging
    logging.basicConfig(level=logging.ERROR)
    from oracle import ppp_oracle as orc
    from oracle import ppp_oracle_np as onp
    rng = np.random.default_rng(args.seed)
    bad = 0
    t0 = time.time()
    for trial in range(args.trials):
        if rng.integers(0, 2) == 0:
            p = int(rng.choice([3, 5]))
            ps = (1, p, p)
            shape = (1, int(rng.integers(p + 2, p + 12)), int(rng.integers(p + 2, p + 12)))
        else:
            ps = (3, 3, 3)
            shape = tuple(int(rng.integers(4, 9)) for _ in range(3))
        th = float(rng.choice([0.5, 0.4, 0.7, 0.9]))
        flags = dict(patch_threshold=th, skipThinCover=bool(rng.integers(0, 2)), mws=bool(rng.integers(0, 2)),
                     includeSinglePatchCCS=bool(rng.integers(0, 2)))
        seed = int(rng.integers(1, 100000))
        cell = [1 if q == 1 else int(rng.integers(3, 7)) for q in ps]
        cfg = dict(shape=list(shape), ps=list(ps), seed=seed, cell=cell, flags=flags)
        case = synth.make_case(shape, ps, seed=seed, cell=cell, overlap_frac=float(rng.choice([0.0, 0.03])),
                               noise=float(rng.choice([0.0, 0.3])))
        case["patchshape"] = list(ps)
        case["pred"] = case["pred"].astype(np.float16).astype(np.float32)
        t1 = time.time()
        try:
            res = gnp.run(case, flags)
        except Exception as e:       # noqa: BLE001
            print("trial %d %s: REFERENCE RAISED %r" % (trial, json.dumps(cfg), e), flush=True)
            continue
        out, kw = res if isinstance(res, tuple) else (res, None)
        t_ref = time.time() - t1
        status = []
        try:
            k = dict(gg.FLYLIGHT)
            k.update(flags)
            k.update(cuda=False, removeIntersection=False, sample=1.0, max_total_patch_distance_in_ps_multiples=2,
                     save_no_intermediates=True, result_folder="/tmp")
            pred, fg = case["pred"], case["foreground"].astype(bool)
            overlap = 1 * (case["numinst"] > 1)
            mask = fg.copy()
            mask[overlap > 0] = 0
            votes = onp.consensus(pred, fg, list(ps), th)
            if not np.array_equal(votes, dense_votes(out, list(ps), shape, onp)):
                status.append("VOTES")
            cs, scores = onp.rank(pred, fg, votes, list(ps), th)
            rc, rs = onp.ranked(cs, scores)
            if not (np.array_equal(rc, out["ranked_coords"]) and np.array_equal(rs, out["ranked_scores"])):
                status.append("RANKS")
            sel = orc.foreground_cover(rc, rs, overlap, mask, pred, list(ps), **k)
            if not np.array_equal(rc[sel], out["cover_coords"]):
                status.append("COVER")
            chosen = rc[sel]
            if "thin_coords" in out:
                chosen = chosen[orc.thin_cover(chosen, mask, pred, list(ps), **k)]
                if not np.array_equal(chosen, out["thin_coords"]):
                    status.append("THINNING")
            srt = chosen[np.argsort(chosen[:, 2], kind="stable")]
            if not np.array_equal(srt, out["selected_sorted"]):
                status.append("SELECTION")
            if int(out["has_pairs"]) and not status:
                rows, w = onp.patch_graph(pred, mask, overlap, votes, srt, list(ps), th, include_single=k["includeSinglePatchCCS"])
                nodes, edges = orc._graph_edges(rows, w, keep_zero=True)
                if [list(u) + list(v) for u, v, _ in edges] != out["edge_rows"].tolist() or [int(x) for _, _, x in edges] != out["edge_weight"].tolist():
                    status.append("EDGES")
                inst = orc.label(rows.astype(np.uint32), w, pred, list(ps), fg.shape, keep_zero_edges=True, **k)
                if not np.array_equal(inst, out["instances"]):
                    status.append("INSTANCES")
        except Exception as e:       # noqa: BLE001
            status.append("ORACLE RAISED %r" % (e,))
            traceback.print_exc()
        print("trial %d %s instances %d edges %d reference %.1f s: %s" % (
            trial, json.dumps(cfg), int(np.max(out["instances"])), len(out.get("edge_weight", [])), t_ref,
            "ok" if not status else "DIFFER " + "; ".join(status)), flush=True)
        bad += bool(status)
    print("%d trials, %d failures, %.0f s" % (args.trials, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
