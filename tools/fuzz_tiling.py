#!/usr/bin/env python3
"""Random tilings against the one-tile result, on the GPU (development aid; the fixed cases are in
tests/test_gpu_parity.py and tests/test_tiling.py).

Every trial draws a volume shape, a patch size, a flag set and a seed, runs the assembly as ONE tile
(the path the goldens and the oracle pin) and then as several random plans -- z-slabs, y / x tiles,
plain / consensus cache / ring of rows, sharded global stage, a forced ranking tile shape -- and
compares pair rows, pair affinities (bit patterns) and the instance map.  Prints one line per trial
and every mismatch with the arguments that reproduce it; exit code 1 if any.

  python tools/fuzz_tiling.py [--trials 60] [--seed 1] [--max 64]
"""
import argparse
import os
import sys
import time
import traceback

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max", type=int, default=64, help="largest extent of an axis")
    ap.add_argument("--plans", type=int, default=3, help="random plans per trial")
    ap.add_argument("--aniso", action="store_true", help="half of the trials with anisotropic patch shapes")
    args = ap.parse_args()
    import torch
    from patchperpix_amd import backend, synth, tiling
    from patchperpix_amd import flags as F
    from patchperpix_amd.vote_instances import vote_instances as vi
    rng = np.random.default_rng(args.seed)
    bad = 0
    t_all = time.time()
    for trial in range(args.trials):
        p = int(rng.choice([3, 5, 7, 9], p=[0.15, 0.3, 0.3, 0.25]))
        ps = (p, p, p)
        if args.aniso and rng.integers(0, 2) == 0:
            # anisotropic patches: compact planes, the gather ranking kernel, no ring / cache (the plans fall back)
            ps = tuple(int(v) for v in rng.choice([3, 5, 7, 9], size=3, p=[0.35, 0.3, 0.2, 0.15]))
            p = ps[0]
        shape = tuple(int(rng.integers(2 * q + 2, max(2 * q + 3, args.max + 1))) for q in ps)
        if max(ps) == 9:      # (keeps a trial in seconds)
            shape = tuple(min(s, 56) for s in shape)
        flagset = str(rng.choice(["shipped", "cc"]))
        f16 = bool(rng.integers(0, 2))
        seed = int(rng.integers(1, 10000))
        cell = int(rng.integers(max(4, max(ps) + 1), 3 * max(ps) + 2))
        kw = dict(F.FLYLIGHT if flagset == "shipped" else F.FLYLIGHT_CC, _instances_dtype=np.uint32)
        P = backend.make_params(shape, ps, **kw)
        lab = synth.cell_labels(shape, [cell] * 3, seed=seed)
        pred = backend.synth_pred(torch.as_tensor(lab.astype(np.int32), device="cuda"), P, seed=seed, f16=f16)
        fg = lab != 0
        numinst = fg.astype(np.uint8)
        if rng.integers(0, 3) == 0:       # some overlap voxels (numinst > 1: excluded from votes and cover)
            ov = (rng.uniform(size=shape) < 0.02) & fg
            numinst[ov] = 2
        fargs = lambda: (fg.copy(), fg.copy(), numinst.copy(), list(ps))     # noqa: E731
        desc = "trial %d: shape %s ps %s %s f16 %d seed %d cell %d" % (trial, shape, "x".join(str(q) for q in ps), flagset, f16, seed, cell)
        try:
            one = dict(kw, _n_slabs=1, _cons_cache=False)
            want = vi.to_instance_seg(pred, *fargs(), **one)[0]
            wi = vi.to_instance_seg(pred, *fargs(), **dict(one, return_intermediates=True))
        except Exception:
            print(desc, "ONE-TILE RUN FAILED")
            traceback.print_exc()
            bad += 1
            continue
        n_pairs = 0 if wi[0] is None else len(wi[0])
        fails = []
        for _ in range(args.plans):
            Z = shape[0]
            n_slabs = int(rng.integers(1, 7))
            yx = (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
            mode = str(rng.choice(["plain", "cache", "ring"]))
            plan = dict(_n_slabs=n_slabs, _yx_tiles=yx, _cons_cache=(mode == "cache"))
            if mode == "ring":
                thick = -(-Z // n_slabs)
                if thick < p - 1:
                    n_slabs = max(1, Z // max(p - 1, 1))
                    plan["_n_slabs"] = n_slabs
                thick = max(b - a for a, b in tiling.plan_slabs(Z, n_slabs))
                plan["_ring_z"] = thick + tiling.ring_margin(p) + int(rng.integers(0, 6))
            if rng.integers(0, 3) == 0:
                plan["_sharded_global"] = True
            env = {}
            if rng.integers(0, 2) == 0:
                env["PPP_RANK_TILE"] = str(rng.choice(["1", "2", "3"]))
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                got = vi.to_instance_seg(pred, *fargs(), **dict(kw, **plan))[0]
                gi = vi.to_instance_seg(pred, *fargs(), **dict(kw, return_intermediates=True, **plan))
                ok = np.array_equal(want, got)
                if wi[0] is None or gi[0] is None:
                    ok = ok and (wi[0] is None) == (gi[0] is None)
                else:
                    ok = ok and np.array_equal(wi[0], gi[0]) and \
                        np.array_equal(np.asarray(wi[1]).view(np.uint32), np.asarray(gi[1]).view(np.uint32))
                if not ok:
                    fails.append((plan, env, "MISMATCH"))
            except Exception as e:       # noqa: BLE001
                fails.append((plan, env, "EXCEPTION %r" % (e,)))
                traceback.print_exc()
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        print("%s instances %d pairs %d: %s" % (desc, int(want.max()), n_pairs, "ok" if not fails else "FAILED"), flush=True)
        for f in fails:
            print("    ", f, flush=True)
        bad += len(fails)
        del pred
    print("%d trials, %d failures, %.0f s" % (args.trials, bad, time.time() - t_all))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
