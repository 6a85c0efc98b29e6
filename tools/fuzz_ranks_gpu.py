#!/usr/bin/env python3
"""Random multi-rank assemblies with the REAL kernels -- the ranks share the one GPU of the box, gloo
is the transport -- against the one-process, one-tile result (development aid; the fixed cases are
tests/test_tiling.py::test_ranks_sharing_one_gpu_equal_whole_volume and ..._provider_local_fields_...).

Every trial draws a volume, a patch size, a flag set and a decomposition: world size 2..4, tiles per
rank in z and y / x, plain tiles / ring of rows / consensus cache, how a rank gets its prediction
(slab with halo, own slices + halo exchange, stale halo refreshed in place, a provider) and its
per-voxel fields (global / local), the cover's zone exchange point to point or by all-reduce.

  python tools/fuzz_ranks_gpu.py [--trials 20] [--seed 1]
  python tools/fuzz_ranks_gpu.py --world 3 --cfg '<the JSON of a trial line>'
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

WORKER = r"""
import json, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {repo!r})
from patchperpix_amd import synth, tiling, backend
from patchperpix_amd import flags as flagsets
cfg = json.loads(os.environ["PPP_FUZZ_CFG"])
torch.cuda.set_device(0)                      # every rank on the one GPU of the box
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ps = cfg["ps"]
shape = tuple(cfg["shape"])
c = synth.make_case(shape, tuple(ps), seed=cfg["seed"], cell=cfg["cell"], overlap_frac=cfg["overlap"])
if cfg.get("empty_top"):
    z = cfg["empty_top"]
    c["pred"][:, z:] = 0.05; c["foreground"][z:] = False; c["numinst"][z:] = 0
kw = dict(flagsets.FLAG_SETS[cfg["flagset"]])
Z = shape[0]
slabs = tiling.plan_slabs(Z, world)
mine = tiling.slabs_of_rank(slabs, rank, world)
lo, hi = tiling.local_range(mine, Z, ps)
if cfg["sub"] > 1:
    a0 = mine[0][0]
    mine = [(a0 + a, a0 + b) for a, b in tiling.plan_slabs(mine[-1][1] - a0, cfg["sub"])]
fields = [c["foreground"].copy(), c["foreground"].copy(), c["numinst"]]
mode = cfg["mode"]
extra = dict(cfg["extra"])
if mode in ("own", "own_local_fields"):
    lo, hi = mine[0][0], mine[-1][1]
if mode in ("own_local_fields", "provider_local_fields"):
    fields = [np.ascontiguousarray(f[lo:hi]) for f in fields]


class ArrayProvider:
    def __init__(self, pred):
        self.pred = pred

    def pred_box(self, box):
        z0, z1, y0, y1, x0, x1 = box
        return torch.from_numpy(np.ascontiguousarray(self.pred[:, z0:z1, y0:y1, x0:x1])).cuda()


if mode.startswith("provider"):
    pred_local = ArrayProvider(c["pred"])
else:
    pred_local = torch.from_numpy(np.ascontiguousarray(c["pred"][:, lo:hi])).cuda()
    if mode == "refresh":
        pred_local[:, :mine[0][0] - lo] = 0
        pred_local[:, mine[-1][1] - lo:] = 0
        extra["_refresh_halo"] = True
if "_yx_tiles" in extra:
    extra["_yx_tiles"] = tuple(extra["_yx_tiles"])
inst, fg = tiling.assemble(pred_local, lo, shape, fields[0], fields[1], fields[2], list(ps), mine,
                           comm=tiling.TorchDistComm(), _instances_dtype=np.uint32, **extra, **kw)
np.save(os.path.join({out!r}, "inst_rank%d.npy" % rank), inst)
json.dump({{k: backend.NOTES.get(k, 0) for k in ("cover_sharded", "thin_sharded", "ring_z", "cons_cache_gb", "cover_p2p")}},
          open(os.path.join({out!r}, "notes_rank%d.json" % rank), "w"))
dist.destroy_process_group()
"""


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


def draw(rng, aniso=False):
    from patchperpix_amd import tiling
    world = int(rng.integers(2, 5))
    p = int(rng.choice([3, 5, 7, 9], p=[0.1, 0.35, 0.35, 0.2]))
    pp = [p, p, p]
    if aniso and rng.integers(0, 2) == 0:
        pp = [int(v) for v in rng.choice([3, 5, 7, 9], size=3, p=[0.35, 0.3, 0.2, 0.15])]
        p = pp[0]
    sub = int(rng.integers(1, 4))
    zmin = world * sub * max(p - 1, 2) + 2
    shape = [int(rng.integers(max(zmin, 2 * p + 2), max(zmin, 2 * p + 2) + 40)), int(rng.integers(2 * pp[1] + 2, 44)), int(rng.integers(2 * pp[2] + 2, 44))]
    if max(pp) == 9:
        shape = [shape[0]] + [min(s, 36) for s in shape[1:]]
    mode = str(rng.choice(["halo", "own", "own_local_fields", "refresh", "provider", "provider_local_fields"]))
    extra = {}
    if rng.integers(0, 2) == 0:
        extra["_yx_tiles"] = [int(rng.integers(1, 3)), int(rng.integers(1, 3))]
    plan = str(rng.choice(["plain", "ring", "cache"]))
    tiles_per_rank = sub * int(np.prod(extra.get("_yx_tiles", [1, 1])))
    if plan == "cache" and not mode.startswith("provider"):
        extra["_cons_cache"] = True
    if plan == "ring" and not mode.startswith("provider") and tiles_per_rank > 1:
        slabs = tiling.plan_slabs(shape[0], world)
        thick = max(max(b - a for a, b in tiling.plan_slabs(s1 - s0, sub)) for s0, s1 in slabs)
        thin = min(min(b - a for a, b in tiling.plan_slabs(s1 - s0, sub)) for s0, s1 in slabs)
        if thin >= p - 1:
            extra["_ring_z"] = thick + tiling.ring_margin(p) + int(rng.integers(0, 5))
    if rng.integers(0, 3) == 0:
        extra["_gather_result"] = False
    cfg = dict(shape=shape, ps=pp, seed=int(rng.integers(1, 10000)), cell=[int(rng.integers(max(4, max(pp)), 2 * max(pp) + 3))] * 3,
               overlap=float(rng.choice([0.0, 0.02])), flagset=str(rng.choice(["shipped", "cc", "nothin_cc"])), sub=sub, mode=mode,
               extra=extra, p2p=str(rng.choice(["1", "1", "0"])))
    if rng.integers(0, 6) == 0:
        cfg["empty_top"] = int(shape[0] * 0.6)
    return world, cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cfg")
    ap.add_argument("--world", type=int)
    ap.add_argument("--aniso", action="store_true", help="half of the trials with anisotropic patch shapes")
    args = ap.parse_args()
    from patchperpix_amd import synth, tiling
    from patchperpix_amd import flags as flagsets
    from patchperpix_amd.vote_instances import vote_instances as vi
    rng = np.random.default_rng(args.seed)
    bad = 0
    t0 = time.time()
    for trial in range(1 if args.cfg else args.trials):
        world, cfg = (args.world, json.loads(args.cfg)) if args.cfg else draw(rng, args.aniso)
        shape, ps = tuple(cfg["shape"]), cfg["ps"]
        c = synth.make_case(shape, tuple(ps), seed=cfg["seed"], cell=cfg["cell"], overlap_frac=cfg["overlap"])
        if cfg.get("empty_top"):
            z = cfg["empty_top"]
            c["pred"][:, z:] = 0.05
            c["foreground"][z:] = False
            c["numinst"][z:] = 0
        kw = dict(flagsets.FLAG_SETS[cfg["flagset"]], _instances_dtype=np.uint32)
        want, _ = vi.to_instance_seg(c["pred"].copy(), c["foreground"].copy(), c["foreground"].copy(), c["numinst"].copy(), list(ps),
                                     **dict(kw, _n_slabs=1, _cons_cache=False))
        status = "ok"
        with tempfile.TemporaryDirectory() as tmp:
            script = os.path.join(tmp, "worker.py")
            open(script, "w").write(WORKER.format(repo=REPO, out=tmp))
            port = free_port()
            env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1", PPP_FUZZ_CFG=json.dumps(cfg),
                       PPP_COVER_P2P=cfg["p2p"])
            r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                                "--master-addr", "127.0.0.1", "--master-port", port, script], env=env, timeout=900,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            notes = ""
            if r.returncode != 0:
                status = "WORKERS FAILED\n" + r.stdout.decode(errors="replace")[-(20000 if args.cfg else 2500):]
            else:
                slabs = tiling.plan_slabs(shape[0], world)
                notes = open(os.path.join(tmp, "notes_rank0.json")).read()
                for rank in range(world):
                    inst = np.load(os.path.join(tmp, "inst_rank%d.npy" % rank))
                    ref = want
                    if cfg["extra"].get("_gather_result") is False:
                        ref = want[slabs[rank][0]:slabs[rank][1]]
                    if inst.shape != ref.shape or not np.array_equal(inst, ref):
                        status = "MISMATCH on rank %d" % rank
                        if inst.shape == ref.shape:
                            d = np.argwhere(inst != ref)
                            status += ": %d voxels, z %d..%d" % (len(d), d[:, 0].min(), d[:, 0].max())
        print("trial %d world %d %s instances %d notes %s: %s" % (trial, world, json.dumps(cfg), int(want.max()), notes, status), flush=True)
        bad += status != "ok"
    print("%d trials, %d failures, %.0f s" % (args.trials, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
