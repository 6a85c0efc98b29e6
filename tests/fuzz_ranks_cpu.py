#!/usr/bin/env python3
"""Random multi-rank assemblies against the whole-volume oracle, on the CPU (development aid; the fixed
cases are tests/test_tiling.py::test_ranks_gloo_*; lives under tests/ because it calls the oracle).

Every trial draws a volume, flags and a decomposition -- world size 2..4, slabs per rank, y / x tiles,
how the rank gets its prediction (slab with halo, own slices + halo exchange, stale halo refreshed in
place, a provider) and its per-voxel fields (global / local) -- starts the ranks under
torch.distributed.run (gloo; the oracle stands in for the kernels: tests/oracle_ops.py) and compares
every rank's instance map with the oracle's result on the whole volume.

  python tests/fuzz_ranks_cpu.py [--trials 20] [--seed 1]
  python tests/fuzz_ranks_cpu.py --world 3 --cfg '<the JSON of a trial line>'      (one configuration again)
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
sys.path.insert(0, os.path.join(REPO, "tests"))

WORKER = r"""
import json, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
from patchperpix_amd import synth, tiling
from patchperpix_amd.flags import FLYLIGHT_NOTHIN_CC
from oracle_ops import OracleOps
cfg = json.loads(os.environ["PPP_FUZZ_CFG"])
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ps = cfg["ps"]
c = synth.make_case(tuple(cfg["shape"]), tuple(ps), seed=cfg["seed"], cell=cfg["cell"], overlap_frac=cfg["overlap"])
if cfg.get("empty_top"):
    z = cfg["empty_top"]
    c["pred"][:, z:] = 0.05; c["foreground"][z:] = False; c["numinst"][z:] = 0
kw = dict(FLYLIGHT_NOTHIN_CC)
kw.update(cfg["flags"])
Z = c["pred"].shape[1]
slabs = tiling.plan_slabs(Z, cfg["n_slabs"])
mine = tiling.slabs_of_rank(slabs, rank, world)
lo, hi = tiling.local_range(mine, Z, ps)
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
        return torch.from_numpy(np.ascontiguousarray(self.pred[:, z0:z1, y0:y1, x0:x1]))


if mode.startswith("provider"):
    pred_local = ArrayProvider(c["pred"])
else:
    pred_local = torch.from_numpy(np.ascontiguousarray(c["pred"][:, lo:hi]))
    if mode == "refresh":
        pred_local[:, :mine[0][0] - lo] = 0
        pred_local[:, mine[-1][1] - lo:] = 0
        extra["_refresh_halo"] = True
if "_yx_tiles" in extra:
    extra["_yx_tiles"] = tuple(extra["_yx_tiles"])
inst, fg = tiling.assemble(pred_local, lo, c["foreground"].shape, fields[0], fields[1], fields[2], ps, mine,
                           comm=tiling.TorchDistComm(), ops=OracleOps(**kw), **extra, **kw)
np.save(os.path.join({out!r}, "inst_rank%d.npy" % rank), inst)
dist.destroy_process_group()
"""


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cfg", help="run ONE configuration: the JSON a trial line printed")
    ap.add_argument("--world", type=int, help="with --cfg: the number of ranks")
    args = ap.parse_args()
    from oracle import ppp_oracle as orc
    from patchperpix_amd import synth
    from patchperpix_amd.flags import FLYLIGHT_NOTHIN_CC
    rng = np.random.default_rng(args.seed)
    bad = 0
    t0 = time.time()
    for trial in range(1 if args.cfg else args.trials):
        if args.cfg:
            cfg = json.loads(args.cfg)
            world, shape, ps, flags, extra = args.world, cfg["shape"], cfg["ps"], cfg["flags"], cfg["extra"]
        else:
            world = int(rng.integers(2, 6))
            per_rank = int(rng.integers(1, 3))
            ps = [3, 3, 3] if rng.integers(0, 5) else [int(v) for v in rng.choice([[3, 5, 3], [3, 3, 5], [5, 3, 3]])]
            lo_z = max(4 * ps[0], (ps[0] + 1) * world * per_rank)
            shape = [int(rng.integers(lo_z, max(lo_z + 1, 49))), int(rng.integers(2 * ps[1] + 1, 17)), int(rng.integers(2 * ps[2] + 1, 17))]
            flags = dict(skipThinCover=bool(rng.integers(0, 2)), mws=bool(rng.integers(0, 2)))
            if rng.integers(0, 4) == 0:
                flags["select_patches_for_sparse_data"] = False
            mode = str(rng.choice(["halo", "own", "own_local_fields", "refresh", "provider", "provider_local_fields"]))
            extra = {}
            if rng.integers(0, 2) == 0:
                extra["_yx_tiles"] = [int(rng.integers(1, 3)), int(rng.integers(1, 3))]
            if rng.integers(0, 3) == 0 and not mode.startswith("provider"):
                extra["_cons_cache"] = True
            if rng.integers(0, 3) == 0:
                extra["_cover_chunk"] = int(rng.integers(100, 900))
            if rng.integers(0, 3) == 0:
                extra["_gather_result"] = False
            if rng.integers(0, 4) == 0 and "local_fields" not in mode:
                extra["_sharded_global"] = True
            cfg = dict(shape=shape, ps=ps, seed=int(rng.integers(1, 10000)), cell=[int(rng.integers(3, 8))] * 3,
                       overlap=float(rng.choice([0.0, 0.02, 0.05])), flags=flags, n_slabs=world * per_rank, mode=mode, extra=extra,
                       p2p=str(rng.choice(["1", "1", "0"])))
            if rng.integers(0, 6) == 0:
                cfg["empty_top"] = int(shape[0] * 0.6)       # nothing above: ranks whose slabs hold no patch
        c = synth.make_case(tuple(shape), tuple(ps), seed=cfg["seed"], cell=cfg["cell"], overlap_frac=cfg["overlap"])
        if cfg.get("empty_top"):
            z = cfg["empty_top"]
            c["pred"][:, z:] = 0.05
            c["foreground"][z:] = False
            c["numinst"][z:] = 0
        kw = dict(FLYLIGHT_NOTHIN_CC)
        kw.update(flags)
        ref = orc.to_instance_seg(c["pred"], c["foreground"], c["foreground"].copy(), c["numinst"], ps, **kw)["instances"]
        status = "ok"
        with tempfile.TemporaryDirectory() as tmp:
            script = os.path.join(tmp, "worker.py")
            open(script, "w").write(WORKER.format(repo=REPO, out=tmp))
            port = free_port()
            env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1", PPP_FUZZ_CFG=json.dumps(cfg),
                       PPP_COVER_P2P=cfg.get("p2p", "1"))
            r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                                "--master-addr", "127.0.0.1", "--master-port", port, script], env=env, timeout=1200,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            if r.returncode != 0:
                status = "WORKERS FAILED\n" + r.stdout.decode(errors="replace")[-(20000 if args.cfg else 3000):]
            else:
                from patchperpix_amd import tiling
                slabs = tiling.plan_slabs(shape[0], cfg["n_slabs"])
                for rank in range(world):
                    inst = np.load(os.path.join(tmp, "inst_rank%d.npy" % rank))
                    want = ref
                    if extra.get("_gather_result") is False:
                        mine = tiling.slabs_of_rank(slabs, rank, world)
                        want = ref[mine[0][0]:mine[-1][1]]
                    if inst.shape != want.shape or not np.array_equal(inst, want):
                        status = "MISMATCH on rank %d" % rank
                        if inst.shape == want.shape:
                            d = np.argwhere(inst != want)
                            status += ": %d voxels, z %d..%d, ids %s vs %s" % (len(d), d[:, 0].min(), d[:, 0].max(), np.unique(inst[inst != want])[:6].tolist(),
                                                                                np.unique(want[inst != want])[:6].tolist())
                        else:
                            status += ": shape %s vs %s" % (inst.shape, want.shape)
        print("trial %d world %d %s instances %d: %s" % (trial, world, json.dumps(cfg), int(ref.max()), status), flush=True)
        bad += status != "ok"
    print("%d trials, %d failures, %.0f s" % (args.trials, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
