#!/usr/bin/env python3
"""bench.py -- Mvoxels/s assembled by vote_instances on MI355X (BASELINE.json metric).

One "step" = one full pass of the hot path (consensus -> ranking -> cover -> pairs ->
patch graph -> labelling) over one synthetic prediction volume that is already resident in
HBM when the timed region starts.  Prints ONE JSON line ("Measurement").

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

N > 1 is launched by torch.distributed.run, one rank per GPU: ONE volume, N times taller than
the 1-GPU workload (weak scaling), is split into z-slabs with patch-radius halos
(patchperpix_amd/tiling.py); the ranks meet in four RCCL all-reduces and every rank ends with
the complete instance map -- see DESIGN.md "Multi-GPU".
"""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (shape, patchshape, cell size of the synthetic instances)
    # BASELINE.json configs[1]: flylight setup01 3-d crop 140^3, 7x7x7 patch
    "flylight140_p7": ((140, 140, 140), (7, 7, 7), (18, 18, 18)),
    # the global volume of an 8-rank weak-scaling run on one device (how the replicated global
    # stages grow with N; run with --slabs 8)
    "flylight140x8_p7": ((1120, 140, 140), (7, 7, 7), (18, 18, 18)),
    "flylight140x2_p7": ((280, 140, 140), (7, 7, 7), (18, 18, 18)),
    # BASELINE.json configs[0]: wormbodies 2-d, 25x25 patch, one 696x520 image (generic kernels)
    "worm2d_p25": ((1, 520, 696), (1, 25, 25), (1, 40, 40)),
    # reduced variants for quick checks
    "synth64_p5": ((64, 64, 64), (5, 5, 5), (12, 12, 12)),
    "synth96_p7": ((96, 96, 96), (7, 7, 7), (18, 18, 18)),
    # the global volume of a 2- / 3-rank run of synth64_p5 (checks of the multi-rank path)
    "synth64x2_p5": ((128, 64, 64), (5, 5, 5), (12, 12, 12)),
    "synth64x3_p5": ((192, 64, 64), (5, 5, 5), (12, 12, 12)),
    "synth64x8_p5": ((512, 64, 64), (5, 5, 5), (12, 12, 12)),
    # large volumes: the consensus no longer fits, the path tiles itself into z-slabs
    "synth256_p7": ((256, 256, 256), (7, 7, 7), (18, 18, 18)),
    "synth256_p9": ((256, 256, 256), (9, 9, 9), (24, 24, 24)),
    "synth128_p9": ((128, 128, 128), (9, 9, 9), (24, 24, 24)),
    # BASELINE.json configs[2] (needs the tiled consensus path)
    "synth512_p9": ((512, 512, 512), (9, 9, 9), (24, 24, 24)),
}
CPU_SAMPLE = {"worm2d_p25": (1, 60, 60), "synth64x8_p5": (24, 24, 24), "synth64x2_p5": (24, 24, 24), "synth64x3_p5": (24, 24, 24), "flylight140_p7": (28, 28, 28), "flylight140x8_p7": (28, 28, 28), "flylight140x2_p7": (28, 28, 28), "synth96_p7": (28, 28, 28),
              "synth256_p7": (28, 28, 28), "synth256_p9": (26, 26, 26), "synth128_p9": (26, 26, 26),
              "synth64_p5": (24, 24, 24), "synth512_p9": (26, 26, 26)}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def device_labels(torch, shape, cell, seed, z_offset=0):
    """patchperpix_amd.synth.cell_labels evaluated with torch ops on the device (plumbing
    for the synthetic input; not part of the measured path)."""
    def hash_u32(x):
        m = 0xFFFFFFFF
        x = x & m
        x = (x ^ (x >> 16)) & m
        x = (x * 0x7FEB352D) & m
        x = (x ^ (x >> 15)) & m
        x = (x * 0x846CA68B) & m
        x = (x ^ (x >> 16)) & m
        return x
    dev = "cuda"
    zz = torch.arange(shape[0], device=dev, dtype=torch.int64).view(-1, 1, 1) + z_offset
    yy = torch.arange(shape[1], device=dev, dtype=torch.int64).view(1, -1, 1)
    xx = torch.arange(shape[2], device=dev, dtype=torch.int64).view(1, 1, -1)
    cz = zz // cell[0]
    sy = hash_u32(cz * 7919 + seed) % max(1, cell[1])
    cy = (yy + sy) // cell[1]
    sx = hash_u32((cz * 131 + cy) * 104729 + seed + 1) % max(1, cell[2])
    cx = (xx + sx) // cell[2]
    key = ((cz * 1000003 + cy) * 1000003 + cx) & 0xFFFFFFFF
    h = hash_u32(key + (seed * 2654435761) % (1 << 32))
    lab = (h % 65000) + 1
    lab = torch.where(((h >> 16) % 16) == 0, torch.zeros_like(lab), lab)
    return lab.to(torch.int32).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="flylight140_p7", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--flags", default="shipped", choices=["shipped", "cc", "nothin_cc"],
                    help="flag set (patchperpix_amd/flags.py): shipped = default.toml "
                         "[vote_instances] (mws + thinning), cc = mws off, nothin_cc = "
                         "kernels-only pipeline")
    ap.add_argument("--slabs", type=int, default=None,
                    help="force the number of z-slabs of the single-GPU tiled path")
    ap.add_argument("--yx", type=int, nargs=2, default=None, metavar=("NY", "NX"),
                    help="cut every z-slab into NY x NX tiles (single-GPU tiled path)")
    args = ap.parse_args()

    import torch
    from patchperpix_amd import backend
    from patchperpix_amd import flags as flagsets
    from patchperpix_amd.vote_instances import vote_instances as vi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # PPP_BENCH_ONE_GPU=1 (a check of the multi-rank path on a 1-GPU box): every rank on device 0,
    # gloo as the transport -- numbers from such a run mean nothing
    one_gpu = os.environ.get("PPP_BENCH_ONE_GPU", "0") == "1"
    torch.cuda.set_device(0 if one_gpu else local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo" if one_gpu else "nccl")  # "nccl" = RCCL
    n_gpus = max(args.gpus, world)

    shape, ps, cell = WORKLOADS[args.workload]
    kw = dict(flagsets.FLAG_SETS[args.flags])
    from patchperpix_amd import tiling
    if world == 1:
        P = backend.make_params(shape, ps, **kw)
        labels = device_labels(torch, shape, cell, seed=0)
        pred = backend.synth_pred(labels, P, seed=0, f16=True)       # resident in HBM
        fg_host = (labels != 0).cpu().numpy()
        numinst = fg_host.astype(np.uint8)
        gshape = shape

        if args.slabs:
            kw["_n_slabs"] = args.slabs
        if args.yx:
            kw["_yx_tiles"] = tuple(args.yx)
            kw.setdefault("_n_slabs", 1)

        def step():
            inst, _ = vi.to_instance_seg(pred, fg_host.copy(), fg_host.copy(), numinst, ps, **kw)
            return inst
    else:
        # ONE volume, `world` times taller than the 1-GPU workload (weak scaling), split into
        # z-slabs: rank r holds the prediction of its slab + halo only; the ranks meet in four
        # RCCL all-reduces (scores, cover bits, pair affinities, painted slabs)
        gshape = (shape[0] * world, shape[1], shape[2])
        slabs = tiling.plan_slabs(gshape[0], world)
        mine = tiling.slabs_of_rank(slabs, rank, world)
        lo, hi = tiling.local_range(mine, gshape[0], ps)
        # generate rz extra slices on both sides so that every channel of the kept range sees
        # its true neighbours, then keep [lo, hi)
        glo, ghi = max(0, lo - ps[0] // 2), min(gshape[0], hi + ps[0] // 2)
        eshape = (ghi - glo, shape[1], shape[2])
        Pe = backend.make_params(eshape, ps, **kw)
        labels_e = device_labels(torch, eshape, cell, seed=0, z_offset=glo)
        pred = backend.synth_pred(labels_e, Pe, seed=0, f16=True,
                                  voxel_offset=glo * shape[1] * shape[2])
        pred = pred[:, lo - glo:hi - glo].contiguous()
        del labels_e
        fg_host = (device_labels(torch, gshape, cell, seed=0) != 0).cpu().numpy()
        numinst = fg_host.astype(np.uint8)
        comm = tiling.TorchDistComm()

        def step():
            inst, _ = tiling.assemble(pred, lo, gshape, fg_host.copy(), fg_host.copy(), numinst,
                                      ps, mine, comm=comm, **kw)
            return inst
    torch.cuda.synchronize()

    for _ in range(args.warmup):
        inst = step()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    backend.EVENTS = {}
    backend.NOTES.clear()
    barrier()
    host_times = None
    if os.environ.get("PPP_BENCH_STAGES", "1") == "1":
        # per-stage wall clock (adds a device sync around every stage)
        backend.HOST_TIMES = host_times = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        inst = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cpu" if one_gpu else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ev = backend.event_times_ms()
    backend.EVENTS = None
    backend.HOST_TIMES = None

    V = float(np.prod(shape))
    C = int(np.prod(ps))
    value = float(np.prod(gshape)) * args.steps / dt / 1e6
    # Roofline of the scoring kernel (S1 consensus).  Algorithmic bytes per base voxel: the
    # prediction block read once (f16 resident: 2*C) + the overlap mask (1); outputs excluded
    # (SURVEY 8d).  A launch processes the base voxels of its consensus box (the whole volume
    # when untiled; the tiled path launches S1 per slab, twice where it recomputes the
    # consensus for the patch-graph stage).
    s1_total_ms = float(np.sum(ev["consensus"])) if ev.get("consensus") else None
    s1_voxels = backend.NOTES.get("s1_base_voxels", 0)
    roofline = None
    if s1_total_ms:
        alg_bytes = (2.0 * C + 1.0) * s1_voxels
        achieved = alg_bytes / (s1_total_ms * 1e-3) / 1e9
        votes = float(C) * (C - 1) / 2.0 * s1_voxels   # upper bound: every voxel foreground
        roofline = {"bound": "hbm", "kernel": "consensus_v2_kernel", "achieved": achieved,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": None, "algorithmic_bytes_per_launch": alg_bytes / len(ev["consensus"]),
                    "launches": len(ev["consensus"]),
                    "avg_ms": s1_total_ms / len(ev["consensus"]),
                    "note": "VALU-bound (PMC: profiles/): C(C-1)/2 pair votes per voxel",
                    "pair_votes_per_s_upper": votes / (s1_total_ms * 1e-3)}

    if rank == 0 and roofline is not None and args.workload == "flylight140_p7" and world == 1 \
            and not args.slabs:
        roofline.update(pmc_traffic("consensus_v2_kernel"))
    # The kernel that takes most of the step is S5 (patch graph).  SURVEY 8d's secondary figure
    # for it: n_pairs * (2 * C * elem + visited * 4) bytes -- both patches' channel vectors and
    # the consensus entries a pair visits; `visited` is not counted at run time, so the
    # figure below is the lower bound without it (reported next to the S1 roofline, which is
    # the one the metric names).
    roofline_pg = None
    if ev.get("patch_graph") and backend.NOTES.get("n_pairs"):
        pg_ms = float(np.sum(ev["patch_graph"]))
        pg_bytes = float(backend.NOTES["n_pairs"]) * 2.0 * C * 2.0 * args.steps
        ach = pg_bytes / (pg_ms * 1e-3) / 1e9
        roofline_pg = {"bound": "hbm", "kernel": "patch_graph_pa_kernel", "achieved": ach,
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                       "traffic": None, "avg_ms": pg_ms / len(ev["patch_graph"]),
                       "launches": len(ev["patch_graph"]),
                       "pairs_per_s": float(backend.NOTES["n_pairs"]) * args.steps / (pg_ms * 1e-3),
                       "note": "VALU-issue bound (PMC: 72 % VALU busy, profiles/): per-lane "
                               "candidate masks, 12 % of the executed add slots are useful"}
        if rank == 0 and args.workload == "flylight140_p7" and world == 1 and not args.slabs:
            roofline_pg.update(pmc_traffic("patch_graph_pa_kernel"))
    if rank == 0:
        out = {
            "metric": "Mvoxels/sec assembled (vote_instances)", "value": value,
            "unit": "Mvoxels/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16 in, f32 accumulate", "data": "synthetic",
            "config": {"workload": args.workload, "volume": list(shape), "patchshape": list(ps),
                       "pred_dtype": "f16 resident, widened to f32 in registers",
                       "flag_set": args.flags, "flags": flagsets.describe(kw),
                       "instances_found": int(len(np.unique(inst)) - 1),
                       "instances_crc32": int(zlib.crc32(np.ascontiguousarray(inst).tobytes())),
                       "global_volume": list(gshape), "parallelism": "z-slabs x%d" % n_gpus if not args.yx else
                       "z-slabs x%d, yx tiles %dx%d" % ((args.slabs or 1,) + tuple(args.yx))},
            "roofline": roofline,
            "roofline_patch_graph": roofline_pg,
            "kernel_ms": {k: float(np.sum(v) / args.steps) for k, v in ev.items()},
            "stage_wall_ms": {k: float(np.sum(v) / args.steps * 1e3)
                              for k, v in (host_times or {}).items()},
            "workload_stats": dict(backend.NOTES),
        }
        if not args.no_cpu_baseline and world == 1:      # (rank 0 at N = 1 only)
            out["cpu_baseline"] = cpu_baseline(args.workload, ps, cell, kw)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def pmc_traffic(kernel):
    """HBM traffic of `kernel` per launch from the committed rocprofv3 PMC passes
    (profiles/*_pmc_fetch_write.txt: FETCH_SIZE and WRITE_SIZE, separate passes, in KiB).
    Raw counter values; on gfx950 FETCH_SIZE under-reports wide (16 B/lane) streaming reads by
    2x and is uncalibrated for the narrow loads of this kernel (MI355X_MICROARCH.md, HBM)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_fetch_write.txt")))
    if not files:
        return {"traffic": None}
    vals = {}
    for ln in open(files[-1]):
        if kernel in ln:
            parts = ln.split()
            for name in ("FETCH_SIZE", "WRITE_SIZE"):
                if name in parts:
                    vals[name] = float(parts[parts.index(name) + 1]) * 1024.0
    if len(vals) < 2:
        return {"traffic": None}
    return {"traffic": vals["FETCH_SIZE"] + vals["WRITE_SIZE"], "traffic_read": vals["FETCH_SIZE"],
            "traffic_write": vals["WRITE_SIZE"],
            "traffic_source": os.path.relpath(files[-1], ROOT) + " (flylight140_p7, raw counters)"}


def cpu_baseline(workload, ps, cell, kw):
    """The CPU oracle (a port of the reference's kernel arithmetic + host stages) timed on a
    bounded sample of the same generator, one host core."""
    from oracle import ppp_oracle as orc
    from patchperpix_amd import synth
    sshape = CPU_SAMPLE[workload]
    lab = synth.cell_labels(sshape, cell, seed=0)
    pred = synth.pred_from_labels(lab, ps, seed=0)
    fg = lab != 0
    orc.lib()
    t0 = time.perf_counter()
    orc.to_instance_seg(pred, fg, fg.copy(), fg.astype(np.uint8), ps, **kw)
    dt = time.perf_counter() - t0
    return {"value": float(np.prod(sshape)) / dt / 1e6, "unit": "Mvoxels/s", "cores": 1,
            "kind": "port", "seconds": dt,
            "sample": "%s sub-volume of the same generator, %s patch, full pipeline "
                      "(oracle/ppp_oracle)" % ("x".join(map(str, sshape)), "x".join(map(str, ps)))}


if __name__ == "__main__":
    main()
