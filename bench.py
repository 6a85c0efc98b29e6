#!/usr/bin/env python3
"""bench.py -- Mvoxels/s assembled by vote_instances on MI355X (BASELINE.json metric).

One "step" = one full pass of the hot path (consensus -> ranking -> greedy cover -> set-cover
thinning -> pairs -> patch graph -> mutex watershed / components -> painting) over one synthetic
prediction volume that is already resident in HBM when the timed region starts.  Prints ONE
JSON line ("Measurement").

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--flags SET]

Default workload: ``synth512_p9`` = BASELINE.json configs[2] (512^3 volume, 9^3 patch, dense
foreground, float16 prediction resident: 196 GB), the configuration the north star's target is
stated on, with the reference's SHIPPED flylight flag set (``--flags shipped``: mutex watershed
and thinning on; exact entries in the JSON's config.flags) and the uint32 instance ids of the
reference's blockwise entry.  A step of that volume takes the better part of a minute; so that a
default run always ends inside the driver's window, the first warm-up step is timed and, if
(W + K) steps would exceed PPP_BENCH_BUDGET_S (default 1500 s), the timed loop falls back to
``flylight140_p7`` (configs[1]) and the one 512^3 step is reported as ``config2_end_to_end``.

At N = 1 the same line also carries
  * ``roofline`` for the scoring kernel S1 on the timed workload (HIP events of this run; for
    flylight140_p7 additionally ``roofline_north_star``: the kernel over the 512^3 / 9^3 volume),
  * ``variants`` (flylight140_p7 only): ms per step of the other flag sets,
  * ``cpu_baseline``: the CPU oracle (C, -O3) on host cores -- one thread and all threads.

N > 1 is launched by torch.distributed.run, one rank per GPU: ONE volume, split into z-slabs
with patch-radius halos (patchperpix_amd/tiling.py); --scaling strong (default): the 1-GPU
volume split over the ranks; --scaling weak: a volume N times taller.
"""
import argparse
import hashlib
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
    # BASELINE.json configs[3]: 1024^3 / 9^3 over the 8 GPUs of a node.  The float16 prediction
    # is 1.57 TB: it is GENERATED tile by tile (a rank holds one tile + halo, the provider mode of
    # patchperpix_amd/tiling.py), fields and every list are local to the rank
    "synth1024_p9": ((1024, 1024, 1024), (9, 9, 9), (24, 24, 24)),
    # the provider mode on volumes a single development GPU finishes quickly
    "synth256_p9_provider": ((256, 256, 256), (9, 9, 9), (24, 24, 24)),
    "synth128_p7_provider": ((128, 128, 128), (7, 7, 7), (18, 18, 18)),
}
PROVIDER_WORKLOADS = ("synth1024_p9", "synth256_p9_provider", "synth128_p7_provider")
# BASELINE.json configs[4] (ppp+dec): the U-Net predicts a 256-d code per voxel, the decoder
# expands it to the 7^3 patch on the GPU, then the vote.  SURVEY 8(d): the literal 25^3 patch is
# infeasible for any implementation (3.9 PB of consensus); the reference's ppp+dec configuration is
# 7^3 (default_train_code.toml).  Random-init decoder of the shipped architecture (no checkpoint
# ships), random codes: the decoded patches are noise -- a throughput workload, not a segmentation.
WORKLOADS["dec256_p7"] = ((256, 256, 256), (7, 7, 7), (18, 18, 18))
WORKLOADS["dec96_p7"] = ((96, 96, 96), (7, 7, 7), (18, 18, 18))
# variant (ii) of SURVEY 8(d): the 2-d decoder with 25 x 25 patches on 256 slices of 256^2, patch
# shape (1, 25, 25) -- where the reference uses 25-wide patches at all (vote_instances.py:488)
WORKLOADS["dec256x256_p25"] = ((256, 256, 256), (1, 25, 25), (1, 40, 40))
WORKLOADS["dec8x256_p25"] = ((8, 256, 256), (1, 25, 25), (1, 40, 40))
WORKLOADS["dec32x256_p25"] = ((32, 256, 256), (1, 25, 25), (1, 40, 40))
DECODE_WORKLOADS = ("dec256_p7", "dec96_p7", "dec256x256_p25", "dec8x256_p25", "dec32x256_p25")
DECODER = dict(activation="relu", num_fmaps=[64, 128], downsample_factors=[[2, 2, 2], [2, 2, 2]],
               upsampling="resize_conv", kernel_size=3, num_repetitions=2, padding="same",
               code_fmaps=32, code_units=256, input_shape_squeezed=(7, 7, 7))
# the same Autoencoder in 2-d (torch_model.py:452-544 is shape-generic): 256 units -> 16 x 4^2,
# three stages 4 -> 8 -> 16 -> 32, crop to 25 x 25
DECODER_2D = dict(activation="relu", num_fmaps=[32, 64, 128], downsample_factors=[[2, 2], [2, 2], [2, 2]],
                  upsampling="resize_conv", kernel_size=3, num_repetitions=2, padding="same",
                  code_fmaps=16, code_units=256, input_shape_squeezed=(25, 25))
DEFAULT_WORKLOAD = "synth512_p9"      # BASELINE.json configs[2]
FALLBACK_WORKLOAD = "flylight140_p7"  # BASELINE.json configs[1]
NORTH_STAR = ((512, 512, 512), (9, 9, 9), (24, 24, 24))   # BASELINE.json configs[2]
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def device_labels(torch, shape, cell, seed, z_offset=0, y_offset=0, x_offset=0):
    """patchperpix_amd.synth.cell_labels evaluated with torch ops on the device (plumbing
    for the synthetic input; not part of the measured path).  The offsets place `shape` as a box
    inside a larger volume."""
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
    yy = torch.arange(shape[1], device=dev, dtype=torch.int64).view(1, -1, 1) + y_offset
    xx = torch.arange(shape[2], device=dev, dtype=torch.int64).view(1, 1, -1) + x_offset
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


def count_instances(torch, inst):
    """Distinct non-zero ids of the instance map (on the device: np.unique sorts 134 M voxels on
    one host core for half a minute at 512^3)."""
    u = torch.unique(torch.from_numpy(np.ascontiguousarray(inst).astype(np.int64)).cuda())
    return int(u.numel()) - int((u == 0).any().item())


def source_sha16():
    """Fingerprint of the kernel sources: ties a committed rocprof summary to the code it
    was taken from (profiles/*.meta.json)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "patchperpix_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        # (device code only: the host-only translation units ppp_host*.cpp hold no kernel)
        if f.endswith((".hip", ".hpp")) or (f.endswith(".cpp") and not f.startswith("ppp_host")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def s1_roofline(ev_ms, base_voxels, C, kernel="consensus_v2_kernel"):
    """algorithmic bytes of S1 = the prediction block read once, (2 C + 1) bytes per
    base voxel with the float16 prediction resident (+ the overlap mask); outputs excluded."""
    total_ms = float(np.sum(ev_ms))
    alg = (2.0 * C + 1.0) * base_voxels
    ach = alg / (total_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_launch": alg / len(ev_ms), "launches": len(ev_ms),
            "avg_ms": total_ms / len(ev_ms),
            "pair_votes_per_s_upper": C * (C - 1) / 2.0 * base_voxels / (total_ms * 1e-3),
            "note": "the kernel is bound by vector-instruction issue, not by HBM: C(C-1)/2 pair "
                    "votes per foreground voxel at ~4.5 (v3: packed, two slices per lane) or ~11 (v2) "
                    "VALU instructions each (DESIGN.md section 4)"}


def north_star_s1(torch, backend, kw):
    """One pass of the scoring kernel over BASELINE configs[2] (512^3, 9^3, dense foreground,
    float16 prediction resident in HBM): slabs of 16 slices of base voxels into one reused
    consensus buffer (the 1.3 TB consensus of that volume cannot exist at once)."""
    shape, ps, cell = NORTH_STAR
    C = int(np.prod(ps))
    T = 16
    planes = ((2 * ps[0] - 1) * (2 * ps[1] - 1) * (2 * ps[2] - 1) - 1) // 2
    need = 2.0 * C * np.prod(shape) + 5.0 * np.prod(shape) + planes * 4.0 * T * shape[1] * shape[2]
    torch.cuda.empty_cache()
    free = torch.cuda.mem_get_info()[0]
    if need > 0.95 * free:
        return {"skipped": "needs %.0f GB of HBM, %.0f GB free" % (need / 1e9, free / 1e9)}
    P = backend.make_params(shape, ps, **kw)
    labels = device_labels(torch, shape, cell, seed=0)
    pred = backend.synth_pred(labels, P, seed=0, f16=True)
    fg_frac = float((labels != 0).float().mean().item())
    del labels
    ov = torch.zeros(shape, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    times, voxels = [], 0
    for z0 in range(0, shape[0], T):
        z1 = min(shape[0], z0 + T)
        Pb = backend.make_params(shape, ps, cons_box=(z0, 0, 0, z1, shape[1], shape[2]), **kw)
        a_ev, b_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_ev.record(torch.cuda.current_stream())
        cons = backend.consensus(pred, ov, Pb)
        b_ev.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        times.append(a_ev.elapsed_time(b_ev))
        voxels += (z1 - z0) * shape[1] * shape[2]
        del cons
    del pred, ov
    torch.cuda.empty_cache()
    r = s1_roofline(times, voxels, C, kernel=backend.NOTES.get("s1_kernel", "consensus_v3_kernel"))
    r.update({"workload": "synth512_p9 (BASELINE configs[2]): one S1 pass over all base voxels",
              "volume": list(shape), "patchshape": list(ps), "foreground_fraction": fg_frac,
              "slab_thickness": T, "total_ms": float(np.sum(times)),
              "Mvoxels_per_s_S1_only": voxels / (float(np.sum(times)) * 1e-3) / 1e6})
    return r


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
    bench.py <same arguments>`` as a child process (this process has not imported torch nor
    touched the GPU; nothing is exec'ed over it), pass its output through and return its exit
    code.  The child's ranks check WORLD_SIZE == --gpus and the visible device count."""
    import socket
    import subprocess
    port = os.environ.get("PPP_BENCH_MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: launching %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr)
    return subprocess.call(cmd, env=env)


def set_host_allocator():
    """backend.tune_host_allocator (the package's drivers call the same function); reported in
    the JSON."""
    from patchperpix_amd import backend
    return backend.tune_host_allocator(cli=True)


class SynthProvider:
    """Prediction provider of the synthetic workloads: pred_box() GENERATES the float16 prediction
    of a box on the device (ppp_synth_pred_box; labels of the box + patch radius evaluated with
    torch ops), bit-identical to generating the whole volume.  Stands where a reader of
    ``volumes/pred_affs`` chunks would stand for real data."""

    def __init__(self, torch, gshape, ps, cell, kw, seed=0):
        self.torch, self.gshape, self.ps, self.cell, self.kw, self.seed = torch, gshape, ps, cell, kw, seed
        self.bytes_max = 0

    def pred_box(self, box):
        from patchperpix_amd import backend
        z0, z1, y0, y1, x0, x1 = [int(v) for v in box]
        r = [p // 2 for p in self.ps]
        lb = (max(0, z0 - r[0]), min(self.gshape[0], z1 + r[0]), max(0, y0 - r[1]),
              min(self.gshape[1], y1 + r[1]), max(0, x0 - r[2]), min(self.gshape[2], x1 + r[2]))
        labels = device_labels(self.torch, (lb[1] - lb[0], lb[3] - lb[2], lb[5] - lb[4]), self.cell,
                               self.seed, z_offset=lb[0], y_offset=lb[2], x_offset=lb[4])
        pred = backend.synth_pred_box(labels, lb, box, self.gshape, self.ps, self.kw, seed=self.seed)
        self.bytes_max = max(self.bytes_max, pred.numel() * pred.element_size())
        return pred


class Workload:
    """One synthetic volume set up for timing: the float16 prediction and the per-voxel fields
    (foreground, numinst) resident in HBM, and ``step(flags)`` = one pass of the hot path."""

    def __init__(self, torch, name, kw, args, world, rank, comm=None):
        from patchperpix_amd import backend, tiling
        from patchperpix_amd.vote_instances import vote_instances as vi
        self.name = name
        shape, ps, cell = WORKLOADS[name]
        self.shape, self.ps, self.cell = shape, ps, cell
        self.kw = kw
        big = int(np.prod(shape)) >= 256 ** 3
        extra = {}
        if big:
            # large volumes carry uint32 ids like the reference's blockwise entry
            # (stitch_patch_graph.py:120): the shipped mws + includeSinglePatchCCS issue one id
            # per selected patch (454 029 at 512^3)
            extra["_instances_dtype"] = np.uint32
        self.ids = "uint32" if big else "uint16"
        self.mode = "resident"
        if name in PROVIDER_WORKLOADS:
            # ---- provider mode: nothing a rank holds is as large as the volume.  The rank's
            # z-range is cut into tiles whose consensus fits next to one generated tile.
            self.mode = "provider"
            gshape = shape
            self.gshape = gshape
            extra["_instances_dtype"] = np.uint32
            self.ids = "uint32"
            own = tiling.slabs_of_rank(tiling.plan_slabs(gshape[0], world), rank, world)
            if not own:
                raise SystemExit("more ranks than z-slabs")
            oz0, oz1 = own[0][0], own[-1][1]
            lo, hi = tiling.local_range(own, gshape[0], ps)
            fg = (device_labels(torch, (hi - lo, shape[1], shape[2]), cell, seed=0, z_offset=lo) != 0).to(torch.uint8)
            free = torch.cuda.mem_get_info()[0] + (torch.cuda.memory_reserved() - torch.cuda.memory_allocated())
            free = min(free, float(os.environ.get("PPP_BENCH_RANK_HBM_GB", "235")) * 1e9)   # per-rank footprint cap
            tile_pred = 2.0 * int(np.prod(ps)) * 180 ** 3                  # one generated tile + halo
            reserve = 150.0 * (oz1 - oz0 + 44) * shape[1] * shape[2] + tile_pred + 6e9
            n, ny, nx = tiling.tiles_needed((oz1 - oz0, shape[1], shape[2]), ps, max(free - reserve, 0.25 * free),
                                            safety=0.92, copies=2.0)
            mine = [(oz0 + a, oz0 + b) for a, b in tiling.plan_slabs(oz1 - oz0, args.slabs or n)]
            yx = tuple(args.yx) if args.yx else (ny, nx)
            self.provider = SynthProvider(torch, gshape, ps, cell, kw)
            self.tiles = (len(mine), yx[0], yx[1])
            self.own_range = (oz0, oz1)
            self.plan = {"rank": rank, "own_z": [oz0, oz1], "held_z": [lo, hi],
                         "z_slabs": [list(m) for m in mine], "yx_tiles": list(yx)}

            def step(flag_kw=kw):
                inst, _ = tiling.assemble(self.provider, lo, gshape, fg, fg.clone(), fg, ps, mine, comm=comm,
                                          _yx_tiles=yx, _gather_result=False, **dict(flag_kw, **extra))
                return inst
            self.pred = None
        elif name in DECODE_WORKLOADS:
            # ---- ppp+dec: codes resident (float16, like predict writes them), decoder weights
            # seeded; a step = decode into the float16 prediction block + logistic + vote
            from patchperpix_amd import decode as dec
            if world != 1:
                raise SystemExit("the decode workloads run on one GPU")
            self.mode = "decode"
            self.gshape = shape
            torch.manual_seed(0)
            dcfg = DECODER_2D if ps[0] == 1 else DECODER
            decoder = dec.PatchDecoder(dict(dcfg)).cuda().eval()
            fg = (device_labels(torch, shape, cell, seed=0) != 0).to(torch.uint8)
            g = torch.Generator(device="cuda").manual_seed(1)
            code = torch.randn((dcfg["code_units"],) + tuple(shape), generator=g, device="cuda",
                               dtype=torch.float16)
            # random weights: spread and centre the logits so that both classes occur
            with torch.no_grad():
                probe = decoder(code[:, :4].reshape(dcfg["code_units"], -1).t().float()[:4096])
                scale = 8.0 / float(probe.std())
                decoder.up_conv[-1][-1].weight.mul_(scale)
                decoder.up_conv[-1][-1].bias.mul_(scale).sub_(float(probe.median()) * scale)
                del probe
            self.pred = code
            fused = os.environ.get("PPP_DECODE_FUSED", "1") != "0"

            def step_2d(flag_kw=kw):
                # 2-d patches are for 2-d data: the slices are decoded and voted one image at a
                # time, as the reference processes 2-d samples (a stack with 2-d patches has no
                # defined result there: vote_instances.to_instance_seg refuses it)
                out = []
                for z in range(shape[0]):
                    fz = fg[z:z + 1]
                    with backend.host_timer("decode"):
                        pred = dec.decode_volume(decoder, code[:, z:z + 1], fz, batch_size=int(os.environ.get("PPP_DECODE_BATCH", "8192")),
                                                 out_dtype=torch.float16, fused=False)
                        flat = pred.reshape(pred.shape[0], -1)
                        idx = torch.nonzero(fz.reshape(-1)).reshape(-1)
                        flat[:, idx] = torch.sigmoid(flat[:, idx])
                    inst, _ = vi.to_instance_seg(pred, fz, fz.clone(), fz, ps, **dict(flag_kw, **extra))
                    out.append(np.asarray(inst))
                return np.concatenate(out, axis=0)

            def step(flag_kw=kw):
                if ps[0] == 1:
                    return step_2d(flag_kw)
                with backend.host_timer("decode"):
                    pred = dec.decode_volume(decoder, code, fg, batch_size=int(os.environ.get("PPP_DECODE_BATCH", "8192")),
                                             out_dtype=torch.float16, fused=None if fused else False)
                    flat = pred.reshape(pred.shape[0], -1)
                    idx = torch.nonzero(fg.reshape(-1)).reshape(-1)
                    for s0 in range(0, int(idx.numel()), 1 << 20):      # loadAffinities' expit on logits
                        sel = idx[s0:s0 + (1 << 20)]
                        flat[:, sel] = torch.sigmoid(flat[:, sel])
                inst, _ = vi.to_instance_seg(pred, fg, fg.clone(), fg, ps, **dict(flag_kw, **extra))
                return inst
        elif world == 1:
            self.gshape = shape
            P = backend.make_params(shape, ps, **kw)
            labels = device_labels(torch, shape, cell, seed=0)
            self.pred = backend.synth_pred(labels, P, seed=0, f16=True)       # resident in HBM
            fg = (labels != 0).to(torch.uint8)
            del labels
            if args.slabs:
                extra["_n_slabs"] = args.slabs
            if args.yx:
                extra["_yx_tiles"] = tuple(args.yx)
                extra.setdefault("_n_slabs", 1)        # (a forced y/x grid without --slabs: one z-slab)
            if args.slabs and os.environ.get("PPP_CONS_CACHE") == "1":
                extra["_cons_cache"] = True            # (a forced grid skips the memory plan)
            if args.slabs and os.environ.get("PPP_RING_Z"):
                extra["_ring_z"] = int(os.environ["PPP_RING_Z"])

            def step(flag_kw=kw):
                # foreground / mask / numinst are inputs like the prediction: resident in HBM
                # (the mask is the pipeline's scratch: a fresh copy per call, on the device)
                inst, _ = vi.to_instance_seg(self.pred, fg, fg.clone(), fg, ps, **dict(flag_kw, **extra))
                return inst
        else:
            # ONE volume split into z-slabs: rank r holds the prediction of its slab + halo only
            gshape = (shape[0] * world, shape[1], shape[2]) if args.scaling == "weak" else shape
            self.gshape = gshape
            slabs = tiling.plan_slabs(gshape[0], world)
            mine = tiling.slabs_of_rank(slabs, rank, world)
            if not mine:
                raise SystemExit("more ranks than z-slabs")
            lo, hi = tiling.local_range(mine, gshape[0], ps)
            # generate rz extra slices on both sides so that every channel of the kept range sees
            # its true neighbours, then keep [lo, hi)
            glo, ghi = max(0, lo - ps[0] // 2), min(gshape[0], hi + ps[0] // 2)
            eshape = (ghi - glo, shape[1], shape[2])
            Pe = backend.make_params(eshape, ps, **kw)
            labels_e = device_labels(torch, eshape, cell, seed=0, z_offset=glo)
            pred = backend.synth_pred(labels_e, Pe, seed=0, f16=True,
                                      voxel_offset=glo * shape[1] * shape[2])
            # PPP_BENCH_HALO=exchange (default): a rank PRODUCES its own slices only -- the U-Net's output
            # written into the middle of a halo-sized buffer -- and every step fetches the patch-radius
            # halo from its neighbours, in place, inside tiling.assemble (`_refresh_halo`:
            # exchange_halo, grouped point-to-point sends / receives over RCCL / xGMI; the north star's
            # wording).  The halo slices start out as zeros.  "generate": every rank generates its slab
            # with the halo (no prediction traffic; up to round 5)
            self.halo_mode = os.environ.get("PPP_BENCH_HALO", "exchange")
            self.pred = pred[:, lo - glo:hi - glo].contiguous()
            if self.halo_mode == "exchange":
                self.pred[:, :mine[0][0] - lo] = 0
                self.pred[:, mine[-1][1] - lo:] = 0
                extra["_refresh_halo"] = True
            del labels_e, pred
            # the fields of the rank's own slices only: the global stage runs sharded
            fg = (device_labels(torch, (hi - lo, shape[1], shape[2]), cell, seed=0, z_offset=lo) != 0).to(torch.uint8)
            # the rank's slab is cut into tiles whose consensus fits next to its prediction
            free = torch.cuda.mem_get_info()[0] + (torch.cuda.memory_reserved() - torch.cuda.memory_allocated())
            if os.environ.get("PPP_BENCH_ONE_GPU", "0") == "1":
                # development mode: the ranks share ONE device -- what is free once everybody's
                # prediction is resident, split evenly
                import torch.distributed as dist
                torch.cuda.synchronize()
                dist.barrier()
                free = 0.9 * torch.cuda.mem_get_info()[0] / world
                dist.barrier()
            oz0, oz1 = mine[0][0], mine[-1][1]
            reserve = 150.0 * (hi - lo) * shape[1] * shape[2] + 6e9
            # consensus cache: the compact planes of the rank's block (own slices + the pairs halo),
            # every base voxel computed once, when they fit next to the rows of one tile
            cz0, cz1 = max(0, oz0 - (ps[0] // 2) - (ps[0] - 1)), min(gshape[0], oz1 + ps[0] // 2)
            n, ny, nx, use_cache = tiling.plan_tiles((oz1 - oz0, shape[1], shape[2]), ps, max(free - reserve, 0.25 * free),
                                                     safety=0.92, copies=2.0, cache_shape=(cz1 - cz0, shape[1], shape[2]))
            ring_z = 0
            if not use_cache and not args.slabs and not args.yx:
                ring = tiling.plan_ring((oz1 - oz0, shape[1], shape[2]), ps, max(free - reserve, 0.25 * free),
                                        safety=0.92, copies=2.0)
                if ring is not None:
                    n, ny, nx, ring_z = ring
            mine = [(oz0 + a, oz0 + b) for a, b in tiling.plan_slabs(oz1 - oz0, args.slabs or n)]
            yx = tuple(args.yx) if args.yx else (ny, nx)
            self.plan = {"rank": rank, "own_z": [oz0, oz1], "held_z": [lo, hi],
                         "z_slabs": [list(m) for m in mine], "yx_tiles": list(yx), "cons_cache": bool(use_cache),
                         "ring_z": ring_z}
            self.tiles = (len(mine), yx[0], yx[1])
            self.plan["prediction_halo"] = self.halo_mode
            # the result stays where it is made: every rank returns its own z-range (the instance map
            # of 1024^3 as uint32 is 4.3 GB per rank and all-gather); the line's checksum is the
            # partition-independent crc of the per-slice crcs.  PPP_BENCH_GATHER=1: whole map everywhere
            self.own_range = (oz0, oz1)
            self.gather = os.environ.get("PPP_BENCH_GATHER", "0") == "1"
            extra.setdefault("_gather_result", self.gather)

            def step(flag_kw=kw):
                inst, _ = tiling.assemble(self.pred, lo, gshape, fg, fg.clone(), fg, ps, mine,
                                          comm=comm, _yx_tiles=yx, _cons_cache=use_cache, _ring_z=ring_z,
                                          **dict(flag_kw, **extra))
                return inst
        self.step = step
        self.fg_fraction = float(fg.float().mean().item())
        self.world = world

    def free(self, torch):
        self.pred = None
        self.step = None
        torch.cuda.empty_cache()


def rooflines(ev, notes, wl, ps, args, rank, world):
    """The three roofline blocks of the JSON line from the HIP-event times of the timed steps (`ev`) and the
    workload statistics (`notes`): S1 -- the contract's `roofline` (HBM-read fraction, counter traffic from the
    committed profile of these sources) and `roofline_valu` (the bound that holds) --, and the secondary
    figures of S2 and S5 (`roofline_other_kernels`)."""
    C = int(np.prod(ps))
    roofline = roofline_valu = None
    if ev.get("consensus"):
        s1_kernel = notes.get("s1_kernel", "consensus_v3_kernel")
        roofline = s1_roofline(ev["consensus"], notes.get("s1_base_voxels", 0), C, kernel=s1_kernel)
        roofline["workload"] = wl.name
        if rank == 0 and world == 1 and not args.slabs and not args.yx:
            roofline.update(pmc_traffic(s1_kernel, wl.name))
        if notes.get("s1_output_bytes"):
            # the algorithmic figure counts the INPUT; the launch also writes the whole
            # consensus -- most of the measured traffic
            roofline["output_bytes_per_launch"] = notes["s1_output_bytes"] / max(1, len(ev["consensus"]))
            roofline["traffic_reading"] = (
                "writes = the consensus output (symmetric voxel-major rows, 2x the stored planes); "
                "reads = the f16 prediction (re-read across the offset rows of a run) + the "
                "read-for-ownership of partially written lines")
        # "bound" keeps the contract's vocabulary and the north star's figure (HBM-read fraction);
        # what limits the kernel is vector-instruction issue: roofline_valu below
        roofline["limited_by"] = "valu-issue (see roofline_valu)"
        n_l = max(1, len(ev["consensus"]))
        fgf = float(getattr(wl, "fg_fraction", 1.0))
        roofline_valu = valu_roofline(s1_kernel, wl.name, float(np.sum(ev["consensus"])) / n_l,
                                      C * (C - 1) / 2.0 * fgf * notes.get("s1_base_voxels", 0) / n_l)
    # Secondary figures for the other two big kernels.  S2 (ranking): the voxel-major consensus
    # rows it reads once + the prediction block once: 4 (2p-1)^3 + 2 C bytes per base voxel.
    # S5 (patch graph): both patches' channel vectors per dispatched pair row, SURVEY 8(d)'s
    # bound without `visited`.
    roofline_other = {}
    if ev.get("rank_patches"):
        W_vm = (2 * ps[0] - 1) * (2 * ps[1] - 1) * (2 * ps[2] - 1)
        ms = float(np.sum(ev["rank_patches"]))
        b = (4.0 * W_vm + 2.0 * C) * notes.get("s2_base_voxels", notes.get("s1_base_voxels", 0))
        roofline_other["rank_patches"] = {
            "bound": "hbm", "achieved": b / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_ms": ms / len(ev["rank_patches"]),
            "launches": len(ev["rank_patches"]),
            "bytes": "voxel-major consensus rows of the launch's box once + the prediction block once",
            "algorithmic_bytes_per_launch": b / len(ev["rank_patches"])}
        if rank == 0 and world == 1 and not args.slabs and not args.yx:
            t = pmc_traffic("rank_wg_kernel", wl.name, read_width="float")
            if t.get("traffic") is not None:
                rd = t.get("traffic_read_corrected", t["traffic_read"])
                roofline_other["rank_patches"].update(
                    traffic_read=t["traffic_read"], traffic_read_corrected=t.get("traffic_read_corrected"),
                    counter_over_true_bytes=t.get("counter_over_true_bytes"), traffic_source=t["traffic_source"],
                    traffic_over_algorithmic=rd / (b / len(ev["rank_patches"])),
                    fabric_gb_per_s=rd / (ms / len(ev["rank_patches"]) * 1e-3) / 1e9)
    if ev.get("patch_graph") and notes.get("n_pairs"):
        ms = float(np.sum(ev["patch_graph"]))
        rows = float(notes.get("s5_rows_dispatched", notes["n_pairs"] * args.steps))
        b = rows * 2.0 * C * 2.0
        roofline_other["patch_graph"] = {
            "bound": "hbm", "achieved": b / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_ms": ms / len(ev["patch_graph"]),
            "launches": len(ev["patch_graph"]), "pair_rows_per_s": rows / (ms * 1e-3),
            "pair_rows_in_list_per_step": notes["n_pairs"]}
        if rank == 0 and world == 1 and not args.slabs and not args.yx:
            t = pmc_traffic("patch_graph_pa_kernel", wl.name, read_width="float")
            if t.get("traffic") is not None:
                rd = t.get("traffic_read_corrected", t["traffic_read"])
                roofline_other["patch_graph"].update(
                    traffic_read=t["traffic_read"], traffic_read_corrected=t.get("traffic_read_corrected"),
                    counter_over_true_bytes=t.get("counter_over_true_bytes"), traffic_source=t["traffic_source"],
                    fabric_gb_per_s=rd / (ms / len(ev["patch_graph"]) * 1e-3) / 1e9)
    return roofline, roofline_valu, roofline_other


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: %s (BASELINE configs[2]) with a fall-back to %s when the run would "
                         "not fit the time budget" % (DEFAULT_WORKLOAD, FALLBACK_WORKLOAD))
    ap.add_argument("--flags", default="shipped", choices=["shipped", "cc", "nothin_cc"],
                    help="flag set (patchperpix_amd/flags.py): shipped = default.toml "
                         "[vote_instances] (mws + thinning), cc = mws off, nothin_cc = "
                         "kernels-only pipeline")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N > 1: strong = the 1-GPU volume is split over the ranks, weak = the volume "
                         "grows with N (N x taller)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="time the CPU oracle and exit")
    ap.add_argument("--no-north-star", action="store_true",
                    help="flylight140_p7: skip the S1 pass over the 512^3 / 9^3 volume (half a minute)")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--slabs", type=int, default=None,
                    help="force the number of z-slabs of the single-GPU tiled path")
    ap.add_argument("--yx", type=int, nargs=2, default=None, metavar=("NY", "NX"),
                    help="cut every z-slab into NY x NX tiles (single-GPU tiled path)")
    ap.add_argument("--dry-run-plan", action="store_true",
                    help="no GPU needed: print what --gpus N ranks would hold and move for the workload "
                         "(z-ranges, halo, tile / ring / cache plan, bytes per collective) and exit")
    args = ap.parse_args()
    if args.dry_run_plan:
        from patchperpix_amd import tiling
        name = args.workload or DEFAULT_WORKLOAD
        shape, ps, _cell = WORKLOADS[name]
        plan = tiling.dry_run_plan(shape, ps, args.gpus, provider=name in PROVIDER_WORKLOADS,
                                   halo_mode=os.environ.get("PPP_BENCH_HALO", "exchange"),
                                   result_gather=os.environ.get("PPP_BENCH_GATHER", "0") == "1")
        print(json.dumps(dict(plan, workload=name)))
        return
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly (`python bench.py --gpus N`): start the N ranks ourselves, as a CHILD
        # launcher, before anything in this process touches torch or the GPU
        sys.exit(self_launch(args.gpus))

    allocator = set_host_allocator()
    import torch
    from patchperpix_amd import backend
    from patchperpix_amd import flags as flagsets

    kw = dict(flagsets.FLAG_SETS[args.flags])
    if args.cpu_baseline_only:
        shape, ps, cell = WORKLOADS[args.workload or DEFAULT_WORKLOAD]
        print(json.dumps(cpu_baseline(ps, cell, kw)))
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PPP_BENCH_ONE_GPU=1 (a check of the multi-rank path on a 1-GPU box): every rank on device 0,
    # gloo as the transport -- numbers from such a run mean nothing
    one_gpu = os.environ.get("PPP_BENCH_ONE_GPU", "0") == "1"
    # The line must describe the run that happened: the number of ranks IS --gpus, and every rank
    # has a device of its own (device_count() does not initialise the GPU).
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d -- start it as `python bench.py --gpus %d` (it "
                 "launches its ranks itself) or through torch.distributed.run with --nproc-per-node %d"
                 % (args.gpus, world, args.gpus, args.gpus))
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        sys.exit("bench.py needs a GPU (no CPU fallback)")
    if not one_gpu and n_dev < world:
        sys.exit("bench.py: --gpus %d but only %d device(s) visible (PPP_BENCH_ONE_GPU=1 lets the ranks "
                 "share device 0 for a functional check; its timings mean nothing)" % (world, n_dev))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(0 if one_gpu else local_rank)
    dist = comm = None
    collective_ranks = 1
    transport = "none (one rank)"
    if world > 1:
        import torch.distributed as dist
        from patchperpix_amd import tiling
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            # "nccl" = RCCL; bound to this rank's device (barriers and object collectives then use it)
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            except TypeError:
                dist.init_process_group("nccl")
        comm = tiling.TorchDistComm()
        # how many ranks the data path's collectives really span: a 1-element SUM all-reduce
        one = torch.ones(1, dtype=torch.int32, device="cpu" if one_gpu else "cuda")
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        collective_ranks = int(one.item())
        transport = "gloo through the host (PPP_BENCH_ONE_GPU=1: ranks share device 0)" if one_gpu \
            else "RCCL (torch.distributed backend nccl)"
        if collective_ranks != world:
            sys.exit("bench.py: all-reduce over the group counted %d ranks, expected %d" % (collective_ranks, world))
    n_gpus = world

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def agree_max(x):
        """the same number on every rank: the maximum"""
        if dist is None:
            return float(x)
        t = torch.tensor([float(x)], device="cpu" if one_gpu else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    budget_s = float(os.environ.get("PPP_BENCH_BUDGET_S", "1500"))
    auto = args.workload is None
    wl = Workload(torch, args.workload or DEFAULT_WORKLOAD, kw, args, world, rank, comm)
    barrier()
    warm_done = 0
    config2 = None
    if auto:
        # the first step of the default workload is timed (it also serves as the first warm-up
        # step): would W + K steps fit the budget?
        backend.NOTES.clear()
        backend.HOST_TIMES = {}
        t0 = time.perf_counter()
        inst = wl.step()
        barrier()
        t_first = agree_max(time.perf_counter() - t0)
        stages = {k: float(np.sum(v) * 1e3) for k, v in backend.HOST_TIMES.items()}
        backend.HOST_TIMES = None
        warm_done = 1
        planned = t_first * (args.steps + args.warmup)
        if planned > budget_s and planned < 1.3 * budget_s and args.warmup >= 2:
            # borderline: the first step also pays first-touch costs (the pooled buffers, library
            # handles); time a second warm-up step before giving the headline workload up
            t1 = time.perf_counter()
            inst = wl.step()
            barrier()
            t_second = agree_max(time.perf_counter() - t1)
            warm_done = 2
            planned = t_first + t_second * (args.steps + args.warmup - 1)
            t_first = min(t_first, t_second)
        if planned > budget_s:
            config2 = {
                "workload": wl.name, "volume": list(wl.gshape), "patchshape": list(wl.ps),
                "flag_set": args.flags, "ids": wl.ids, "steps": 1, "warmup": 0,
                "ms_per_step": t_first * 1e3, "value": float(np.prod(wl.gshape)) / t_first / 1e6,
                "unit": "Mvoxels/s", "instances_found": count_instances(torch, inst),
                "instances_crc32": int(zlib.crc32(np.ascontiguousarray(inst).tobytes())),
                "stage_wall_ms": stages, "workload_stats": dict(backend.NOTES),
                "why_not_timed": "%d steps of %.1f s = %.0f s exceed the budget of %.0f s "
                                 "(PPP_BENCH_BUDGET_S)" % (args.steps + args.warmup, t_first, planned, budget_s)}
            del inst
            wl.free(torch)
            wl = Workload(torch, FALLBACK_WORKLOAD, kw, args, world, rank, comm)
            warm_done = 0
            barrier()
    shape, ps, cell, gshape = wl.shape, wl.ps, wl.cell, wl.gshape
    step = wl.step

    for _ in range(max(0, args.warmup - warm_done)):
        inst = step()

    backend.EVENTS = {}
    backend.NOTES.clear()
    barrier()
    # The timed loop runs WITHOUT per-stage device syncs (round 6): the stage breakdown
    # (`stage_wall_ms`) comes from ONE extra step after it, with a sync around every stage.
    # PPP_BENCH_STAGES=inline: the syncs inside the timed loop, as up to round 5; =0: no breakdown.
    stages_mode = os.environ.get("PPP_BENCH_STAGES", "1")
    host_times = None
    if stages_mode == "inline":
        backend.HOST_TIMES = host_times = {}
    t0 = time.perf_counter()
    step_ends = []
    for _ in range(args.steps):
        inst = step()
        step_ends.append(time.perf_counter())     # (the step ends with a device -> host copy)
    barrier()
    dt = time.perf_counter() - t0
    step_ms = [1e3 * (b - a) for a, b in zip([t0] + step_ends[:-1], step_ends)]
    dt = agree_max(dt)
    ev = backend.event_times_ms()
    notes = dict(backend.NOTES)
    backend.EVENTS = None
    backend.HOST_TIMES = None
    stage_steps = args.steps
    staged_step_ms = None
    if stages_mode not in ("0", "inline"):
        backend.HOST_TIMES = host_times = {}
        keep_notes = dict(backend.NOTES)
        barrier()
        t1 = time.perf_counter()
        inst = step()
        barrier()
        staged_step_ms = agree_max(time.perf_counter() - t1) * 1e3
        backend.HOST_TIMES = None
        backend.NOTES.clear()
        backend.NOTES.update(keep_notes)
        stage_steps = 1

    C = int(np.prod(ps))
    # per-rank HBM footprint: the allocator's peak (prediction / tile, consensus pool, lists, work
    # space) -- the maximum over the ranks
    peak_gb = agree_max(torch.cuda.max_memory_allocated() / 1e9)
    value = float(np.prod(gshape)) * args.steps / dt / 1e6
    roofline, roofline_valu, roofline_other = rooflines(ev, notes, wl, ps, args, rank, world)
    # A checksum that does not depend on how the volume was split: crc32 of the per-slice crc32s
    # in z order.  With the result gathered every rank holds all slices; in provider mode a rank
    # returns its own z-range only and the per-slice values are gathered.
    arr = np.ascontiguousarray(inst)
    slice_crc = np.array([zlib.crc32(arr[z].tobytes()) for z in range(arr.shape[0])], dtype=np.int64)
    if dist is not None and getattr(wl, "own_range", None) is not None and arr.shape[0] != gshape[0]:
        sizes = [None] * world
        dist.all_gather_object(sizes, (int(wl.own_range[0]), slice_crc.tolist()))
        slice_crc = np.array([c for _, cs in sorted(sizes) for c in cs], dtype=np.int64)
    volume_crc = int(zlib.crc32(slice_crc.tobytes())) if len(slice_crc) == gshape[0] else None
    n_found = count_instances(torch, inst)
    own_only = dist is not None and arr.shape[0] != gshape[0]
    if own_only:
        # distinct ids over all ranks (ids are global; a rank's slab holds a few thousand)
        ids = torch.unique(torch.from_numpy(arr.astype(np.int64)).cuda()).cpu().tolist()
        all_ids = [None] * world
        dist.all_gather_object(all_ids, ids)
        n_found = len(set(i for part in all_ids for i in part if i != 0))
    # per-rank wall clock of every stage (the extra staged step): max / min over the ranks, so that
    # a scaling curve explains itself -- the replicated stages show as max = min = the 1-rank time
    stage_ranks = None
    if dist is not None and host_times is not None:
        mine_ms = {k: float(np.sum(v) / stage_steps * 1e3) for k, v in host_times.items()}
        every = [None] * world
        dist.all_gather_object(every, mine_ms)
        keys = sorted(set(k for d in every for k in d))
        stage_ranks = {k: {"max": round(max(d.get(k, 0.0) for d in every), 1),
                           "min": round(min(d.get(k, 0.0) for d in every), 1)} for k in keys}
    plan = getattr(wl, "plan", None)
    if plan is None and world == 1 and backend.LAST_PLAN is not None:
        # N = 1: the plan to_instance_seg made for the timed steps (tiles, ring, cache, free HBM)
        plan = dict(backend.LAST_PLAN, rank=0, ring_z_used=notes.get("ring_z", 0),
                    ring_z_scores_pass=notes.get("ring_z_scores", 0), rank_group=notes.get("rank_group", 0),
                    cons_cache_gb=notes.get("cons_cache_gb"),
                    # S2's tile of centres per workgroup, chosen per call by timing (1 = 8x8x16, 3 = 16x8x16;
                    # None: the library's rule) and the two trial launches' ns per centre
                    rank_tile=notes.get("rank_tile"), rank_tile_trial_ns_per_centre=notes.get("rank_tile_trial_ns_per_centre"))
    if dist is not None and plan is not None:
        plans = [None] * world
        dist.all_gather_object(plans, plan)
        plan = sorted(plans, key=lambda p: p["rank"])
    out = None
    if rank == 0:
        per_step = ("s1_base_voxels", "s2_base_voxels", "s5_rows_dispatched", "s1_output_bytes")
        out = {
            "metric": "Mvoxels/sec assembled (vote_instances)", "value": value,
            "unit": "Mvoxels/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            # (one volume whatever N: with the default --scaling strong the N = 1 run is the
            # first point of the same series)
            "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f16 in, f32 accumulate", "data": "synthetic",
            "config": {"workload": wl.name, "volume": list(shape), "patchshape": list(ps),
                       "pred_dtype": "f16 resident, widened to f32 in registers",
                       "flag_set": args.flags, "flags": flagsets.describe(kw),
                       "instance_ids": wl.ids, "foreground_fraction": wl.fg_fraction,
                       "instances_found": n_found,
                       "instances_crc32": int(zlib.crc32(np.ascontiguousarray(inst).tobytes())) if not own_only else None,
                       "instances_slice_crc32": volume_crc,
                       "global_volume": list(gshape),
                       "parallelism": ("z-ranges x%d ranks" % n_gpus) + (
                           ", %d x %d x %d tiles per rank" % wl.tiles if getattr(wl, "tiles", None) else
                           (", yx tiles %dx%d" % tuple(args.yx) if args.yx else "")),
                       "prediction": "resident in HBM" if wl.mode == "resident" else
                                     ("decoded inside the timed step from the resident float16 code (%d units per voxel; "
                                      "random-init decoder of the shipped architecture: %s)" % (
                                          DECODER["code_units"],
                                          "2-d, 25 x 25 patches, torch-ROCm convolutions (MIOpen) + index scatter" if ps[0] == 1 else
                                          "head as float32 library GEMMs, tail as the fused MFMA kernel")
                                      if wl.mode == "decode" else
                                      "generated tile by tile (provider): largest tile %.1f GB" % (wl.provider.bytes_max / 1e9)),
                       "result": "whole instance map on every rank" if not own_only else
                                 "own z-range per rank (instances_slice_crc32: crc of the per-slice crcs of all ranks; "
                                 "instances_found: distinct ids over all ranks)",
                       "per_rank_peak_hbm_gb": peak_gb,
                       "ranks": world, "rccl_ranks": collective_ranks if not one_gpu else None,
                       "collective_ranks": collective_ranks, "transport": transport,
                       "plan": plan,
                       "host_allocator": allocator},
            "roofline": roofline,
            "roofline_valu": roofline_valu,
            "roofline_other_kernels": roofline_other,
            "step_ms": [round(v, 1) for v in step_ms],
            **({"stage_lists_ms": {k: [1e3 * x for x in v] for k, v in (host_times or {}).items()}}
               if os.environ.get("PPP_BENCH_DUMP_STAGES") else {}),
            "kernel_ms": {k: float(np.sum(v) / args.steps) for k, v in ev.items()},
            "stage_wall_ms": {k: float(np.sum(v) / stage_steps * 1e3)
                              for k, v in (host_times or {}).items()},
            "stage_wall_ms_ranks": stage_ranks,
            "stage_wall_from": ("the timed steps (a device sync around every stage inside the timed loop)"
                                if stages_mode == "inline" else
                                "one extra step after the timed loop with a device sync around every stage "
                                "(%.0f ms; the timed steps run without those syncs)" % staged_step_ms
                                if staged_step_ms is not None else None),
            "workload_stats": {k: (v // args.steps if k in per_step else v) for k, v in notes.items()},
        }
        if config2 is not None:
            out["config2_end_to_end"] = config2
    if world == 1:
        small = wl.name == FALLBACK_WORKLOAD
        # ---- the other flag sets on the same volume (a few steps each; small volume only)
        if not args.no_variants and small:
            variants = {}
            for name in ("shipped", "cc", "nothin_cc"):
                if name == args.flags:
                    continue
                vkw = dict(flagsets.FLAG_SETS[name])
                step(vkw)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                n_v = 2
                for _ in range(n_v):
                    vinst = step(vkw)
                torch.cuda.synchronize()
                vdt = (time.perf_counter() - t1) / n_v
                variants[name] = {"ms_per_step": vdt * 1e3, "value": float(np.prod(gshape)) / vdt / 1e6,
                                  "instances_found": count_instances(torch, vinst),
                                  "differs_in": {k: vkw[k] for k in ("mws", "skipThinCover")}}
            out["variants"] = variants
        if os.environ.get("PPP_BENCH_CALIBRATE") == "1" and getattr(wl, "pred", None) is not None:
            # counter calibration (tools/profile_round.sh sets this for its --pmc passes): a known
            # byte count read from the resident prediction / written to a scratch buffer, outside
            # the timed region
            n_read = min(int(wl.pred.numel()), 1 << 31)
            scratch = torch.empty((1 << 28,), dtype=torch.float32, device="cuda")
            rb, wb = backend.counter_calibration(wl.pred.reshape(-1), n_read, scratch, scratch.numel())
            # the same bytes read back as floats (4 bytes per lane: the width of S2's / S5's loads)
            rb32, _ = backend.counter_calibration(scratch, scratch.numel(), scratch[:1], 0)
            torch.cuda.synchronize()
            out["counter_calibration_bytes"] = {"read": rb, "write": wb, "read_f32": rb32}
            del scratch
        wl.free(torch)
        # ---- the north-star shape of the scoring kernel, in this run (when the timed workload is
        # not that shape itself)
        if not args.no_north_star and small and config2 is None:
            out["roofline_north_star"] = north_star_s1(torch, backend, kw)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ps, cell, kw)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


VALU_CLOCK_GHZ = 2.4          # MI355X_MICROARCH.md: peak engine clock; a wave64 vector instruction issues in 4 cycles
N_SIMD = 256 * 4


def pmc_sq(kernel, workload):
    """SQ counters of `kernel` per launch from the committed passes of THIS workload and THESE kernel
    sources (profiles/*_pmc_sq.txt next to the same .meta.json as the FETCH / WRITE passes;
    tools/profile_round.sh): {counter: mean per launch} or {} when none matches."""
    import glob
    sha = source_sha16()
    found = None
    for meta_f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*.meta.json"))):
        try:
            meta = json.load(open(meta_f))
        except ValueError:
            continue
        f = meta_f.replace(".meta.json", "_pmc_sq.txt")
        if meta.get("workload", FALLBACK_WORKLOAD) == workload and meta.get("src_sha16") == sha and os.path.exists(f):
            found = f
    if found is None:
        return {}
    out = {"source": os.path.relpath(found, ROOT)}
    for ln in open(found):
        if kernel not in ln:
            continue
        parts = ln.split()
        for i, tok in enumerate(parts):
            if tok.startswith(("SQ_", "GRBM_")) and i + 1 < len(parts):
                try:
                    out[tok] = float(parts[i + 1])
                except ValueError:
                    pass
    return out


PK_F32_ISSUE_CYCLES = 4.7


def valu_roofline(kernel, workload, launch_ms, pair_votes_per_launch):
    """The S1 kernel against the bound that actually holds for it, vector-instruction issue.
    chain floor: the vote chain is 8 packed instructions per PAIR of votes (two z-slices per lane),
    64 lanes per instruction, one instruction per 4 cycles and SIMD, 1024 SIMDs; `instr_per_launch`
    (SQ_INSTS_VALU) and the busy cycles come from the committed SQ pass of these sources."""
    chain_instr = pair_votes_per_launch / 2.0 * 8.0 / 64.0
    floor_ms = chain_instr * 4.0 / (N_SIMD * VALU_CLOCK_GHZ * 1e9) * 1e3
    # measured issue cost of a packed float32 instruction at 4 waves per SIMD (tools/ubench/valu_rate2.hip,
    # profiles/r05_zf_issue_costs.txt: v_pk_fma / mul / add_f32 4.5-4.8 cycles per SIMD, v_fma_f32 2.9):
    # the floor with THAT cost instead of the nominal 4 cycles
    floor_measured_ms = floor_ms * PK_F32_ISSUE_CYCLES / 4.0
    out = {"kernel": kernel, "bound": "valu-issue", "launch_ms": launch_ms,
           "chain_instr_per_launch": chain_instr, "chain_floor_ms": floor_ms,
           "achieved_over_floor": floor_ms / launch_ms if launch_ms else None,
           "pk_f32_issue_cycles_measured": PK_F32_ISSUE_CYCLES, "chain_floor_measured_issue_ms": floor_measured_ms,
           "achieved_over_measured_floor": floor_measured_ms / launch_ms if launch_ms else None,
           "clock_ghz_assumed": VALU_CLOCK_GHZ,
           "chain": "x = ta*tb; dp = clamp(x*ga - 1/4); dn = clamp(-x - 1/4); d = dp - dn; q = d*lo(4/3); "
                    "y = fma(d, hi(4/3), q); acc += y; cnt = mad_u16(ca, cb, cnt)"}
    sq = pmc_sq(kernel, workload)
    if sq.get("SQ_INSTS_VALU"):
        instr = sq["SQ_INSTS_VALU"]
        out.update(instr_per_launch=instr, chain_share_of_instr=chain_instr / instr,
                   instr_floor_ms=instr * 4.0 / (N_SIMD * VALU_CLOCK_GHZ * 1e9) * 1e3,
                   issue_frac=instr * 4.0 / (N_SIMD * VALU_CLOCK_GHZ * 1e9) * 1e3 / launch_ms if launch_ms else None,
                   sq_source=sq["source"])
        if sq.get("SQ_BUSY_CYCLES"):
            # SQ_BUSY_CYCLES is summed over the 32 shader engines: cycles of the launch = / 32
            cyc = sq["SQ_BUSY_CYCLES"] / 32.0
            out.update(issue_frac_profiled_clock=instr * 4.0 / (N_SIMD * cyc),
                       clock_ghz_profiled=cyc / (launch_ms * 1e-3) / 1e9 if launch_ms else None)
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "SQ_INSTS_LDS",
                  "SQ_INSTS_SALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
            if k in sq:
                out[k] = sq[k]
    else:
        out["instr_per_launch"] = None
        out["sq_note"] = "no SQ pass of %s from these kernel sources is committed (tools/profile_round.sh)" % workload
    return out


def pmc_traffic(kernel, workload, read_width="pred"):
    """HBM traffic of `kernel` per launch from the committed rocprofv3 PMC passes of THIS workload
    (profiles/*_pmc_fetch_write.txt: FETCH_SIZE and WRITE_SIZE, separate --pmc passes, KiB,
    per-dispatch means) -- only from a profile taken from the kernel sources of THIS tree (its
    .meta.json holds their fingerprint and the workload); otherwise traffic stays null.  Among
    several matching profiles the last tag in name order is taken (not file times: they are
    arbitrary after a checkout).  Calibration (MI355X_MICROARCH.md: counters are exact for some
    access widths, halved for 16 B / lane streaming reads, uncalibrated otherwise): when the
    profile holds the transpose kernel, which moves a known byte count with this kernel's access
    width, its counter / true-bytes ratios are reported next to the raw values."""
    import glob
    sha = source_sha16()
    match, stale = [], 0
    for meta_f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*.meta.json"))):
        try:
            meta = json.load(open(meta_f))
        except ValueError:
            continue
        f = meta_f.replace(".meta.json", "_pmc_fetch_write.txt")
        if meta.get("workload", FALLBACK_WORKLOAD) != workload or not os.path.exists(f):
            continue
        if meta.get("src_sha16") == sha:
            match.append((f, meta))
        else:
            stale += 1
    if not match:
        return {"traffic": None,
                "traffic_note": "no PMC profile of %s from these kernel sources is committed (%d from "
                                "other sources): rerun tools/profile_round.sh" % (workload, stale)}
    f, meta = match[-1]
    vals, calib, known = {}, {}, {}
    for ln in open(f):
        parts = ln.split()
        for name in ("FETCH_SIZE", "WRITE_SIZE"):
            if name in parts:
                v = float(parts[parts.index(name) + 1]) * 1024.0
                if kernel in ln:
                    vals[name] = v
                if "cons_voxel_major_kernel" in ln:
                    calib[name] = v
                # read calibration in the kernel's own access width: calib_read_kernel<__half> (one
                # 2-byte prediction element per lane: S1's staging loads) or <float> (4 bytes per
                # lane: the row and mask loads of S2 / S5)
                want_read = "calib_read_kernel<float>" if read_width == "float" else "calib_read_kernel<__half>"
                if (want_read in ln and name == "FETCH_SIZE") or \
                        ("calib_write_kernel" in ln and name == "WRITE_SIZE"):
                    known[name] = v
    if len(vals) < 2:
        return {"traffic": None, "traffic_note": "kernel not in %s" % os.path.relpath(f, ROOT)}
    out = {"traffic": vals["FETCH_SIZE"] + vals["WRITE_SIZE"], "traffic_read": vals["FETCH_SIZE"],
           "traffic_write": vals["WRITE_SIZE"],
           "traffic_source": os.path.relpath(f, ROOT) + " (raw counters, mean per launch)"}
    read_bytes = meta.get("calib_read_f32_bytes") if read_width == "float" else meta.get("calib_read_bytes")
    if len(known) == 2 and read_bytes:
        # MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern"
        r = known["FETCH_SIZE"] / read_bytes
        w = known["WRITE_SIZE"] / meta["calib_write_bytes"]
        out["counter_over_true_bytes"] = {
            "read": r, "write": w,
            "from": "calib_read_kernel<%s> / calib_write_kernel of the same profile: %.3g GB read with one "
                    "%s per lane and load, %.3g GB written with one float per lane and store"
                    % ("float" if read_width == "float" else "__half", read_bytes / 1e9,
                       "float" if read_width == "float" else "prediction element", meta["calib_write_bytes"] / 1e9)}
        out["traffic_corrected"] = vals["FETCH_SIZE"] / r + vals["WRITE_SIZE"] / w
        out["traffic_read_corrected"] = vals["FETCH_SIZE"] / r
    elif len(calib) == 2 and meta.get("transpose_true_read_bytes"):
        out["counter_over_true_bytes"] = {
            "read": calib["FETCH_SIZE"] / meta["transpose_true_read_bytes"],
            "write": calib["WRITE_SIZE"] / meta["transpose_true_write_bytes"],
            "from": "cons_voxel_major_kernel of the same profile (4 B / lane coalesced loads and stores)"}
    return out


def cpu_baseline(ps, cell, kw):
    """The CPU oracle -- the restatement of the reference's kernels in C (gcc -O3, OpenMP) with the
    host stages in the package's host C++ -- on bounded samples of the same generator and flags:
    (i) one thread on a 32^3 sample, (ii) all host cores on a 64^3 sample."""
    from oracle import ppp_oracle as orc
    from patchperpix_amd import synth
    orc.lib()
    cores, cores_note = usable_cores()
    out = {"unit": "Mvoxels/s", "kind": "port",
           "what": "oracle/ppp_oracle.c (gcc -O3 -fopenmp; S1 in gather form over offset planes, "
                   "S2 over centres, S5 over pair rows) + host stages (ppp_host_* C++, one thread)"}
    runs = []
    # bounded samples (10-30 s each): the per-voxel work grows with C^2 (4.5x from 7^3 to 9^3)
    # (all cores: the 64^3 sample SURVEY 8(d) prescribes -- about half a minute of 16 cores at 9^3)
    samples = ((1, (24, 24, 24)), (cores, (64, 64, 64))) if int(np.prod(ps)) > 343 else \
        ((1, (32, 32, 32)), (cores, (64, 64, 64)))
    for threads, sshape in samples:
        sshape = tuple(min(s, 64) if p > 1 else 1 for s, p in zip(sshape, ps))
        if ps[0] == 1:
            sshape = (1, 96, 96) if threads == 1 else (1, 192, 192)
        lab = synth.cell_labels(sshape, cell, seed=0)
        pred = synth.pred_from_labels(lab, ps, seed=0)
        fg = lab != 0
        orc.set_threads(threads)
        CPU_STAGE_SECONDS.clear()
        t0 = time.perf_counter()
        inst = cpu_pipeline(orc, pred, fg, ps, kw)
        dt = time.perf_counter() - t0
        runs.append({"threads": threads, "sample": "x".join(map(str, sshape)), "seconds": dt,
                     "value": float(np.prod(sshape)) / dt / 1e6,
                     "instances_found": int(len(np.unique(inst)) - 1),
                     "stage_seconds": {k: round(v, 3) for k, v in CPU_STAGE_SECONDS.items()}})
    orc.set_threads(0)
    out["single_thread"], out["all_cores"] = runs
    # the headline of this object: all cores (the stronger baseline)
    out.update(value=runs[1]["value"], cores=runs[1]["threads"], cores_note=cores_note,
               sample="%s sub-volume of the same generator, %s patch, flags as timed, full pipeline"
                      % (runs[1]["sample"], "x".join(map(str, ps))))
    return out


CPU_STAGE_SECONDS = {}


def usable_cores():
    """Host cores this process may actually use: the affinity mask, cut down to the cgroup CPU
    quota when there is one (the GPU boxes expose 256 hardware threads under a quota of 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = "%d hardware threads in the affinity mask" % n
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            q = max(1, int(int(quota) / int(period)))
            if q < n:
                note += ", cgroup cpu.max quota = %d" % q
                n = q
    except (OSError, ValueError):
        pass
    return n, note


def cpu_pipeline(orc, pred, fg, ps, kw):
    """to_instance_seg on the CPU: the oracle's C loops for S1 / S2 / S5, the library's HOST
    functions (no device involved) for sort, cover, thinning, pairs and the mutex watershed,
    NumPy for the patch bits and the painting."""
    from patchperpix_amd import backend
    shape = fg.shape
    rad = [p // 2 for p in ps]
    numinst = fg.astype(np.uint8)
    ov = 1 * (numinst > 1)
    mask = fg.copy()
    t_stage = time.perf_counter()

    def lap(name):
        nonlocal t_stage
        now = time.perf_counter()
        CPU_STAGE_SECONDS[name] = CPU_STAGE_SECONDS.get(name, 0.0) + now - t_stage
        t_stage = now

    cons = orc.consensus_planes(pred, ov, ps, **kw)
    lap("s1_consensus")
    score = orc.rank(pred, cons, ov, ps, **kw)
    lap("s2_rank")
    lin = backend.host_rank_order(score, fg, ps)
    lap("sort")
    if len(lin) == 0:
        return np.zeros(shape, np.uint16)
    C = int(np.prod(ps))
    words = (C + 31) // 32

    def bits_of(lin_idx):
        vals = pred.reshape(C, -1)[:, lin_idx] > np.float32(kw["fc_threshold"])      # [C, n]
        out = np.zeros((len(lin_idx), words), dtype=np.uint32)
        for r in range(C):
            out[:, r // 32] |= vals[r].astype(np.uint32) << np.uint32(r % 32)
        return out

    bits = bits_of(lin)
    lap("patch_bits")
    running, _owner = backend.padded_mask(mask)
    selected = np.zeros(len(lin), dtype=np.uint8)
    radslice = tuple(slice(r, s - r) for r, s in zip(rad, shape))
    remaining = int(np.count_nonzero(running[radslice]))
    backend.host_cover_pass(running, np.ascontiguousarray(ov > 0).astype(np.uint8), ps, lin,
                            score.reshape(-1)[lin], bits, 0, None, selected, remaining)
    sel = np.flatnonzero(selected)
    lap("s3_cover")
    if not kw.get("skipThinCover") and len(sel):
        keep = backend.host_thin_cover(mask.astype(np.uint8), ps, lin[sel], bits[sel])
        sel = sel[keep]
    lap("s4_thin")
    sel_lin = lin[sel]
    coords = np.stack(np.unravel_index(sel_lin, shape), axis=1).astype(np.int32)
    _, pairs = backend.host_patch_pairs(coords, ps, kw.get("max_total_patch_distance_in_ps_multiples", 2),
                                        kw["includeSinglePatchCCS"])
    lap("pairs")
    if pairs is None:
        return np.zeros(shape, np.uint16)
    aff = orc.patch_graph(pred, cons, pairs, ps, **kw)
    lap("s5_patch_graph")
    if kw.get("mws"):
        nodes, labels, _ = backend.host_mws(pairs, aff, shape)
    else:
        # components of the positive edges (any component order: a baseline only counts them)
        from scipy.sparse import coo_matrix
        from scipy.sparse.csgraph import connected_components
        P64 = pairs.astype(np.int64)
        a_lin = (P64[:, 0] * shape[1] + P64[:, 1]) * shape[2] + P64[:, 2]
        b_lin = (P64[:, 3] * shape[1] + P64[:, 4]) * shape[2] + P64[:, 5]
        pos = aff > 0
        uniq, inv = np.unique(np.concatenate([a_lin[pos], b_lin[pos]]), return_inverse=True)
        k = int(pos.sum())
        g = coo_matrix((np.ones(k), (inv[:k], inv[k:])), shape=(len(uniq), len(uniq)))
        _, comp = connected_components(g, directed=False)
        nodes = np.stack(np.unravel_index(uniq, shape), axis=1)
        labels = comp + 1
    inst = np.zeros(shape, np.uint16)
    th = np.float32(kw["patch_threshold"])
    order = np.argsort(labels, kind="stable")          # later components overwrite earlier ones
    for c, lab in zip(np.asarray(nodes)[order], np.asarray(labels)[order]):
        win = tuple(slice(int(v) - r, int(v) + r + 1) for v, r in zip(c, rad))
        patch = pred[(slice(None),) + tuple(int(v) for v in c)].reshape(ps)
        inst[win][patch > th] = lab
    lap("s6_label_paint")
    return inst


if __name__ == "__main__":
    main()
