#!/usr/bin/env python3
"""S1 (scoring / consensus kernel) roofline run at a named shape, e.g. BASELINE config [2]:
512^3 volume, 9^3 patch, dense foreground, prediction resident in HBM as float16.

The consensus of such a volume (2456 planes x 134 M voxels x 4 B = 1.3 TB) cannot be
materialised, so the kernel is launched slab by slab (base voxels [z0, z1) x Y x X) into one
reused buffer; every launch reads the prediction of its slab + halo once.  Prints one JSON line
with the algorithmic HBM-read rate ((2C + 1) bytes per base voxel) and the vote rate.

usage: run_s1_roofline.py [Z Y X] [p] [slab_thickness]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from patchperpix_amd import backend
from patchperpix_amd.flags import FLYLIGHT_NOTHIN_CC as FLYLIGHT

a = sys.argv[1:]
shape = tuple(int(v) for v in a[0:3]) if len(a) >= 3 else (512, 512, 512)
p = int(a[3]) if len(a) > 3 else 9
T = int(a[4]) if len(a) > 4 else 16
ps = (p, p, p)
C = p ** 3
cell = (24, 24, 24) if p == 9 else (18, 18, 18)
free, total = torch.cuda.mem_get_info()
planes = ((2 * p - 1) ** 3 - 1) // 2
need = 2.0 * C * np.prod(shape) + 4.0 * np.prod(shape) + planes * 4.0 * T * shape[1] * shape[2]
print("free %.1f GB, need %.1f GB" % (free / 1e9, need / 1e9), flush=True)
if need > 0.92 * free:
    sys.exit("not enough free HBM for this shape")
P = backend.make_params(shape, ps, **FLYLIGHT)
labels = bench.device_labels(torch, shape, cell, seed=0)
pred = backend.synth_pred(labels, P, seed=0, f16=True)
fg_frac = float((labels != 0).float().mean().item())
del labels
ov = torch.zeros(shape, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
times, voxels = [], 0
cons = None
for z0 in range(0, shape[0], T):
    z1 = min(shape[0], z0 + T)
    Pb = backend.make_params(shape, ps, cons_box=(z0, 0, 0, z1, shape[1], shape[2]), **FLYLIGHT)
    a_ev, b_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a_ev.record()
    cons = backend.consensus(pred, ov, Pb)
    b_ev.record()
    torch.cuda.synchronize()
    times.append(a_ev.elapsed_time(b_ev))
    voxels += (z1 - z0) * shape[1] * shape[2]
    del cons
total_ms = float(np.sum(times))
alg = (2.0 * C + 1.0) * voxels
print(json.dumps({
    "kernel": "consensus_v2_kernel", "volume": list(shape), "patchshape": list(ps),
    "foreground_fraction": fg_frac, "launches": len(times), "slab_thickness": T,
    "total_ms": total_ms, "avg_launch_ms": total_ms / len(times),
    "algorithmic_bytes": alg, "achieved_GBps": alg / (total_ms * 1e-3) / 1e9,
    "frac_of_8TBps": alg / (total_ms * 1e-3) / 1e9 / 8000.0,
    "Mvoxels_per_s_S1_only": voxels / (total_ms * 1e-3) / 1e6,
    "pair_votes_per_s_upper": C * (C - 1) / 2.0 * voxels / (total_ms * 1e-3)}))
