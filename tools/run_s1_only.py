#!/usr/bin/env python3
"""Run only the S1 consensus kernel (and optionally S2) on a synthetic volume -- a small driver
for rocprofv3 counter collection.  usage: run_s1_only.py [workload] [reps] [s2]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from patchperpix_amd import backend
from patchperpix_amd.flags import FLYLIGHT_NOTHIN_CC as FLYLIGHT

name = sys.argv[1] if len(sys.argv) > 1 else "flylight140_p7"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
do_s2 = len(sys.argv) > 3
shape, ps, cell = bench.WORKLOADS[name]
P = backend.make_params(shape, ps, **FLYLIGHT)
labels = bench.device_labels(torch, shape, cell, seed=0)
pred = backend.synth_pred(labels, P, seed=0, f16=True)
ov = torch.zeros(shape, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for r in range(reps):
    t0 = time.perf_counter()
    cons = backend.consensus(pred, ov, P)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if do_s2:
        backend.rank_patches(pred, cons, ov, P)
        torch.cuda.synchronize()
    print("S1 %.1f ms  S2 %.1f ms" % ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
    del cons
