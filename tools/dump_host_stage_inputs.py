#!/usr/bin/env python3
"""Development aid: run one flylight140_p7 step with the shipped flags on the GPU and save what the
two host stages (set-cover thinning, mutex watershed) receive and return, so that they can be
profiled / re-implemented off the GPU box.  Writes gpurun_out/host_stage_inputs.npz."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from patchperpix_amd import backend, flags  # noqa: E402
from patchperpix_amd.vote_instances import vote_instances as vi  # noqa: E402

shape, ps, cell = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "flylight140_p7"]
kw = dict(flags.FLYLIGHT)
P = backend.make_params(shape, ps, **kw)
labels = bench.device_labels(torch, shape, cell, seed=0)
pred = backend.synth_pred(labels, P, seed=0, f16=True)
fg = (labels != 0).cpu().numpy()
saved = {}
orig_thin, orig_mws = backend.host_thin_cover, backend.host_mws


def thin(mask, patchshape, sel_lin, bits):
    t0 = time.perf_counter()
    keep = orig_thin(mask, patchshape, sel_lin, bits)
    saved.update(thin_mask=np.packbits(mask.astype(bool)), thin_sel_lin=sel_lin, thin_bits=bits,
                 thin_keep=keep, thin_seconds=time.perf_counter() - t0)
    return keep


def mws(pairs, aff, shp):
    t0 = time.perf_counter()
    out = orig_mws(pairs, aff, shp)
    saved.update(mws_pairs=np.asarray(pairs).astype(np.uint8), mws_aff=aff, mws_nodes=out[0],
                 mws_labels=out[1], mws_n_labels=out[2], mws_seconds=time.perf_counter() - t0)
    return out


backend.host_thin_cover, backend.host_mws = thin, mws
inst, _ = vi.to_instance_seg(pred, fg.copy(), fg.copy(), fg.astype(np.uint8), ps, **kw)
print("instances", len(np.unique(inst)) - 1, "thin s", saved.get("thin_seconds"), "mws s", saved.get("mws_seconds"))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "host_stage_inputs.npz"), shape=np.array(shape),
                    patchshape=np.array(ps), instances=inst, **saved)
