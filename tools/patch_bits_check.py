"""Dense (per-voxel pass + gather) vs per-centre patch bits on a large volume (development aid).
usage: patch_bits_check.py Z Y X p"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from patchperpix_amd import backend
from patchperpix_amd.flags import FLYLIGHT_NOTHIN_CC as FLYLIGHT
a = sys.argv[1:]
shape = tuple(int(v) for v in a[0:3]); p = int(a[3]); ps = (p, p, p)
cell = (24, 24, 24) if p == 9 else (18, 18, 18)
P = backend.make_params(shape, ps, **FLYLIGHT)
labels = bench.device_labels(torch, shape, cell, seed=0)
pred = backend.synth_pred(labels, P, seed=0, f16=True)
fg = labels != 0
r = p // 2
inner = torch.zeros_like(fg); inner[r:shape[0]-r, r:shape[1]-r, r:shape[2]-r] = True
lin = torch.nonzero((fg & inner).reshape(-1)).reshape(-1)
Y, X = shape[1], shape[2]
centres = torch.stack([lin // (Y * X), (lin // X) % Y, lin % X], 1).to(torch.int32).contiguous()
print("centres", centres.shape[0], flush=True)
dense = backend.patch_bits(pred, centres, 0.5, P)
os.environ["PPP_PATCH_BITS"] = "sparse"
bad = 0
step = 1 << 24
for s in range(0, centres.shape[0], step):
    sp = backend.patch_bits(pred, centres[s:s + step].contiguous(), 0.5, P)
    bad += int((sp != dense[s:s + step]).sum().item())
print("mismatching words:", bad)
