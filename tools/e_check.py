import sys, torch
sys.path.insert(0, '.')
import bench
from patchperpix_amd import backend, flags
shape, ps, cell = (160, 512, 512), (9, 9, 9), (24, 24, 24)
P = backend.make_params(shape, ps, **flags.FLYLIGHT)
lab = bench.device_labels(torch, shape, cell, seed=0)
pred = backend.synth_pred(lab, P, seed=0, f16=True)
n = 0
for c in range(0, pred.shape[0], 81):
    n += int((pred[c:c + 81] == 0.5).sum().item())
print("values == 0.5:", n, "of", pred.numel(), "min", float(pred.min()), "max", float(pred.max()))
