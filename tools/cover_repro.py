"""Run the pipeline up to the cover several times and print checksums of the scores, the ranked
list and the selected patches (development aid).  usage: cover_repro.py Z Y X p [repeats]"""
import os, sys
os.environ["PPP_DEBUG_CRC"] = "1"
os.environ["PPP_STOP_AFTER_COVER"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from patchperpix_amd import backend
from patchperpix_amd.flags import FLYLIGHT_NOTHIN_CC as FLYLIGHT
from patchperpix_amd.vote_instances import vote_instances as vi
a = sys.argv[1:]
shape = tuple(int(v) for v in a[0:3]); p = int(a[3]); ps = (p, p, p)
rep = int(a[4]) if len(a) > 4 else 2
cell = (24, 24, 24) if p == 9 else (18, 18, 18)
P = backend.make_params(shape, ps, **FLYLIGHT)
labels = bench.device_labels(torch, shape, cell, seed=0)
pred = backend.synth_pred(labels, P, seed=0, f16=True)
fg = (labels != 0).cpu().numpy()
del labels
for i in range(rep + 1):
    if i == rep:
        os.environ["PPP_PATCH_BITS"] = "sparse"
    backend.NOTES.clear()
    vi.to_instance_seg(pred, fg.copy(), fg.copy(), fg.astype(np.uint8), ps, **dict(FLYLIGHT))
    print("run", i, os.environ.get("PPP_PATCH_BITS", "auto"), dict(backend.NOTES), flush=True)
