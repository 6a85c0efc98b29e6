"""Repeat the tiled pipeline and print the instance-map CRC / cover size of every run
(development aid).  usage: determinism_check.py Z Y X p  n_slabs ny nx  [repeats]"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from patchperpix_amd import backend
from patchperpix_amd.flags import FLYLIGHT_NOTHIN_CC as FLYLIGHT
from patchperpix_amd.vote_instances import vote_instances as vi
a = sys.argv[1:]
shape = tuple(int(v) for v in a[0:3]); p = int(a[3]); ps = (p, p, p)
ns, ny, nx = int(a[4]), int(a[5]), int(a[6])
rep = int(a[7]) if len(a) > 7 else 3
cell = (24, 24, 24) if p == 9 else (18, 18, 18)
P = backend.make_params(shape, ps, **FLYLIGHT)
labels = bench.device_labels(torch, shape, cell, seed=0)
pred = backend.synth_pred(labels, P, seed=0, f16=True)
fg = (labels != 0).cpu().numpy()
for cfg in [dict(_n_slabs=1)] + [dict(_n_slabs=ns, _yx_tiles=(ny, nx))] * rep:
    kw = dict(FLYLIGHT, **cfg)
    inst, _ = vi.to_instance_seg(pred, fg.copy(), fg.copy(), fg.astype(np.uint8), ps, **kw)
    print(cfg, "crc", zlib.crc32(np.ascontiguousarray(inst).tobytes()), dict(backend.NOTES), flush=True)
