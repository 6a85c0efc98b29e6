import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from patchperpix_amd import backend
from patchperpix_amd.flags import FLYLIGHT_NOTHIN_CC as FLYLIGHT
from patchperpix_amd.vote_instances import foreground_cover as fc
shape, ps, cell = bench.WORKLOADS["flylight140_p7"]
kw = dict(FLYLIGHT)
P = backend.make_params(shape, ps, **kw)
labels = bench.device_labels(torch, shape, cell, seed=0)
pred = backend.synth_pred(labels, P, seed=0, f16=True)
fg = (labels != 0).cpu().numpy()
ov = torch.zeros(shape, dtype=torch.uint8, device="cuda")
cons = backend.consensus(pred, ov, P)
score = backend.rank_patches(pred, cons, ov, P)
del cons
lin, s = backend.rank_order_device(score, fg, ps)
coords = np.stack(np.unravel_index(lin, shape), axis=1).astype(np.int32)
print("candidates", len(lin))
running, _o = backend.padded_mask(fg)
rad = [3,3,3]
radslice = tuple(slice(3, shape[i]-3) for i in range(3))
remaining = int(np.count_nonzero(running[radslice]))
selected = np.zeros(len(lin), np.uint8)
ovh = np.zeros(shape, np.uint8)
CH = 1 << 20
for st in range(0, len(lin), CH):
    e = min(len(lin), st + CH)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bits = fc._bits_for(pred, coords[st:e], 0.5, P)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    remaining, _ = backend.host_cover_pass(running, ovh, ps, lin[st:e], s[st:e], bits, 0, None, selected[st:e], remaining)
    t2 = time.perf_counter()
    print("chunk", st, "bits %.1f ms" % ((t1-t0)*1e3), "host %.1f ms" % ((t2-t1)*1e3), "remaining", remaining, "selected", int(selected.sum()))
    if remaining <= 0: break
