"""Stage-by-stage comparison HIP vs oracle for 2-d patches on a stack of slices (Z > 1)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ppp_oracle as orc
from patchperpix_amd import backend, synth
from patchperpix_amd.flags import FLYLIGHT
from patchperpix_amd.vote_instances import vote_instances as vi

p = int(sys.argv[1]) if len(sys.argv) > 1 else 9
Z = int(sys.argv[2]) if len(sys.argv) > 2 else 2
noise = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
shape, ps = (Z, 40, 44), [1, p, p]
kw = dict(FLYLIGHT, overlapping_inst=False)
case = synth.make_case(shape, ps, seed=3, cell=[1, 14, 14])
rng = np.random.default_rng(0)
pred = case["pred"]
if noise:
    pred = np.clip(pred + rng.uniform(-noise, noise, size=pred.shape), 0, 1)
pred = pred.astype(np.float16).astype(np.float32)
fg = case["foreground"].astype(bool)
ref = orc.to_instance_seg(pred, fg, fg.copy(), fg.astype(np.uint8), ps, **kw)
P = backend.make_params(shape, ps, **kw)
pd = torch.from_numpy(pred).cuda()
cons = backend.consensus(pd, None, P)
print("S1 kernel", backend.lib().ppp_consensus_kernel_name().decode())
bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
print("cons equal", np.array_equal(bits(cons.cpu().numpy()), bits(orc.positive_planes(ref["cons"], ps))))
score = backend.rank_patches(pd, cons, None, P).cpu().numpy()
print("score equal", np.array_equal(bits(score), bits(ref["scores"])), np.abs(score - ref["scores"]).max())
for pipe in ("fused", "stages"):
    os.environ["PPP_PIPELINE"] = pipe
    inst, _ = vi.to_instance_seg(pred.copy(), fg.copy(), fg.copy(), fg.astype(np.uint8), ps, **kw)
    print(pipe, "instances equal", np.array_equal(inst, ref["instances"]), len(np.unique(inst)), len(np.unique(ref["instances"])))
    res = vi.to_instance_seg(pred.copy(), fg.copy(), fg.copy(), fg.astype(np.uint8), ps, **dict(kw, return_intermediates=True))
    if res[0] is not None:
        print(pipe, "pairs equal", np.array_equal(res[0], ref["pairs"]), len(res[0]), len(ref["pairs"]),
              "aff equal", len(res[1]) == len(ref["aff"]) and np.array_equal(bits(res[1]), bits(ref["aff"])))
