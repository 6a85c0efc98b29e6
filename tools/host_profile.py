"""cProfile of the host side of the benchmark step (where the wall time outside the kernels goes).
   python tools/host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import bench
    from patchperpix_amd import backend, flags
    from patchperpix_amd.vote_instances import vote_instances as vi
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    shape, ps, cell = bench.WORKLOADS["flylight140_p7"]
    kw = dict(flags.FLYLIGHT)
    P = backend.make_params(shape, ps, **kw)
    labels = bench.device_labels(torch, shape, cell, seed=0)
    pred = backend.synth_pred(labels, P, seed=0, f16=True)
    fg = (labels != 0).cpu().numpy()
    numinst = fg.astype(np.uint8)

    def step():
        return vi.to_instance_seg(pred, fg.copy(), fg.copy(), numinst, ps, **kw)[0]
    step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
