"""Time the scoring kernel S1 alone (kernel experiments; not the benchmark).

    python tools/time_s1.py [--case 140p7|slab9|...] [--reps N]

Cases: 140p7 = the whole 140^3 / 7^3 benchmark volume; slab9 = 16 slices of base voxels of a
(48, 512, 512) / 9^3 volume (the slab shape of the north-star pass); slab7 likewise at 7^3.
PPP_LIB / PPP_S1_* select library and kernel variants.  Prints one JSON line."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {
    "140p7": ((140, 140, 140), (7, 7, 7), None, (20, 20, 20)),
    "slab9": ((48, 512, 512), (9, 9, 9), (16, 0, 0, 32, 512, 512), (24, 24, 24)),
    "slab7": ((44, 512, 512), (7, 7, 7), (14, 0, 0, 30, 512, 512), (20, 20, 20)),
    "96p5": ((96, 96, 96), (5, 5, 5), None, (12, 12, 12)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="140p7")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import bench
    from patchperpix_amd import backend, flags
    shape, ps, box, cell = CASES[args.case]
    kw = dict(flags.FLYLIGHT)
    P = backend.make_params(shape, ps, **kw)
    labels = bench.device_labels(torch, shape, cell, seed=0)
    pred = backend.synth_pred(labels, P, seed=0, f16=True)
    ov = torch.zeros(shape, dtype=torch.uint8, device="cuda")
    Pb = backend.make_params(shape, ps, cons_box=box, **kw)
    times = []
    crc = None
    for r in range(args.reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(torch.cuda.current_stream())
        cons = backend.consensus(pred, ov, Pb)
        b.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        if r:
            times.append(a.elapsed_time(b))
        else:
            crc = int(cons.view(torch.int32).sum(dtype=torch.int64).item()) & 0xFFFFFFFF
        del cons
    print(json.dumps({"case": args.case, "lib": os.path.basename(backend.library_path()),
                      "env": {k: v for k, v in os.environ.items() if k.startswith("PPP_S1")},
                      "ms": [round(t, 2) for t in times], "min_ms": round(min(times), 2),
                      "checksum": crc}))


if __name__ == "__main__":
    main()
