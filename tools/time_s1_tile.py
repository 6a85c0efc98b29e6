"""Time S1 on ONE tile box of a large resident volume (the shape the 512^3 step launches), with the
library of the tree named by PPP_TREE (default: this one) -- kernel A/B across source trees.

    python tools/time_s1_tile.py [--shape Z Y X] [--box z0 y0 x0 z1 y1 x1] [--reps N]
Prints one JSON line (min ms of the voxel-major launch, checksum of the rows)."""
import argparse
import json
import os
import sys

ROOT = os.environ.get("PPP_TREE") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", type=int, nargs=3, default=[512, 512, 512])
    ap.add_argument("--box", type=int, nargs=6, default=[200, 167, 167, 311, 346, 346])
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import bench
    from patchperpix_amd import backend, flags
    shape, ps, cell = tuple(args.shape), (9, 9, 9), (24, 24, 24)
    kw = dict(flags.FLYLIGHT)
    P = backend.make_params(shape, ps, **kw)
    labels = bench.device_labels(torch, shape, cell, seed=0)
    pred = backend.synth_pred(labels, P, seed=0, f16=True)
    del labels
    ov = torch.zeros(shape, dtype=torch.uint8, device="cuda")
    Pb = backend.make_params(shape, ps, cons_box=tuple(args.box), **kw)
    Pb = backend.with_pred_clean(pred, Pb)        # (decided once, outside the timed launches)
    nvox = (args.box[3] - args.box[0]) * (args.box[4] - args.box[1]) * (args.box[5] - args.box[2])
    W = 17 ** 3
    pool = torch.empty(nvox * W, dtype=torch.float32, device="cuda")
    times, crc = [], None
    for r in range(args.reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(torch.cuda.current_stream())
        rows, _ = backend.consensus_voxel_major(pred, ov, Pb, out=pool, open_rows=True)
        b.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        if r:
            times.append(a.elapsed_time(b))
        else:
            inner = rows[40:42, 8:-8, 8:-8]       # (two slices; entries of open rows near the faces are undefined)
            crc = int(inner.reshape(-1).view(torch.int32).sum(dtype=torch.int64).item()) & 0xFFFFFFFF
    print(json.dumps({"tree": os.path.basename(ROOT), "shape": shape, "box": args.box, "base_voxels": nvox,
                      "ms": [round(t, 2) for t in times], "min_ms": round(min(times), 2),
                      "Mvox_per_s": round(nvox / min(times) / 1e3, 2), "checksum": crc, "pred_clean": int(Pb.pred_clean),
                      "kernel": backend.lib().ppp_consensus_kernel_name().decode()}))


if __name__ == "__main__":
    main()
