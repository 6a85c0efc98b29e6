"""Time the ranking kernels S2 alone on the voxel-major consensus of a synthetic volume (kernel
experiments; not the benchmark).   python tools/time_s2.py [--case 140p7|96p9] [--reps N]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = {"140p7": ((140, 140, 140), (7, 7, 7), (20, 20, 20)), "96p9": ((96, 96, 96), (9, 9, 9), (24, 24, 24)),
         "128p9": ((128, 128, 128), (9, 9, 9), (24, 24, 24)), "176p9": ((112, 176, 176), (9, 9, 9), (24, 24, 24)),
         # one / two / four rounds of workgroups (1024 / 2048 / 4096 tiles of 8 x 8 x 16 centres)
         "wg1024": ((24, 264, 264), (9, 9, 9), (24, 24, 24)), "wg2048": ((40, 264, 264), (9, 9, 9), (24, 24, 24)),
         "wg4096": ((72, 264, 264), (9, 9, 9), (24, 24, 24))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="140p7")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--burst", type=int, default=1, help="launches between the two events of a repetition")
    args = ap.parse_args()
    import torch
    import bench
    from patchperpix_amd import backend, flags
    shape, ps, cell = CASES[args.case]
    kw = dict(flags.FLYLIGHT)
    P = backend.make_params(shape, ps, **kw)
    labels = bench.device_labels(torch, shape, cell, seed=0)
    pred = backend.synth_pred(labels, P, seed=0, f16=True)
    ov = torch.zeros(shape, dtype=torch.uint8, device="cuda")
    vm, Pv = backend.consensus_voxel_major(pred, ov, P)
    times, crc = [], None
    for r in range(args.reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(torch.cuda.current_stream())
        for _ in range(args.burst):
            sc = backend.rank_patches(pred, vm, ov, Pv)
        b.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        if r:
            times.append(a.elapsed_time(b))
        else:
            crc = int(sc.view(torch.int32).sum(dtype=torch.int64).item()) & 0xFFFFFFFF
    print(json.dumps({"case": args.case, "lib": os.path.basename(backend.library_path()),
                      "burst": args.burst,
                      "ms": [round(t, 2) for t in times], "min_ms": round(min(times), 2), "checksum": crc}))


if __name__ == "__main__":
    main()
