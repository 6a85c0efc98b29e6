"""Does S1 (consensus_v3_kernel) of one tile run BESIDE S2 (rank_wg_kernel) of another?  Round 6
experiment: the two kernels of the scores pass on two HIP streams (S1 filling a second row buffer
while S2 ranks from the first) against the same two launches back to back on one stream.

    python tools/time_overlap.py [--case wg2048] [--reps N]
Prints one JSON line: ms of S1 alone, S2 alone, both in sequence, both on two streams."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = {"wg1024": ((24, 264, 264), (9, 9, 9), (24, 24, 24)), "wg2048": ((40, 264, 264), (9, 9, 9), (24, 24, 24)),
         "wg4096": ((72, 264, 264), (9, 9, 9), (24, 24, 24)), "140p7": ((140, 140, 140), (7, 7, 7), (20, 20, 20))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="wg2048")
    ap.add_argument("--reps", type=int, default=3)
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
    pool2 = torch.empty_like(vm)
    score = torch.zeros(shape, dtype=torch.float32, device="cuda")
    main_s = torch.cuda.current_stream()
    side = torch.cuda.Stream()

    def s1():
        backend.consensus_voxel_major(pred, ov, P, out=pool2.reshape(-1))

    def s2():
        return backend.rank_patches(pred, vm, ov, Pv)

    def timed(fn):
        best = None
        for _ in range(args.reps):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(main_s)
            fn()
            b.record(main_s)
            torch.cuda.synchronize()
            t = a.elapsed_time(b)
            best = t if best is None else min(best, t)
        return round(best, 2)

    def both_seq():
        s1()
        s2()

    def both_par():
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            s2()
        s1()
        main_s.wait_stream(side)

    def both_par_rev():
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            s1()
        s2()
        main_s.wait_stream(side)

    s1(); s2(); torch.cuda.synchronize()      # warm-up (allocations of S2's masks)
    out = {"case": args.case, "shape": shape, "s1_ms": timed(s1), "s2_ms": timed(s2), "sequence_ms": timed(both_seq),
           "two_streams_ms": timed(both_par), "two_streams_s1_on_side_ms": timed(both_par_rev)}
    # same scores either way (the concurrent S1 writes another buffer)
    sc = s2()
    out["checksum"] = int(sc.view(torch.int32).sum(dtype=torch.int64).item()) & 0xFFFFFFFF
    print(json.dumps(out))


if __name__ == "__main__":
    main()
