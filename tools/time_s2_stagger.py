"""S2 (rank_wg_kernel): is the FIRST round of workgroups slow because its workgroups run in lockstep?
A launch of 1 024 workgroups takes 131 ms, one of 2 048 takes 196 ms (round 6, profiles/r06_b_*): the
second round's workgroups start one by one as the first round's finish.  Experiment: the 2 048-tile
launch as ONE launch, as two 1 024-tile launches on two streams started together, and with the second
stream delayed by d ms (torch.cuda._sleep) -- same scores every time.

    python tools/time_s2_stagger.py [--case wg2048] [--parts 2|4] [--delays 0 10 20 40 65]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = {"wg2048": ((40, 264, 264), (9, 9, 9), (24, 24, 24)), "wg4096": ((72, 264, 264), (9, 9, 9), (24, 24, 24))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="wg2048")
    ap.add_argument("--parts", type=int, default=2)
    ap.add_argument("--delays", type=float, nargs="*", default=[0, 10, 20, 40, 65])
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
    main_s = torch.cuda.current_stream()
    sides = [torch.cuda.Stream() for _ in range(args.parts - 1)]
    # z-ranges of centres, cut at multiples of the kernel's 8-slice tiles from the first interior slice
    z0, z1 = ps[0] // 2, shape[0] - ps[0] // 2
    n_t = (z1 - z0) // 8
    cuts = [z0 + 8 * (n_t * k // args.parts) for k in range(args.parts)] + [shape[0]]
    cuts[0] = 0
    boxes = [(cuts[k], 0, 0, cuts[k + 1], shape[1], shape[2]) for k in range(args.parts)]
    # cycles per ms of torch.cuda._sleep (it spins on the device clock)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); torch.cuda._sleep(100_000_000); b.record(); torch.cuda.synchronize()
    cyc_per_ms = 100_000_000 / a.elapsed_time(b)

    def timed(fn):
        best = None
        for _ in range(args.reps):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(main_s)
            out = fn()
            b.record(main_s)
            torch.cuda.synchronize()
            t = a.elapsed_time(b)
            best = t if best is None else min(best, t)
        return round(best, 2), out

    def whole():
        return backend.rank_patches(pred, vm, ov, Pv)

    def split(delay_ms):
        def run():
            out = torch.zeros(shape, dtype=torch.float32, device="cuda")
            for s in sides:
                s.wait_stream(main_s)
            for k, box in enumerate(boxes):
                if k == 0:
                    backend.rank_patches(pred, vm, ov, Pv, score_box=box, out=out)
                else:
                    with torch.cuda.stream(sides[k - 1]):
                        if delay_ms > 0:
                            torch.cuda._sleep(int(k * delay_ms * cyc_per_ms))
                        backend.rank_patches(pred, vm, ov, Pv, score_box=box, out=out)
            for s in sides:
                main_s.wait_stream(s)
            return out
        return run

    whole(); split(0)(); torch.cuda.synchronize()
    t_whole, sc = timed(whole)
    crc = lambda t: int(t.view(torch.int32).sum(dtype=torch.int64).item()) & 0xFFFFFFFF    # noqa: E731
    res = {"case": args.case, "parts": args.parts, "z_cuts": cuts, "one_launch_ms": t_whole, "checksum": crc(sc), "split_ms": {}}
    t_first, _ = timed(lambda: backend.rank_patches(pred, vm, ov, Pv, score_box=boxes[0]))
    res["first_part_alone_ms"] = t_first
    for d in args.delays:
        t, out = timed(split(d))
        res["split_ms"]["%g" % d] = t
        assert crc(out) == res["checksum"], "scores differ"
    print(json.dumps(res))


if __name__ == "__main__":
    main()
