"""When does every workgroup of a rank_wg_kernel launch start and end, and on which CU?  Needs the
diagnostic build of the library (PPP_EXTRA_FLAGS=-DPPP_RW_STAMPS, e.g. variants/libppp_rwstamps.so via
PPP_LIB): the kernel writes (start, end, HW_ID, tile) per workgroup into the tile-weight array of its
workspace.  Prints a summary: distribution of the workgroups' durations, of their start times, per XCD
and per round; round 6, "why does a launch of 1 024 workgroups take 131 ms and one of 2 048 only 196".

    PPP_LIB=$PWD/variants/libppp_rwstamps.so python tools/s2_wg_times.py [--case wg1024|wg2048|wg4096]"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = {"wg1024": ((24, 264, 264), (9, 9, 9), (24, 24, 24)), "wg2048": ((40, 264, 264), (9, 9, 9), (24, 24, 24)),
         "wg4096": ((72, 264, 264), (9, 9, 9), (24, 24, 24)),
         # the launch of the 512^3 step: 32 x 256 x 256 centres (2 048 tiles, none at a border of the rows)
         "ring2048": ((48, 272, 272), (9, 9, 9), (24, 24, 24), (8, 8, 8, 40, 264, 264))}
RW_ORDER_MAX = 16384


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="wg2048")
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from patchperpix_amd import backend, flags
    shape, ps, cell = CASES[args.case][:3]
    sbox = CASES[args.case][3] if len(CASES[args.case]) > 3 else None
    kw = dict(flags.FLYLIGHT)
    P = backend.make_params(shape, ps, **kw)
    labels = bench.device_labels(torch, shape, cell, seed=0)
    pred = backend.synth_pred(labels, P, seed=0, f16=True)
    ov = torch.zeros(shape, dtype=torch.uint8, device="cuda")
    vm, Pv = backend.consensus_voxel_major(pred, ov, P)
    L = backend.lib()
    box = ctypes.byref(backend.Box(*sbox)) if sbox else None
    nbytes = int(L.ppp_rank_workspace_bytes(box, ctypes.byref(Pv)))
    out = torch.zeros(shape, dtype=torch.float32, device="cuda")
    res = {"case": args.case, "lib": os.path.basename(backend.library_path())}
    for rep in range(2):
        work = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        backend.check(L.ppp_rank_patches_vm(backend._dev_ptr(pred), backend.pred_dtype_code(pred), backend._dev_ptr(vm),
                                            backend._dev_ptr(ov), backend._dev_ptr(out), box, backend._dev_ptr(work),
                                            ctypes.byref(Pv), backend._stream()))
        b.record()
        torch.cuda.synchronize()
        res["launch_ms"] = round(a.elapsed_time(b), 2)
    # where the tile-weight array sits in the workspace (rank_wg_workspace_bytes, ppp_rank_wg.hip): after the
    # masks (48 words per centre at 9^3, rows of 16 x-neighbours), the per-centre info, the validity bytes
    up = lambda v: (v + 255) // 256 * 256                                           # noqa: E731
    C = int(np.prod(ps))
    mw = (((C + 15) // 16) + 3) & ~3
    sb = sbox or ((0, 0, 0) + tuple(shape))
    sZ, sY, sX = sb[3] - sb[0], sb[4] - sb[1], sb[5] - sb[2]
    off = up(sZ * sY * ((sX + 15) // 16 * 16) * mw * 4) + up(sZ * sY * sX * 4) + up(int(np.prod(shape))) + 256
    st = work[off: off + RW_ORDER_MAX * 4].cpu().numpy().view(np.uint32).reshape(-1, 4)
    res["workgroups_launched"] = int(np.sum(st[:, 1] != 0))
    st = st[(st[:, 1] != 0)]
    if len(st) == 0:
        print(json.dumps(dict(res, error="no stamps: not the -DPPP_RW_STAMPS build")))
        return
    # s_memtime is a counter PER XCD, and the eight are not synchronised: cluster the workgroups by their
    # raw start (gaps far larger than a launch), take every cluster's own first start as its zero
    raw_s, raw_e = st[:, 0].astype(np.int64), st[:, 1].astype(np.int64)
    raw_e = np.where(raw_e < raw_s, raw_e + (1 << 32), raw_e)
    cluster = (st[:, 3] >> 24).astype(np.int64)                  # XCC_ID of the workgroup's XCD
    st[:, 3] &= 0xFFFFFF
    start = np.zeros(len(st)); end = np.zeros(len(st))
    for c in np.unique(cluster):
        m = cluster == c
        z = raw_s[m].min()
        start[m], end[m] = raw_s[m] - z, raw_e[m] - z
    res["xcd_clusters"] = int(len(np.unique(cluster)))
    # ticks (of 256 counts) -> ms: the longest cluster spans the launch
    scale = res["launch_ms"] / max(end.max(), 1e-9)
    start, end = start * scale, end * scale
    dur = end - start
    hw = st[:, 2]
    # HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (, xcc via XCC_ID elsewhere)
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5)
    res.update(workgroups=int(len(st)), clock_scale=round(float(scale), 4),
               duration_ms={k: round(float(v), 2) for k, v in zip(("min", "p10", "median", "p90", "max"),
                                                                 np.percentile(dur, [0, 10, 50, 90, 100]))},
               start_ms={k: round(float(v), 2) for k, v in zip(("p10", "median", "p90", "max"), np.percentile(start, [10, 50, 90, 100]))},
               first_round=int(np.sum(start < 1.0)),
               workgroups_per_xcd=[int(np.sum(cluster == c)) for c in np.unique(cluster)],
               last_end_per_xcd_ms=[round(float(end[cluster == c].max()), 1) for c in np.unique(cluster)])
    first = start < 1.0
    res["first_round_duration_ms"] = {k: round(float(v), 2) for k, v in zip(("min", "median", "max"), np.percentile(dur[first], [0, 50, 100]))}
    if (~first).any():
        res["later_rounds_duration_ms"] = {k: round(float(v), 2) for k, v in zip(("min", "median", "max"), np.percentile(dur[~first], [0, 50, 100]))}
    # how busy the slots are over time: resident workgroups in 10 slices of the launch
    edges = np.linspace(0, end.max(), 11)
    res["resident_workgroups_over_time"] = [int(np.sum((start < hi) & (end > lo))) for lo, hi in zip(edges[:-1], edges[1:])]
    # per CU id (within its shader engine): spread of the total busy time
    busy = {}
    for c, d in zip(cu.tolist(), dur.tolist()):
        busy[c] = busy.get(c, 0.0) + d
    v = np.array(list(busy.values()))
    res["per_cu_id_busy_ms"] = {"ids": len(busy), "min": round(float(v.min()), 1), "median": round(float(np.median(v)), 1), "max": round(float(v.max()), 1)}
    # duration against the tile's position in z (the tiles of a launch differ in where they sit)
    tiles = st[:, 3].astype(np.int64)
    tz = tiles // (tiles.max() // max(1, -(-sZ // 8)) + 1)
    res["median_duration_by_z_tile"] = {int(k): round(float(np.median(dur[tz == k])), 2) for k in np.unique(tz)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
