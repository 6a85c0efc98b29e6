"""Histogram of dispatched pair rows per patch in the per-patch S5 kernel, over one bench step.
Usage: python3 tools/s5_rows_hist.py [workload]"""
import ctypes
import io
import os
import sys
from contextlib import redirect_stdout

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wl = sys.argv[1] if len(sys.argv) > 1 else "synth256_p9"
sys.argv = ["bench.py", "--workload", wl, "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
import torch  # noqa: E402
import bench  # noqa: E402
from patchperpix_amd import backend  # noqa: E402

hist = torch.zeros(4096, dtype=torch.int64)
orig = backend.patch_graph_by_patch


def wrapped(pred, cons_vm, pairs, Pv):
    n = int(pairs.shape[0])
    if n:
        keys = torch.empty((n,), dtype=torch.int64, device=pairs.device)
        backend.check(backend.lib().ppp_pair_group_keys(backend._dev_ptr(pairs), n, backend._dev_ptr(keys),
                                                        ctypes.byref(Pv), backend._stream()))
        keys = keys[keys < backend.PAIR_KEY_FAR] >> 18
        _, counts = torch.unique_consecutive(torch.sort(keys)[0], return_counts=True)
        hist.add_(torch.bincount(counts.clamp(max=4095), minlength=4096).cpu())
    return orig(pred, cons_vm, pairs, Pv)


backend.patch_graph_by_patch = wrapped
with redirect_stdout(io.StringIO()):
    bench.main()
tot = int(hist.sum())
rows = int((hist * torch.arange(4096)).sum())
print("%s: %d patch groups, %d rows, %.1f rows per group" % (wl, tot, rows, rows / max(tot, 1)))
edges = [1, 17, 33, 49, 65, 73, 81, 97, 113, 129, 193, 257, 4096]
for a, b in zip(edges[:-1], edges[1:]):
    h = hist[a:b]
    r = int((h * torch.arange(a, b)).sum())
    print("  %4d..%4d rows: %5.1f %% of groups, %5.1f %% of rows" % (a, b - 1, 100.0 * int(h.sum()) / max(tot, 1),
                                                                   100.0 * r / max(rows, 1)))
waves = int((hist * ((torch.arange(4096) + 63) // 64)).sum())
print("  waves with work (64 rows each): %d = %.2f per group; rows beyond a group's first 64: %.1f %%" % (
    waves, waves / max(tot, 1), 100.0 * int((hist * (torch.arange(4096) - 64).clamp(min=0)).sum()) / max(rows, 1)))
