"""Time the watershed's host loop (ppp_host_mws_sorted) on a synthetic edge list: a 3-d grid of
nodes, edges to the neighbours within `reach`, random order (the loop takes edges that are already
sorted by |aff|), a given share of repulsive edges -- the decoded-noise workload of dec256_p7 has
about half.   python tools/time_mws.py [--n 96] [--reach 2] [--repulsive 0.5]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=96)
    ap.add_argument("--reach", type=int, default=2)
    ap.add_argument("--repulsive", type=float, default=0.5)
    ap.add_argument("--keep", type=float, default=0.5, help="share of the candidate edges kept")
    args = ap.parse_args()
    from patchperpix_amd import backend
    rng = np.random.default_rng(0)
    n = args.n
    idx = np.arange(n ** 3, dtype=np.int64).reshape(n, n, n)
    eu, ev = [], []
    r = args.reach
    for dz in range(0, r + 1):
        for dy in range(-r, r + 1):
            for dx in range(-r, r + 1):
                if (dz, dy, dx) <= (0, 0, 0):
                    continue
                a = idx[:n - dz, max(0, -dy):n - max(0, dy), max(0, -dx):n - max(0, dx)]
                b = idx[dz:, max(0, dy):n - max(0, -dy), max(0, dx):n - max(0, -dx)]
                keep = rng.random(a.shape) < args.keep
                eu.append(a[keep]); ev.append(b[keep])
    eu = np.concatenate(eu).astype(np.int32); ev = np.concatenate(ev).astype(np.int32)
    order = rng.permutation(len(eu))
    eu, ev = eu[order], ev[order]
    attractive = rng.random(len(eu)) >= args.repulsive
    ev = np.where(attractive, ev | np.int32(-2 ** 31), ev).astype(np.int32)
    labels = np.zeros(n ** 3, dtype=np.int32)
    L = backend.lib()
    t0 = time.perf_counter()
    issued = int(L.ppp_host_mws_sorted(backend._np_ptr(eu), backend._np_ptr(ev), len(eu), n ** 3, backend._np_ptr(labels)))
    dt = time.perf_counter() - t0
    print("nodes %d edges %d repulsive %.2f: %.2f s (%.0f ns / edge), ids issued %d, labels checksum %d"
          % (n ** 3, len(eu), args.repulsive, dt, dt / len(eu) * 1e9, issued, int(labels.astype(np.int64).sum() % (1 << 31))))


if __name__ == "__main__":
    main()
