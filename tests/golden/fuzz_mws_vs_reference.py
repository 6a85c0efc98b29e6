#!/usr/bin/env python3
"""Mutex watershed: the reference's graph_mws.mws on the graph its setAffgraph builds (both imported in place)
against the oracle's restatement and the library's native ppp_host_mws, on random patch graphs -- ties, zero
weights, repeated pairs, self pairs, all-repulsive and all-attractive graphs.  A component's id is its position
in the reference's output list + 1 (graph_to_labeling.py:73-84; emptied components keep their position).
Development container only.

  python tests/golden/fuzz_mws_vs_reference.py [--trials 400]
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    gg.install_stubs()
    gg.install_fake_cuda_code()
    sys.path.insert(0, gg.REF_VI)
    import logging
    logging.basicConfig(level=logging.ERROR)
    import aff_patch_graph as apg
    import graph_mws
    from oracle import ppp_oracle as orc
    from patchperpix_amd import backend
    rng = np.random.default_rng(args.seed)
    bad = 0
    for trial in range(args.trials):
        shape = tuple(int(rng.integers(2, 7)) for _ in range(3))
        n_nodes = int(rng.integers(2, 25))
        coords = np.unique(np.stack([rng.integers(0, s, size=n_nodes) for s in shape], axis=1), axis=0)
        n = int(rng.integers(1, 80))
        a_idx, b_idx = rng.integers(0, len(coords), size=n), rng.integers(0, len(coords), size=n)
        pairs = np.concatenate([coords[a_idx], coords[b_idx]], axis=1).astype(np.uint32)
        # node pairs never repeat in the library's own pair list (either orientation); keep the first
        seen, keep = set(), []
        for i, (u, v) in enumerate(zip(map(tuple, coords[a_idx]), map(tuple, coords[b_idx]))):
            k = (min(u, v), max(u, v))
            if k not in seen:
                seen.add(k)
                keep.append(i)
        pairs = np.ascontiguousarray(pairs[keep])
        kind = str(rng.choice(["mixed", "ties", "repulsive", "attractive"]))
        if kind == "ties":
            aff = rng.choice([-0.5, -0.25, 0.0, 0.25, 0.5], size=len(pairs)).astype(np.float32)
        else:
            aff = rng.uniform(-1, 1, size=len(pairs)).astype(np.float32)
            aff[rng.random(len(pairs)) < 0.1] = 0.0
            if kind == "repulsive":
                aff = -np.abs(aff)
            if kind == "attractive":
                aff = np.abs(aff)
        graph = apg.setAffgraph(aff, pairs)
        ref_ccs = graph_mws.mws(graph) if graph.number_of_nodes() else []
        want = {}
        for k, cc in enumerate(ref_ccs):
            for node in cc:
                want[tuple(int(v) for v in node)] = k + 1
        o_ccs = orc.mutex_watershed(pairs, aff)
        got_o = {tuple(int(v) for v in node): k + 1 for k, cc in enumerate(o_ccs) for node in cc}
        nodes, labels, n_labels = backend.host_mws(pairs, aff, shape)
        got_n = {tuple(int(v) for v in node): int(l) for node, l in zip(nodes, labels)}
        status = []
        if got_o != want:
            status.append("ORACLE")
        if got_n != want:
            status.append("NATIVE")
        if n_labels != len(ref_ccs):
            status.append("IDS ISSUED (%d vs %d)" % (n_labels, len(ref_ccs)))
        if status:
            bad += 1
            print("trial", trial, shape, kind, len(pairs), "DIFFER", status, flush=True)
    print("%d trials, %d failures" % (args.trials, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
