#!/usr/bin/env python3
"""Random small volumes / block sizes / flags through the reference's BLOCKWISE driver (stitch_patch_graph.main,
imported in place as gen_golden_blockwise.py does) and through patchperpix_amd.blockwise served by the CPU oracle
(as tests/test_blockwise.py::test_blockwise_matches_reference_cpu does with the three committed goldens):
every stored block / inter-block group (pair rows, affinity bits), the set of groups, the written instance map.
Development container only.

  python tests/golden/fuzz_blockwise_vs_reference.py [--trials 12] [--seed 1]
"""
import argparse
import os
import shutil
import sys
import tempfile
import time
import traceback
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import gen_golden as gg  # noqa: E402
import gen_golden_blockwise as gb  # noqa: E402
from patchperpix_amd import blockwise, minizarr, synth  # noqa: E402


def stored(path):
    out = {}
    res = minizarr.open(path, "r")
    blocks = res["volumes/blocks"]

    def walk(g, prefix):
        for k in sorted(g.keys()):
            item = g[k]
            if hasattr(item, "keys"):
                walk(item, prefix + "/" + k)
            else:
                out[prefix + "/" + k] = np.asarray(item[...])
    walk(blocks, "")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=12)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    recorded = {}
    gg.install_stubs()
    gg.install_fake_cuda_code()
    gb.install_blockwise_stubs(recorded)
    sys.path.insert(0, gg.REF_VI)
    import logging
    logging.basicConfig(level=logging.ERROR)
    import aff_patch_graph as apg
    import stitch_patch_graph as spg
    tree = gb.sorted_kdtree()
    apg.scipy = types.SimpleNamespace(spatial=types.SimpleNamespace(cKDTree=tree))
    spg.spatial = types.SimpleNamespace(cKDTree=tree)
    import test_blockwise as tb
    from patchperpix_amd.vote_instances import vote_instances as vi
    rng = np.random.default_rng(args.seed)
    bad = 0
    t0 = time.time()
    for trial in range(args.trials):
        p = int(rng.choice([3, 3, 5]))
        ps = (p, p, p)
        shape = tuple(int(rng.integers(2 * p + 3, 2 * p + 12)) for _ in range(3))
        chunk = [int(rng.integers(p + 3, max(p + 4, s))) for s in shape]
        flags = dict(mws=bool(rng.integers(0, 2)), skipThinCover=bool(rng.integers(0, 2)),
                     includeSinglePatchCCS=bool(rng.integers(0, 3) != 0))
        seed = int(rng.integers(1, 100000))
        cell = [int(rng.integers(p, 2 * p + 3))] * 3
        desc = "trial %d shape %s p %d chunk %s %s seed %d cell %d" % (trial, shape, p, chunk, flags, seed, cell[0])
        work = tempfile.mkdtemp(prefix="ppp_bwf_")
        try:
            case = synth.make_case(shape, ps, seed=seed, cell=cell, overlap_frac=0.0)
            pred_file = os.path.join(work, "sample.zarr")
            pred16 = case["pred"].astype(np.float16)
            g = minizarr.open(pred_file, "w")
            g.create_dataset("volumes/pred_affs", data=pred16, chunks=(pred16.shape[0], 8, 8, 8))
            kw = dict(gg.FLYLIGHT)
            kw.update(gg.FIXED)
            kw.update(flags)
            kw.update(patchshape=list(ps), chunksize=list(chunk), aff_key="volumes/pred_affs", numinst_key=None,
                      fg_key=None, res_key="vote_instances", output_format="hdf", only_bb=False,
                      ignore_small_comps=0, blockwise=True, overlapping_inst=False, remove_small_comps=0, save_mip=False)
            kw.pop("result_folder", None)
            recorded.clear()
            t1 = time.time()
            ref_dir = os.path.join(work, "ref")
            os.makedirs(ref_dir)
            try:
                spg.main(pred_file, result_folder=ref_dir, **kw)
            except Exception as e:       # noqa: BLE001
                # (e.g. no block with a pair anywhere: stitch_patch_graph.py:369 reads affgraph.nodes of None)
                print(desc, "REFERENCE RAISED %r (not counted)" % (e,), flush=True)
                continue
            t_ref = time.time() - t1
            want_inst = recorded.get("vote_instances")
            a = stored(os.path.join(ref_dir, "sample.zarr"))
            # this repository's driver, served by the oracle
            written = {}
            real = (blockwise._do_block, blockwise.label_graph, vi.write_result)
            blockwise._do_block, blockwise.label_graph = tb.oracle_do_block, tb.cpu_label_graph
            vi.write_result = lambda fn, ds: written.update(ds)
            try:
                k2 = {k: v for k, v in kw.items() if k not in ("mutex", "context")}
                k2.setdefault("max_total_patch_distance_in_ps_multiples", 2)
                blockwise.main(pred_file, result_folder=os.path.join(work, "mine"), **k2)
            finally:
                blockwise._do_block, blockwise.label_graph, vi.write_result = real
            b = stored(os.path.join(work, "mine", "sample.zarr"))
            status = []
            if sorted(a) != sorted(b):
                status.append("GROUPS (%d vs %d)" % (len(a), len(b)))
            else:
                for k in a:
                    va, vb = a[k], b[k]
                    if va.shape != vb.shape or not np.array_equal(va.view(np.uint32) if va.dtype == np.float32 else va,
                                                                  vb.view(np.uint32) if vb.dtype == np.float32 else vb):
                        status.append("DATASET " + k)
            if want_inst is None or "vote_instances" not in written or not np.array_equal(written["vote_instances"], want_inst):
                status.append("INSTANCES")
            print(desc, "groups %d instances %d reference %.1f s:" % (len(a), 0 if want_inst is None else int(np.max(want_inst)), t_ref),
                  "ok" if not status else "DIFFER " + "; ".join(status[:6]), flush=True)
            bad += bool(status)
        except Exception as e:       # noqa: BLE001
            print(desc, "EXCEPTION %r" % (e,), flush=True)
            traceback.print_exc()
            bad += 1
        finally:
            shutil.rmtree(work, ignore_errors=True)
    print("%d trials, %d failures, %.0f s" % (args.trials, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
