#!/usr/bin/env python3
"""Benchmark-scale fixtures of the ORACLE (not of the reference: the reference itself finishes
32^3 / 5^3 in a minute and is what pins the oracle, tests/golden/gen_golden.py).

  python tests/golden/gen_scale_fixture.py [name ...]

Runs oracle/ppp_oracle_scale.to_instance_seg -- the oracle's C loops for S1 / S2 / S5 and the
scale forms of its host stages, each of which tests/test_oracle_scale.py holds equal to the literal
restatement -- on the synthetic generator of bench.py at the shapes below, with the SHIPPED
flylight flags (set-cover thinning + mutex watershed), and writes tests/golden/scale_<name>.npz:
the instance map, the selected patches, and hashes of the float stages.  The GPU parity test
(tests/test_gpu_parity.py::test_benchmark_scale_against_the_oracle) regenerates the input from the
seed and compares.  96^3 / 9^3 takes about a quarter of an hour on 8 cores and 25 GB of memory.
"""
import hashlib
import json
import os
import sys
import time
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

CASES = {
    # BASELINE config [2]'s patch and generator, 3/16 of its edge
    "s96_p9": ((96, 96, 96), (9, 9, 9), (24, 24, 24)),
    # BASELINE config [1]'s patch and generator at 64^3
    "s64_p7": ((64, 64, 64), (7, 7, 7), (18, 18, 18)),
    # BASELINE config [1] AT ITS STATED SIZE: 140^3, 7^3 (bench.py's flylight140_p7; round 6)
    "f140_p7": ((140, 140, 140), (7, 7, 7), (18, 18, 18)),
    # BASELINE config [0]'s image AT ITS STATED SIZE: one 696 x 520 image, 25 x 25 patches, kernel
    # semantics (bench.py's worm2d_p25; round 6)
    "w696x520_p25": ((1, 520, 696), (1, 25, 25), (1, 40, 40)),
}


def main(names):
    from oracle import ppp_oracle_scale as ors
    from patchperpix_amd import synth
    from patchperpix_amd.flags import FLYLIGHT
    for name in names:
        shape, ps, cell = CASES[name]
        kw = dict(FLYLIGHT)
        lab = synth.cell_labels(shape, list(cell), seed=0)
        pred = synth.pred_from_labels(lab, list(ps), seed=0)
        fg = lab != 0
        t0 = time.perf_counter()
        out = ors.to_instance_seg(pred, fg.copy(), fg.copy(), fg.astype(np.uint8), list(ps), dtype=np.uint32, **kw)
        dt = time.perf_counter() - t0
        inst = out["instances"]
        sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()   # noqa: E731
        np.savez_compressed(
            os.path.join(HERE, "scale_%s.npz" % name),
            shape=np.array(shape), patchshape=np.array(ps), cell=np.array(cell), seed=np.array(0),
            flags=json.dumps({k: v for k, v in kw.items() if isinstance(v, (bool, int, float, str))}),
            pred_f16_crc32=np.array(zlib.crc32(np.ascontiguousarray(pred.astype(np.float16)).tobytes())),
            instances=inst, n_ids=np.array(out["n_ids"]),
            cover_coords=out["cover_coords"].astype(np.int16), thin_coords=out["thin_coords"].astype(np.int16),
            scores_sha256=sha(out["scores"]), aff_sha256=sha(out["aff"]), pairs_sha256=sha(out["pairs"]),
            n_pairs=np.array(len(out["pairs"])), oracle_seconds=np.array(dt), oracle_threads=np.array(os.cpu_count()))
        print(name, "%.0f s" % dt, "instances", len(np.unique(inst)) - 1, "ids", out["n_ids"],
              "cover", len(out["cover_coords"]), "thin", len(out["thin_coords"]), "pairs", len(out["pairs"]))


if __name__ == "__main__":
    main(sys.argv[1:] or list(CASES))
