#!/usr/bin/env python3
"""setKernelBuildOptions: which -D switches the reference would compile its kernels with
(utilVoteInstances.py:389-449 -- precedence of the three background rules, the th < 0.5 switch, DEFAULTS of absent
keys, the value rule, the rank / patch-graph switches) against the package's mirror, which derives them from
backend.make_params -- i.e. from the same decisions the kernels run with.  Random flag dictionaries with keys present
or ABSENT; where the reference raises, the mirror must raise too.  Development container only.

  python tests/golden/fuzz_build_options_vs_reference.py [--trials 3000]
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
    ap.add_argument("--trials", type=int, default=3000)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    gg.install_stubs()
    gg.install_fake_cuda_code()
    sys.path.insert(0, gg.REF_VI)
    import logging
    logging.basicConfig(level=logging.CRITICAL)
    import utilVoteInstances as ref
    from patchperpix_amd.vote_instances import utilVoteInstances as mine
    rng = np.random.default_rng(args.seed)
    keys = ["vi_bg_use_inv_th", "vi_bg_use_half_th", "vi_bg_use_less_than_th", "overlapping_inst",
            "consensus_norm_prob_product", "consensus_prob_product", "consensus_norm_aff", "consensus_interleaved_cnt",
            "rank_norm_patch_score", "rank_int_counter", "patch_graph_norm_aff"]
    bad = both_raised = 0
    for trial in range(args.trials):
        kw = {"patch_threshold": float(rng.choice([0.3, 0.5, 0.7, 0.9]))}
        for k in keys:
            r = rng.integers(0, 3)
            if r < 2:
                kw[k] = bool(r)
        step = [None, "consensus", "rank", "patch_graph"][int(rng.integers(0, 4))]
        try:
            want = ref.setKernelBuildOptions(step=step, **kw)
            err = None
        except Exception as e:       # noqa: BLE001
            want, err = None, e
        try:
            got = mine.setKernelBuildOptions(step=step, **kw)
            gerr = None
        except Exception as e:       # noqa: BLE001
            got, gerr = None, e
        if err is not None and gerr is not None:
            both_raised += 1
            continue
        if (err is None) != (gerr is None) or want != got:
            bad += 1
            if bad <= 15:
                print("trial", trial, step, kw, "reference:", want if err is None else repr(err), "mirror:", got if gerr is None else repr(gerr))
    print("%d trials, %d failures (%d where both refuse)" % (args.trials, bad, both_raised))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
