"""Which S1 kernel serves each fresh-input parity case (tests/test_gpu_parity.py CASES)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from patchperpix_amd import backend, synth  # noqa: E402
from test_gpu_parity import CASES  # noqa: E402
from tests_flags import FLYLIGHT  # noqa: E402

for i, (shape, ps, skw, flags) in enumerate(CASES):
    kw = dict(FLYLIGHT, **flags)
    c = synth.make_case(shape, ps, **skw)
    P = backend.make_params(shape, ps, **kw)
    pred = torch.from_numpy(c["pred"].astype(np.float32)).cuda()
    ov = torch.from_numpy((c["numinst"] > 1).astype(np.uint8)).cuda() if P.use_overlap else None
    backend.consensus(pred, ov, P)
    print(i, shape, ps, backend.lib().ppp_consensus_kernel_name().decode())
