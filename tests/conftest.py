import glob
import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def golden_names():
    # (bw_*.npz: goldens of the blockwise driver, tests/test_blockwise.py loads them itself;
    # ds_* / ae_forward_*.npz: the reference's decode_sample / Autoencoder.forward, tests/test_decode.py;
    # scale_*.npz: the oracle's benchmark-scale fixtures, tests/test_gpu_parity.py; np_*.npz: the
    # reference's NumPy-semantics path, tests/test_numpy_semantics.py; large_*.npz: the larger
    # reference-made case whose input is regenerated, tests/test_large_golden.py)
    return sorted(n for n in (os.path.splitext(os.path.basename(p))[0]
                              for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
                  if not n.startswith(("bw_", "scale_", "np_", "large_", "ds_", "ae_forward_")))


class Golden:
    """One golden case: seeded inputs + the reference's stage-by-stage outputs
    (tests/golden/gen_golden.py)."""

    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.pred = self.z["pred_f16"].astype(np.float32)
        self.foreground = self.z["foreground"]
        self.numinst = self.z["numinst"]
        self.patchshape = [int(p) for p in self.z["patchshape"]]
        self.kw = json.loads(str(self.z["flags"]))
        self.kw.setdefault("max_total_patch_distance_in_ps_multiples", 2)
        self.overlap_mask = 1 * (self.numinst > 1)

    def has(self, key):
        return key in self.z.files

    def __getitem__(self, key):
        return self.z[key]


@pytest.fixture(params=golden_names())
def golden(request):
    return Golden(request.param)


def same_partition(a, b):
    """True if label volumes a and b are equal up to a permutation of the ids."""
    a = np.asarray(a).ravel().astype(np.int64)
    b = np.asarray(b).ravel().astype(np.int64)
    if a.shape != b.shape or not np.array_equal(a == 0, b == 0):
        return False
    pairs = np.unique(np.stack([a, b], axis=1), axis=0)
    return len(np.unique(pairs[:, 0])) == len(pairs) and \
        len(np.unique(pairs[:, 1])) == len(pairs)


@pytest.fixture(autouse=True)
def _library_switches_follow_the_environment():
    """The library reads its PPP_* development switches once.  A test that changed one
    (monkeypatch.setenv + backend.reload_env) must not leave its value behind: after every test
    -- this fixture is set up first, hence finalised after monkeypatch has restored the
    environment -- a loaded library is told to look again."""
    yield
    from patchperpix_amd import backend
    if backend._LIB is not None:
        backend.reload_env()
