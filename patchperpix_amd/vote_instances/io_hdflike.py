"""HDF / zarr containers (reference: PatchPerPix/vote_instances/io_hdflike.py).

The reference reads predictions with h5py / zarr.  Those packages are optional here; when
they are not importable a clear error is raised at open time (nothing else in the hot path
depends on them -- ``.npy`` inputs and in-memory arrays always work).
"""
import contextlib
import logging

import numpy as np

logger = logging.getLogger(__name__)


@contextlib.contextmanager
def open_container(path, mode="r"):
    if path.endswith(".zarr"):
        try:
            import zarr
        except ImportError as e:  # pragma: no cover - depends on the image
            raise RuntimeError("reading %s needs the `zarr` package" % path) from e
        yield zarr.open(path, mode)
        return
    try:
        import h5py
    except ImportError as e:  # pragma: no cover
        raise RuntimeError("reading/writing %s needs the `h5py` package" % path) from e
    f = h5py.File(path, mode)
    try:
        yield f
    finally:
        f.close()


class IoBase:
    """io_hdflike.py:6-109: block reader/writer over a set of keys."""

    def __init__(self, path, keys, mode="r", channel_order=None, voxel_size=None):
        self.path, self.keys, self.mode = path, list(keys), mode
        self.channel_order = channel_order
        self.voxel_size = voxel_size
        self._cm = open_container(path, mode)
        self.ff = self._cm.__enter__()
        self.datasets = [self.ff[k] for k in self.keys]

    @property
    def shape(self):
        return self.datasets[0].shape

    def read(self, bb, key=None):
        ds = self.datasets[0] if key is None else self.datasets[self.keys.index(key)]
        return np.array(ds[bb])

    def write(self, out, out_bb):
        for ds in self.datasets:
            ds[out_bb] = out

    def close(self):
        self._cm.__exit__(None, None, None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class IoZarr(IoBase):
    pass


class IoHDF5(IoBase):
    pass
