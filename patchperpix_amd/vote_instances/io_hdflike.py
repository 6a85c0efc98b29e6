"""HDF / zarr containers (reference: PatchPerPix/vote_instances/io_hdflike.py).

zarr directory stores -- what the reference's prediction step writes (``output_format =
"zarr"``, predict_no_gp.py:243-257) -- are read and written WITHOUT the ``zarr`` package:
``patchperpix_amd.minizarr`` implements format 2 with the reference's Blosc-zstd-bitshuffle
compressor.  When ``zarr`` is importable it is used instead.  HDF5 files still need ``h5py``
(absent from this image): opening one without it raises; results are then written as a zarr
store carrying the same dataset names and attributes (``write_datasets``).
"""
import contextlib
import logging
import os

import numpy as np

logger = logging.getLogger(__name__)


def _zarr_module():
    if os.environ.get("PPP_ZARR", "auto") != "mini":
        try:
            import zarr
            return zarr
        except ImportError:
            pass
    from .. import minizarr
    return minizarr


@contextlib.contextmanager
def open_container(path, mode="r"):
    if path.rstrip("/").endswith(".zarr"):
        yield _zarr_module().open(path, mode)
        return
    try:
        import h5py
    except ImportError as e:  # pragma: no cover
        raise RuntimeError("reading/writing %s needs the `h5py` package (zarr stores do not)" % path) from e
    f = h5py.File(path, mode)
    try:
        yield f
    finally:
        f.close()


def write_datasets(out_fn, datasets, attrs=None):
    """The result file of the `label` task (vote_instances.py:542-554, stitch_patch_graph.py:
    849-870): datasets ``<res_key>`` / ``vote_foreground`` (/ ``<res_key>_masked``), gzip, attrs
    ``offset`` and ``resolution``.  ``out_fn`` ending in ``.hdf`` is written with h5py when that
    is importable; otherwise (and for ``.zarr`` names) a zarr store ``<stem>.zarr`` with the same
    dataset names, dtypes and attributes is written.  Returns the path written."""
    attrs = attrs or {"offset": (0, 0, 0), "resolution": (1, 1, 1)}
    if not out_fn.endswith(".zarr"):
        try:
            import h5py
            with h5py.File(out_fn, "w") as f2:
                for key, data in datasets.items():
                    f2.create_dataset(key, data=data, compression="gzip")
                    for k, v in attrs.items():
                        f2[key].attrs[k] = v
            return out_fn
        except ImportError:
            out_fn = os.path.splitext(out_fn)[0] + ".zarr"
            logger.warning("h5py not available: writing %s (same datasets and attributes)", out_fn)
    zf = _zarr_module().open(out_fn, mode="w")
    for key, data in datasets.items():
        data = np.asarray(data)
        chunks = tuple(min(int(s), 128) for s in data.shape)
        if hasattr(zf, "create_dataset") and zf.__class__.__module__.startswith("zarr"):
            ds = zf.create_dataset(key, data=data, chunks=chunks)
        else:
            ds = zf.create(key, shape=data.shape, chunks=chunks, dtype=data.dtype, overwrite=True)
            ds[...] = data
        for k, v in attrs.items():
            ds.attrs[k] = list(v)
    return out_fn


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
