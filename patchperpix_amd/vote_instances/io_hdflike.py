"""HDF / zarr containers (reference: PatchPerPix/vote_instances/io_hdflike.py).

Neither ``zarr`` nor ``h5py`` is needed:

* zarr directory stores -- what the reference's prediction step writes (``output_format =
  "zarr"``, predict_no_gp.py:243-257) -- go through ``patchperpix_amd.minizarr`` (format 2 with
  the reference's Blosc-zstd-bitshuffle compressor); when ``zarr`` is importable it is used;
* HDF5 files -- ``.hdf`` predictions and the result files of the `label` task -- go through h5py
  when it is importable, otherwise through ``patchperpix_amd.minihdf5`` (the HDF5 C library of the
  image, bound with ctypes).  Only when neither exists are results written as a zarr store with
  the same dataset names and attributes (``write_datasets``) and does opening a ``.hdf`` raise.
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


def _hdf5_module():
    """h5py, else the ctypes layer over libhdf5, else None"""
    if os.environ.get("PPP_HDF5", "auto") != "mini":
        try:
            import h5py
            return h5py
        except ImportError:
            pass
    from .. import minihdf5
    return minihdf5 if minihdf5.available() else None


@contextlib.contextmanager
def open_container(path, mode="r"):
    if path.rstrip("/").endswith(".zarr"):
        yield _zarr_module().open(path, mode)
        return
    h5 = _hdf5_module()
    if h5 is None:  # pragma: no cover
        raise RuntimeError("reading/writing %s needs h5py or an HDF5 C library (zarr stores do not)" % path)
    f = h5.File(path, mode)
    try:
        yield f
    finally:
        f.close()


def write_datasets(out_fn, datasets, attrs=None):
    """The result file of the `label` task (vote_instances.py:542-554, stitch_patch_graph.py:
    849-870): datasets ``<res_key>`` / ``vote_foreground`` (/ ``<res_key>_masked``), gzip, attrs
    ``offset`` and ``resolution``.  ``out_fn`` ending in ``.hdf`` is written as HDF5 (h5py, or
    the HDF5 C library through ``minihdf5``); without either (and for ``.zarr`` names) a zarr store
    ``<stem>.zarr`` with the same dataset names, dtypes and attributes is written.  Returns the path written."""
    attrs = attrs or {"offset": (0, 0, 0), "resolution": (1, 1, 1)}
    if not out_fn.endswith(".zarr"):
        h5 = _hdf5_module()
        if h5 is not None:
            with h5.File(out_fn, "w") as f2:
                for key, data in datasets.items():
                    f2.create_dataset(key, data=data, compression="gzip")
                    for k, v in attrs.items():
                        f2[key].attrs[k] = v
            return out_fn
        out_fn = os.path.splitext(out_fn)[0] + ".zarr"
        logger.warning("no HDF5 library available: writing %s (same datasets and attributes)", out_fn)
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
            ds.attrs[k] = v if isinstance(v, str) else list(v)
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
