"""patchperpix_amd.minihdf5: HDF5 through the HDF5 C library (ctypes), no h5py.  The files are
written and read back by libhdf5 itself; what is tested here is the binding (dtypes incl. the
hand-built IEEE float16 type, hyperslab selections, attributes, groups created on the way) and the
`label` task's result / prediction files going through it."""
import os

import numpy as np
import pytest

from patchperpix_amd import minihdf5

pytestmark = pytest.mark.skipif(not minihdf5.available(), reason="no HDF5 C library in this image")

DTYPES = ["u1", "u2", "u4", "u8", "i1", "i2", "i4", "i8", "f2", "f4", "f8"]


def test_roundtrip_dtypes_slices_attributes(tmp_path):
    fn = str(tmp_path / "a.hdf")
    rng = np.random.default_rng(0)
    arrays = {dt: (rng.random((5, 6, 7)) * 100 - 20).astype(dt) for dt in DTYPES}
    with minihdf5.File(fn, "w") as f:
        for dt, a in arrays.items():
            ds = f.create_dataset("volumes/x_" + dt, data=a, compression="gzip")
            ds.attrs["offset"] = (0, 0, 0)
            ds.attrs["resolution"] = [1, 1, 1]
            ds.attrs["note"] = "abc"
            ds.attrs["scale"] = 2.5
        late = f.create_dataset("late", shape=(4, 5), dtype=np.float32)
        late[1:3, :] = np.ones((2, 5))
        late[0, 0] = 7
        f.create_dataset("flags", data=np.array([True, False, True]))
    with open(fn, "rb") as raw:
        assert raw.read(8) == b"\x89HDF\r\n\x1a\n"                # the HDF5 superblock signature
    with minihdf5.File(fn, "r") as f:
        assert sorted(f.keys()) == ["flags", "late", "volumes"]
        assert sorted(f["volumes"].keys()) == sorted("x_" + d for d in DTYPES)
        assert "volumes/x_u1" in f and "volumes/none/deeper" not in f and "none" not in f
        with pytest.raises(KeyError):
            f["none"]
        for dt, a in arrays.items():
            ds = f["volumes/x_" + dt]
            assert ds.dtype == np.dtype(dt) and ds.shape == a.shape and ds.ndim == 3
            assert np.array_equal(ds[...], a)
            assert np.array_equal(ds[1:4, 2, :], a[1:4, 2, :])
            assert np.array_equal(ds[-1], a[-1])
            assert np.array_equal(np.asarray(ds), a)
            assert ds[2, 3, 4] == a[2, 3, 4]
            assert list(ds.attrs["offset"]) == [0, 0, 0] and list(ds.attrs["resolution"]) == [1, 1, 1]
            assert ds.attrs["note"] == "abc" and ds.attrs["scale"] == 2.5 and "nothing" not in ds.attrs
        want = np.zeros((4, 5), dtype=np.float32)
        want[1:3] = 1
        want[0, 0] = 7
        assert np.array_equal(f["late"][...], want)
        assert f["flags"][...].tolist() == [1, 0, 1]
        out = np.empty((2, 6, 7), dtype=np.float16)
        f["volumes/x_f2"].read_into((slice(1, 3),), out)
        assert np.array_equal(out, arrays["f2"][1:3])
        with pytest.raises(IndexError):
            f["late"][::2]


def test_append_and_overwrite(tmp_path):
    fn = str(tmp_path / "b.hdf")
    with minihdf5.File(fn, "w") as f:
        f.create_dataset("a", data=np.arange(5))
    with minihdf5.File(fn, "a") as f:
        f.create_dataset("b/c", data=np.arange(3, dtype=np.uint16))
        f.create_dataset("a", data=np.arange(7, dtype=np.int32))          # replaces
    with minihdf5.File(fn, "r") as f:
        assert f["a"].shape == (7,) and f["a"].dtype == np.int32
        assert f["b/c"][...].tolist() == [0, 1, 2]
    with pytest.raises(OSError):
        minihdf5.File(str(tmp_path / "missing.hdf"), "r")


def test_result_file_of_the_label_task_is_hdf5(tmp_path, monkeypatch):
    """write_datasets / open_container (the reference's result format, vote_instances.py:542-554)
    without h5py: an .hdf file with gzip datasets and the offset / resolution attributes."""
    from patchperpix_amd.vote_instances import io_hdflike
    monkeypatch.setenv("PPP_HDF5", "mini")
    inst = (np.arange(4 * 5 * 6) % 7).astype(np.uint16).reshape(4, 5, 6)
    out = io_hdflike.write_datasets(str(tmp_path / "s.hdf"), {"vote_instances": inst,
                                                              "vote_foreground": (inst > 0).astype(np.uint8)})
    assert out.endswith("s.hdf") and os.path.isfile(out)
    with io_hdflike.open_container(out, "r") as f:
        assert sorted(f.keys()) == ["vote_foreground", "vote_instances"]
        assert np.array_equal(np.array(f["vote_instances"]), inst)
        assert f["vote_instances"].dtype == np.uint16
        assert list(f["vote_instances"].attrs["resolution"]) == [1, 1, 1]


def test_hdf_prediction_loads_like_the_zarr_one(tmp_path, monkeypatch):
    """utilVoteInstances.loadAffinities on an .hdf prediction (aff_key / numinst_key datasets)
    gives what it gives on the same arrays in a zarr store."""
    from patchperpix_amd import minizarr
    from patchperpix_amd.vote_instances import utilVoteInstances as util
    monkeypatch.setenv("PPP_HDF5", "mini")
    rng = np.random.default_rng(3)
    affs = rng.random((27, 6, 7, 8)).astype(np.float16)
    prob = rng.random((3, 6, 7, 8)).astype(np.float16)
    with minihdf5.File(str(tmp_path / "p.hdf"), "w") as f:
        f.create_dataset("volumes/pred_affs", data=affs, compression="gzip")
        f.create_dataset("volumes/pred_numinst", data=prob, compression="gzip")
    z = minizarr.open(str(tmp_path / "p.zarr"), "w")
    z.create_dataset("volumes/pred_affs", data=affs)
    z.create_dataset("volumes/pred_numinst", data=prob)
    kw = dict(aff_key="volumes/pred_affs", numinst_key="volumes/pred_numinst", patchshape=[3, 3, 3],
              overlapping_inst=True, numinst_threshs=[0.9, 0.1], fg_thresh_vi=-1.0, patch_threshold=0.5)
    a = util.loadAffinities(str(tmp_path / "p.hdf"), "", **kw)
    b = util.loadAffinities(str(tmp_path / "p.zarr"), "", **kw)
    assert a is not None and len(a) == len(b) == 3
    for x, y in zip(a, b):
        assert np.array_equal(np.asarray(x), np.asarray(y))
