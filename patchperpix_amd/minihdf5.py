"""HDF5 containers without ``h5py``: a small h5py-shaped layer over the HDF5 C library
(``libhdf5``, loaded with ctypes).

The reference reads ``.hdf`` predictions and writes its results as HDF5 through h5py
(vote_instances.py:542-554, stitch_patch_graph.py:849-870, utilVoteInstances.py:136-322,
io_hdflike.py).  h5py is absent from this image, the C library it wraps is not (HDF5 1.10 under
/opt/conda/lib): this module binds the two dozen calls the path needs -- files, groups (created
on the way), datasets (contiguous or chunked + gzip like ``compression="gzip"``), hyperslab
reads / writes for basic slices, scalar / 1-d numeric and string attributes.  The bytes on disk
are produced by the HDF5 library itself, so any HDF5 reader (h5py included) opens them.

Only what the path uses is covered: numeric dtypes (u/i 8-64, f16/32/64, native order), basic
indexing (ints, slices with step 1, Ellipsis).  ``available()`` tells whether a libhdf5 was found.
"""
import ctypes
import ctypes.util
import os

import numpy as np

hid_t = ctypes.c_int64
hsize_t = ctypes.c_uint64
herr_t = ctypes.c_int

_LIB = None
_SEARCH = ("libhdf5.so", "/opt/conda/lib/libhdf5.so", "/opt/conda/lib/libhdf5.so.103",
           "libhdf5_serial.so", "libhdf5.so.103", "libhdf5.so.200", "libhdf5.so.310")

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC, H5F_ACC_EXCL = 0, 1, 2, 4
H5S_SELECT_SET = 0
H5I_GROUP, H5I_DATASET = 2, 5
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
H5T_SGN_NONE = 0
H5T_VARIABLE = ctypes.c_size_t(-1).value
H5_INDEX_NAME, H5_ITER_INC = 0, 0


def _load():
    global _LIB
    if _LIB is not None:
        return _LIB
    names = list(_SEARCH)
    found = ctypes.util.find_library("hdf5")
    if found:
        names.insert(0, found)
    if os.environ.get("PPP_LIBHDF5"):
        names.insert(0, os.environ["PPP_LIBHDF5"])
    err = None
    for n in names:
        try:
            L = ctypes.CDLL(n)
        except OSError as e:
            err = e
            continue
        if L.H5open() < 0:
            continue
        sig = {
            "H5Fcreate": (hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t, hid_t]),
            "H5Fopen": (hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t]),
            "H5Fclose": (herr_t, [hid_t]), "H5Fflush": (herr_t, [hid_t, ctypes.c_int]),
            "H5Screate_simple": (hid_t, [ctypes.c_int, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)]),
            "H5Screate": (hid_t, [ctypes.c_int]),
            "H5Sclose": (herr_t, [hid_t]),
            "H5Sselect_hyperslab": (herr_t, [hid_t, ctypes.c_int, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t),
                                             ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)]),
            "H5Sget_simple_extent_ndims": (ctypes.c_int, [hid_t]),
            "H5Sget_simple_extent_dims": (ctypes.c_int, [hid_t, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)]),
            "H5Pcreate": (hid_t, [hid_t]), "H5Pclose": (herr_t, [hid_t]),
            "H5Pset_chunk": (herr_t, [hid_t, ctypes.c_int, ctypes.POINTER(hsize_t)]),
            "H5Pset_deflate": (herr_t, [hid_t, ctypes.c_uint]),
            "H5Pset_create_intermediate_group": (herr_t, [hid_t, ctypes.c_uint]),
            "H5Dcreate2": (hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
            "H5Dopen2": (hid_t, [hid_t, ctypes.c_char_p, hid_t]), "H5Dclose": (herr_t, [hid_t]),
            "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
            "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
            "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]),
            "H5Gcreate2": (hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t]),
            "H5Gclose": (herr_t, [hid_t]),
            "H5Oopen": (hid_t, [hid_t, ctypes.c_char_p, hid_t]), "H5Oclose": (herr_t, [hid_t]),
            "H5Iget_type": (ctypes.c_int, [hid_t]),
            "H5Lexists": (ctypes.c_int, [hid_t, ctypes.c_char_p, hid_t]),
            "H5Ldelete": (herr_t, [hid_t, ctypes.c_char_p, hid_t]),
            "H5Tcopy": (hid_t, [hid_t]), "H5Tclose": (herr_t, [hid_t]),
            "H5Tset_size": (herr_t, [hid_t, ctypes.c_size_t]),
            "H5Tset_fields": (herr_t, [hid_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                                       ctypes.c_size_t, ctypes.c_size_t]),
            "H5Tset_ebias": (herr_t, [hid_t, ctypes.c_size_t]),
            "H5Tget_class": (ctypes.c_int, [hid_t]), "H5Tget_size": (ctypes.c_size_t, [hid_t]),
            "H5Tget_sign": (ctypes.c_int, [hid_t]), "H5Tis_variable_str": (ctypes.c_int, [hid_t]),
            "H5Acreate2": (hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t, hid_t]),
            "H5Aopen": (hid_t, [hid_t, ctypes.c_char_p, hid_t]), "H5Aclose": (herr_t, [hid_t]),
            "H5Awrite": (herr_t, [hid_t, hid_t, ctypes.c_void_p]),
            "H5Aread": (herr_t, [hid_t, hid_t, ctypes.c_void_p]),
            "H5Aexists": (ctypes.c_int, [hid_t, ctypes.c_char_p]),
            "H5Adelete": (herr_t, [hid_t, ctypes.c_char_p]),
            "H5Aget_space": (hid_t, [hid_t]), "H5Aget_type": (hid_t, [hid_t]),
            "H5Eset_auto2": (herr_t, [hid_t, ctypes.c_void_p, ctypes.c_void_p]),
            "H5free_memory": (herr_t, [ctypes.c_void_p]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        L.H5Eset_auto2(0, None, None)            # errors are reported through return codes
        _LIB = L
        return L
    raise RuntimeError("no HDF5 C library found (tried %s): %s" % (", ".join(names), err))


def available():
    try:
        _load()
        return True
    except RuntimeError:
        return False


def _g(name):
    return hid_t.in_dll(_load(), name).value


def _dims(seq):
    return (hsize_t * len(seq))(*[int(v) for v in seq])


_NATIVE = {"u1": "H5T_NATIVE_UINT8_g", "u2": "H5T_NATIVE_UINT16_g", "u4": "H5T_NATIVE_UINT32_g",
           "u8": "H5T_NATIVE_UINT64_g", "i1": "H5T_NATIVE_INT8_g", "i2": "H5T_NATIVE_INT16_g",
           "i4": "H5T_NATIVE_INT32_g", "i8": "H5T_NATIVE_INT64_g", "f4": "H5T_NATIVE_FLOAT_g",
           "f8": "H5T_NATIVE_DOUBLE_g", "b1": "H5T_NATIVE_UINT8_g"}


def _mem_type(dtype):
    """(hid_t, must_close) for a NumPy dtype"""
    dtype = np.dtype(dtype)
    code = dtype.kind + str(dtype.itemsize)
    if code == "f2":
        L = _load()
        t = L.H5Tcopy(_g("H5T_NATIVE_FLOAT_g"))
        # IEEE binary16: sign bit 15, exponent 10..14, mantissa 0..9, bias 15 (fields before size:
        # the size can only shrink once the fields fit)
        if L.H5Tset_fields(t, 15, 10, 5, 0, 10) < 0 or L.H5Tset_size(t, 2) < 0 or L.H5Tset_ebias(t, 15) < 0:
            raise RuntimeError("cannot build the float16 datatype")
        return t, True
    if code not in _NATIVE:
        raise TypeError("dtype %s is not supported" % dtype)
    return _g(_NATIVE[code]), False


def _np_type(tid):
    L = _load()
    cls, size = L.H5Tget_class(tid), int(L.H5Tget_size(tid))
    if cls == H5T_INTEGER:
        return np.dtype(("u" if L.H5Tget_sign(tid) == H5T_SGN_NONE else "i") + str(size))
    if cls == H5T_FLOAT:
        return np.dtype("f" + str(size))
    raise TypeError("HDF5 datatype class %d is not supported" % cls)


class _Attrs:
    def __init__(self, obj_id):
        self._id = obj_id

    def __contains__(self, name):
        return _load().H5Aexists(self._id, name.encode()) > 0

    def __setitem__(self, name, value):
        L = _load()
        if name in self:
            L.H5Adelete(self._id, name.encode())
        if isinstance(value, (str, bytes)):
            raw = value.encode() if isinstance(value, str) else value
            t = L.H5Tcopy(_g("H5T_C_S1_g"))
            L.H5Tset_size(t, max(1, len(raw)))
            space = L.H5Screate(0)                  # H5S_SCALAR
            a = L.H5Acreate2(self._id, name.encode(), t, space, 0, 0)
            buf = ctypes.create_string_buffer(raw, max(1, len(raw)))
            ok = a >= 0 and L.H5Awrite(a, t, buf) >= 0
            if a >= 0:
                L.H5Aclose(a)
            L.H5Sclose(space)
            L.H5Tclose(t)
        else:
            arr = np.ascontiguousarray(np.asarray(value))
            if arr.dtype.kind not in "uifb":
                raise TypeError("attribute %s: unsupported value %r" % (name, value))
            if arr.dtype.kind == "b":
                arr = arr.astype(np.uint8)
            t, close = _mem_type(arr.dtype)
            space = L.H5Screate(0) if arr.ndim == 0 else L.H5Screate_simple(arr.ndim, _dims(arr.shape), None)
            a = L.H5Acreate2(self._id, name.encode(), t, space, 0, 0)
            ok = a >= 0 and L.H5Awrite(a, t, arr.ctypes.data_as(ctypes.c_void_p)) >= 0
            if a >= 0:
                L.H5Aclose(a)
            L.H5Sclose(space)
            if close:
                L.H5Tclose(t)
        if not ok:
            raise OSError("cannot write attribute %s" % name)

    def __getitem__(self, name):
        L = _load()
        a = L.H5Aopen(self._id, name.encode(), 0)
        if a < 0:
            raise KeyError(name)
        try:
            t, space = L.H5Aget_type(a), L.H5Aget_space(a)
            try:
                nd = L.H5Sget_simple_extent_ndims(space)
                shape = ()
                if nd > 0:
                    d = (hsize_t * nd)()
                    L.H5Sget_simple_extent_dims(space, d, None)
                    shape = tuple(int(v) for v in d)
                if L.H5Tget_class(t) == H5T_STRING:
                    if L.H5Tis_variable_str(t) > 0:
                        p = ctypes.c_char_p()
                        if L.H5Aread(a, t, ctypes.byref(p)) < 0:
                            raise OSError("cannot read attribute %s" % name)
                        out = (p.value or b"").decode()
                        L.H5free_memory(p)
                        return out
                    buf = ctypes.create_string_buffer(int(L.H5Tget_size(t)) + 1)
                    if L.H5Aread(a, t, buf) < 0:
                        raise OSError("cannot read attribute %s" % name)
                    return buf.value.decode()
                dt = _np_type(t)
                out = np.empty(shape, dtype=dt)
                mt, close = _mem_type(dt)
                ok = L.H5Aread(a, mt, out.ctypes.data_as(ctypes.c_void_p)) >= 0
                if close:
                    L.H5Tclose(mt)
                if not ok:
                    raise OSError("cannot read attribute %s" % name)
                return out[()] if out.ndim == 0 else out
            finally:
                L.H5Sclose(space)
                L.H5Tclose(t)
        finally:
            L.H5Aclose(a)

    def get(self, name, default=None):
        return self[name] if name in self else default


def _normalise(sel, shape):
    """basic index -> (start, count, squeeze axes); ints, step-1 slices, Ellipsis"""
    if not isinstance(sel, tuple):
        sel = (sel,)
    if any(s is Ellipsis for s in sel):
        i = [k for k, s in enumerate(sel) if s is Ellipsis][0]
        sel = sel[:i] + (slice(None),) * (len(shape) - (len(sel) - 1)) + sel[i + 1:]
    sel = sel + (slice(None),) * (len(shape) - len(sel))
    if len(sel) != len(shape):
        raise IndexError("too many indices")
    start, count, squeeze = [], [], []
    for ax, (s, n) in enumerate(zip(sel, shape)):
        if isinstance(s, (int, np.integer)):
            s = int(s) + (n if s < 0 else 0)
            if not 0 <= s < n:
                raise IndexError("index out of range")
            start.append(s), count.append(1), squeeze.append(ax)
        elif isinstance(s, slice):
            a, b, st = s.indices(n)
            if st != 1:
                raise IndexError("only step-1 slices are supported")
            start.append(a), count.append(max(0, b - a))
        else:
            raise IndexError("unsupported index %r" % (s,))
    return start, count, tuple(squeeze)


class Dataset:
    def __init__(self, did, name):
        L = _load()
        self._id, self.name = did, name
        space, t = L.H5Dget_space(did), L.H5Dget_type(did)
        nd = L.H5Sget_simple_extent_ndims(space)
        d = (hsize_t * max(nd, 1))()
        if nd > 0:
            L.H5Sget_simple_extent_dims(space, d, None)
        self.shape = tuple(int(v) for v in d[:nd])
        self.dtype = _np_type(t)
        L.H5Sclose(space)
        L.H5Tclose(t)
        self.attrs = _Attrs(did)

    ndim = property(lambda self: len(self.shape))
    size = property(lambda self: int(np.prod(self.shape)))

    def __len__(self):
        return self.shape[0]

    def _transfer(self, sel, buf, write):
        L = _load()
        start, count, squeeze = _normalise(sel, self.shape)
        if write:
            arr = np.ascontiguousarray(np.broadcast_to(np.asarray(buf, dtype=self.dtype),
                                                       [c for ax, c in enumerate(count) if ax not in squeeze]))
        else:
            arr = np.empty(count, dtype=self.dtype) if buf is None else buf
            if buf is not None and (buf.dtype != self.dtype or not buf.flags.c_contiguous
                                    or int(buf.size) != int(np.prod(count))):
                raise ValueError("read_into needs a C-contiguous %s buffer of %d elements" %
                                 (self.dtype, int(np.prod(count))))
        if int(np.prod(count)) > 0:
            mt, close = _mem_type(self.dtype)
            fs = L.H5Dget_space(self._id)
            ms = fs
            if self.shape:
                L.H5Sselect_hyperslab(fs, H5S_SELECT_SET, _dims(start), None, _dims(count), None)
                ms = L.H5Screate_simple(len(count), _dims(count), None)
            fn = L.H5Dwrite if write else L.H5Dread
            ok = fn(self._id, mt, ms if self.shape else 0, fs if self.shape else 0, 0,
                    arr.ctypes.data_as(ctypes.c_void_p)) >= 0
            if self.shape:
                L.H5Sclose(ms)
            L.H5Sclose(fs)
            if close:
                L.H5Tclose(mt)
            if not ok:
                raise OSError("HDF5 %s of %s failed" % ("write" if write else "read", self.name))
        if write or buf is not None:
            return None
        return arr.reshape([c for ax, c in enumerate(count) if ax not in squeeze])

    def __getitem__(self, sel):
        out = self._transfer(sel, None, False)
        return out[()] if out.ndim == 0 else out

    def __setitem__(self, sel, value):
        self._transfer(sel, value, True)

    def read_into(self, sel, out):
        """read the selection straight into `out` (C-contiguous, this dataset's dtype)"""
        self._transfer(sel, out, False)

    def __array__(self, dtype=None, copy=None):
        a = self[...]
        return a if dtype is None else a.astype(dtype)

    def _close(self):
        if self._id >= 0:
            _load().H5Dclose(self._id)
            self._id = -1


class Group:
    def __init__(self, file, gid, name):
        self._file, self._id, self.name = file, gid, name
        self.attrs = _Attrs(gid)

    def _path(self, key):
        return key if key.startswith("/") else (self.name.rstrip("/") + "/" + key)

    def __contains__(self, key):
        L = _load()
        path = self._path(key).strip("/")
        # every link of the path must exist (H5Lexists fails on a missing intermediate group)
        cur = ""
        for part in path.split("/"):
            cur += "/" + part
            if L.H5Lexists(self._file._id, cur.encode(), 0) <= 0:
                return False
        return True

    def __getitem__(self, key):
        return self._file._open(self._path(key))

    def keys(self):
        L = _load()
        names = []
        cb_t = ctypes.CFUNCTYPE(herr_t, hid_t, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p)

        def cb(_g, name, _info, _data):
            names.append(name.decode())
            return 0
        L.H5Literate.restype = herr_t
        L.H5Literate.argtypes = [hid_t, ctypes.c_int, ctypes.c_int, ctypes.POINTER(hsize_t), cb_t, ctypes.c_void_p]
        idx = hsize_t(0)
        if L.H5Literate(self._id, H5_INDEX_NAME, H5_ITER_INC, ctypes.byref(idx), cb_t(cb), None) < 0:
            raise OSError("cannot list %s" % self.name)
        return names

    def __iter__(self):
        return iter(self.keys())

    def create_group(self, key):
        return self._file._create_group(self._path(key))

    def require_group(self, key):
        return self[key] if key in self else self.create_group(key)

    def create_dataset(self, key, shape=None, dtype=None, data=None, chunks=None, compression=None,
                       compression_opts=None, **_ignored):
        return self._file._create_dataset(self._path(key), shape, dtype, data, chunks, compression,
                                          compression_opts)

    def __delitem__(self, key):
        if _load().H5Ldelete(self._file._id, self._path(key).encode(), 0) < 0:
            raise KeyError(key)


class File(Group):
    """``minihdf5.File(path, mode)``: modes "r", "r+", "a", "w", "w-" as h5py."""

    def __init__(self, path, mode="r"):
        L = _load()
        p = os.fsencode(path)
        if mode == "r":
            fid = L.H5Fopen(p, H5F_ACC_RDONLY, 0)
        elif mode == "r+" or (mode == "a" and os.path.exists(path)):
            fid = L.H5Fopen(p, H5F_ACC_RDWR, 0)
        elif mode in ("w", "a"):
            fid = L.H5Fcreate(p, H5F_ACC_TRUNC, 0, 0)
        elif mode in ("w-", "x"):
            fid = L.H5Fcreate(p, H5F_ACC_EXCL, 0, 0)
        else:
            raise ValueError("mode %r" % mode)
        if fid < 0:
            raise OSError("cannot open %s (mode %s) as HDF5" % (path, mode))
        self.filename, self.mode = path, mode
        self._open_objs = []
        Group.__init__(self, self, fid, "/")

    def _lcpl(self):
        L = _load()
        lcpl = L.H5Pcreate(_g("H5P_CLS_LINK_CREATE_ID_g"))
        L.H5Pset_create_intermediate_group(lcpl, 1)
        return lcpl

    def _open(self, path):
        L = _load()
        if path.strip("/") == "":
            return self
        if path not in self:
            raise KeyError(path)
        oid = L.H5Oopen(self._id, path.encode(), 0)
        if oid < 0:
            raise KeyError(path)
        kind = L.H5Iget_type(oid)
        if kind == H5I_DATASET:
            L.H5Oclose(oid)
            ds = Dataset(L.H5Dopen2(self._id, path.encode(), 0), path)
            self._open_objs.append(ds)
            return ds
        if kind == H5I_GROUP:
            g = Group(self, oid, path)
            self._open_objs.append(g)
            return g
        L.H5Oclose(oid)
        raise TypeError("%s is neither a group nor a dataset" % path)

    def _create_group(self, path):
        L = _load()
        lcpl = self._lcpl()
        gid = L.H5Gcreate2(self._id, path.encode(), lcpl, 0, 0)
        L.H5Pclose(lcpl)
        if gid < 0:
            raise OSError("cannot create group %s" % path)
        g = Group(self, gid, path)
        self._open_objs.append(g)
        return g

    def _create_dataset(self, path, shape, dtype, data, chunks, compression, compression_opts):
        L = _load()
        if data is not None:
            data = np.asarray(data)
            if data.dtype == np.bool_:
                data = data.astype(np.uint8)
            shape = data.shape if shape is None else tuple(shape)
            dtype = data.dtype if dtype is None else np.dtype(dtype)
            data = np.ascontiguousarray(data.astype(dtype, copy=False)).reshape(shape)
        if shape is None or dtype is None:
            raise TypeError("create_dataset needs data or shape + dtype")
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        dtype = np.dtype(dtype)
        if compression not in (None, "gzip", False) and not isinstance(compression, int):
            raise NotImplementedError("compression %r (gzip only)" % (compression,))
        level = None
        if compression == "gzip":
            level = 4 if compression_opts is None else int(compression_opts)
        elif isinstance(compression, int) and not isinstance(compression, bool):
            level = int(compression)
        if chunks is True or (chunks is None and level is not None):
            # about 1 MB per chunk, cut along the leading axes first (h5py guesses likewise)
            c = list(shape)
            while c and int(np.prod(c)) * dtype.itemsize > (1 << 20):
                ax = int(np.argmax(c))
                c[ax] = (c[ax] + 1) // 2
            chunks = tuple(max(1, v) for v in c)
        space = L.H5Screate(0) if not shape else L.H5Screate_simple(len(shape), _dims(shape), None)
        dcpl = L.H5Pcreate(_g("H5P_CLS_DATASET_CREATE_ID_g"))
        if chunks and shape and int(np.prod(shape)) > 0:
            L.H5Pset_chunk(dcpl, len(shape), _dims([min(int(c), max(1, s)) for c, s in zip(chunks, shape)]))
            if level is not None:
                L.H5Pset_deflate(dcpl, level)
        t, close = _mem_type(dtype)
        lcpl = self._lcpl()
        if path in self:
            L.H5Ldelete(self._id, path.encode(), 0)
        did = L.H5Dcreate2(self._id, path.encode(), t, space, lcpl, dcpl, 0)
        L.H5Pclose(lcpl)
        L.H5Pclose(dcpl)
        L.H5Sclose(space)
        if close:
            L.H5Tclose(t)
        if did < 0:
            raise OSError("cannot create dataset %s" % path)
        ds = Dataset(did, path)
        self._open_objs.append(ds)
        if data is not None and data.size:
            ds[...] = data
        return ds

    def flush(self):
        _load().H5Fflush(self._id, 1)

    def close(self):
        L = _load()
        for o in self._open_objs:
            if isinstance(o, Dataset):
                o._close()
            elif o._id >= 0:
                L.H5Oclose(o._id)
                o._id = -1
        self._open_objs = []
        if self._id >= 0:
            L.H5Fclose(self._id)
            self._id = -1

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def open(path, mode="r"):     # noqa: A001  (same entry as minizarr.open / zarr.open)
    return File(path, mode)
