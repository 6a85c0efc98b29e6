"""A small zarr (format 2, directory store) reader / writer that needs neither ``zarr`` nor
``numcodecs`` -- the two packages the reference reads its predictions with and that this image
lacks (reference: experiments/flylight/setups/setup01/predict_no_gp.py:243-257 writes
``volumes/pred_affs`` as float16, chunks ``[C, o/2, o/2, o/2]``, ``Blosc(cname='zstd', clevel=3,
shuffle=Blosc.BITSHUFFLE)``; PatchPerPix/vote_instances/stitch_patch_graph.py:36 uses the same
compressor for its block graphs; utilVoteInstances.py:136-322 and io_hdflike.py read them).

Supported: C-order arrays of any fixed-size dtype; compressors ``null``, ``zlib``, ``gzip`` and ``blosc``
(Blosc-1 frames with the zstd, lz4 / lz4hc or zlib codec, no / byte / bit shuffle, split or
unsplit blocks, memcpy'ed frames); nested groups; ``.zattrs``; both chunk-key separators.
Decompression calls the system ``libzstd`` / ``liblz4`` through ctypes (they release the GIL, so
chunks are decoded by a thread pool) and zlib from the standard library.  Writing produces
Blosc-1 frames a stock numcodecs decodes (one stream per block, "do not split" flag set).

PINNING: the reference tree holds ONE store written by a real zarr / numcodecs -- the example
experiments/flylight/JRC_SS05008-20160318_24_B2_crop.zip (gzip chunks); its metadata and chunks are
a data fixture (tests/golden/ref_zarr_fixture, tests/test_minizarr.py::test_reference_example_store)
and pin the metadata / chunk-grid / edge-chunk / gzip side of this reader.  There is no Blosc
implementation in this container to produce reference frames, and the tree holds no Blosc data.  The frame layout follows c-blosc's README_HEADER.rst
(format version 2) and bitshuffle's element/bit order as documented there; the tests round-trip
writer -> reader and decode hand-assembled frames (split streams, memcpy'ed, byte shuffle).
Parity against files written by a real numcodecs is UNPINNED and says so in DESIGN.md.
"""
import ctypes
import ctypes.util
import json
import os
import struct
import zlib
from concurrent.futures import ThreadPoolExecutor

import builtins

import numpy as np

_fopen = builtins.open        # (this module defines its own ``open``, like zarr)

BLOSC_VERSION_FORMAT = 2
BLOSC_DOSHUFFLE, BLOSC_MEMCPYED, BLOSC_DOBITSHUFFLE, BLOSC_DONT_SPLIT = 0x1, 0x2, 0x4, 0x10
BLOSC_CODECS = {0: "blosclz", 1: "lz4", 2: "snappy", 3: "zlib", 4: "zstd"}
BLOSC_MAX_SPLITS, BLOSC_MIN_BUFFERSIZE = 16, 128

_ZSTD = _LZ4 = None


def _zstd():
    global _ZSTD
    if _ZSTD is None:
        name = ctypes.util.find_library("zstd") or "libzstd.so.1"
        L = ctypes.CDLL(name)
        L.ZSTD_decompress.restype = ctypes.c_size_t
        L.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        L.ZSTD_compress.restype = ctypes.c_size_t
        L.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                    ctypes.c_int]
        L.ZSTD_compressBound.restype = ctypes.c_size_t
        L.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
        L.ZSTD_isError.restype = ctypes.c_uint
        L.ZSTD_isError.argtypes = [ctypes.c_size_t]
        _ZSTD = L
    return _ZSTD


def _lz4():
    global _LZ4
    if _LZ4 is None:
        name = ctypes.util.find_library("lz4") or "liblz4.so.1"
        L = ctypes.CDLL(name)
        L.LZ4_decompress_safe.restype = ctypes.c_int
        L.LZ4_decompress_safe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        _LZ4 = L
    return _LZ4


def _addr(buf, offset=0):
    return ctypes.c_void_p(np.frombuffer(buf, dtype=np.uint8).ctypes.data + offset) \
        if not isinstance(buf, np.ndarray) else ctypes.c_void_p(buf.ctypes.data + offset)


# ---------------------------------------------------------------------------------------------
# shuffles (Blosc applies them per block)
# ---------------------------------------------------------------------------------------------
def byte_unshuffle(block, typesize):
    """Inverse of Blosc's byte shuffle: the block holds typesize runs of n bytes (byte j of every
    element), followed by the leftover bytes that do not fill an element."""
    n = len(block) // typesize
    body = block[:n * typesize].reshape(typesize, n).T
    return np.concatenate([np.ascontiguousarray(body).reshape(-1), block[n * typesize:]])


def byte_shuffle(block, typesize):
    n = len(block) // typesize
    body = block[:n * typesize].reshape(n, typesize).T
    return np.concatenate([np.ascontiguousarray(body).reshape(-1), block[n * typesize:]])


def bit_unshuffle(block, typesize):
    """Inverse of Blosc's bit shuffle (bitshuffle's bshuf_trans_bit_elem on the largest multiple
    of 8 elements, the rest copied): the shuffled block is 8*typesize bit planes of n/8 bytes,
    plane (byte j, bit k) first by j then by k, element e at bit e % 8 of byte e // 8."""
    n = (len(block) // typesize) // 8 * 8
    if n == 0:
        return block.copy()
    planes = np.unpackbits(block[:n * typesize].reshape(8 * typesize, n // 8), axis=1, bitorder="little")
    body = np.packbits(np.ascontiguousarray(planes.T), axis=1, bitorder="little")     # (n, typesize)
    return np.concatenate([body.reshape(-1), block[n * typesize:]])


def bit_shuffle(block, typesize):
    n = (len(block) // typesize) // 8 * 8
    if n == 0:
        return block.copy()
    bits = np.unpackbits(block[:n * typesize].reshape(n, typesize), axis=1, bitorder="little")  # (n, 8 ts)
    body = np.packbits(np.ascontiguousarray(bits.T), axis=1, bitorder="little")      # (8 ts, n / 8)
    return np.concatenate([body.reshape(-1), block[n * typesize:]])


# ---------------------------------------------------------------------------------------------
# Blosc-1 frames
# ---------------------------------------------------------------------------------------------
def blosc_decode(frame, unshuffle=True):
    """Blosc-1 frame (bytes) -> uint8 array of the decompressed chunk.  With unshuffle=False the
    blocks are returned still shuffled, together with (typesize, blocksize, shuffle kind) -- the
    device path un-shuffles on the GPU."""
    buf = np.frombuffer(frame, dtype=np.uint8)
    if len(buf) < 16:
        raise ValueError("not a Blosc frame")
    version, _versionlz, flags, typesize = (int(v) for v in buf[:4])
    nbytes, blocksize, cbytes = struct.unpack_from("<III", frame, 4)
    if version != BLOSC_VERSION_FORMAT:
        raise ValueError("Blosc format version %d is not supported" % version)
    if cbytes > len(buf):
        raise ValueError("truncated Blosc frame")
    shuffle = "bit" if flags & BLOSC_DOBITSHUFFLE else ("byte" if flags & BLOSC_DOSHUFFLE else None)
    out = np.empty(nbytes, dtype=np.uint8)
    if nbytes == 0:
        return out if unshuffle else (out, typesize, blocksize, None)
    if flags & BLOSC_MEMCPYED:
        out[:] = buf[16:16 + nbytes]
        return out if unshuffle else (out, typesize, blocksize, None)
    codec = BLOSC_CODECS.get(flags >> 5)
    nblocks = (nbytes + blocksize - 1) // blocksize
    bstarts = struct.unpack_from("<%di" % nblocks, frame, 16)
    for b in range(nblocks):
        bsize = min(blocksize, nbytes - b * blocksize)
        leftover = bsize != blocksize
        split = not (flags & BLOSC_DONT_SPLIT) and typesize <= BLOSC_MAX_SPLITS and \
            blocksize // typesize >= BLOSC_MIN_BUFFERSIZE and not leftover
        nstreams = typesize if split else 1
        neblock = bsize // nstreams
        pos = bstarts[b]
        dst = b * blocksize
        for _ in range(nstreams):
            (cs,) = struct.unpack_from("<i", frame, pos)
            pos += 4
            if cs == neblock:
                out[dst:dst + neblock] = buf[pos:pos + cs]
            else:
                _codec_decode(codec, buf, pos, cs, out, dst, neblock)
            pos += cs
            dst += neblock
        # c-blosc applies the BIT shuffle whenever its flag is set (also to 1-byte types: masks,
        # numinst, raw); only the BYTE shuffle is a no-op for typesize 1
        if unshuffle and (shuffle == "bit" or (shuffle == "byte" and typesize > 1)):
            blk = out[b * blocksize:b * blocksize + bsize]
            blk[:] = bit_unshuffle(blk, typesize) if shuffle == "bit" else byte_unshuffle(blk, typesize)
    if unshuffle:
        return out
    return out, typesize, blocksize, (shuffle if (shuffle == "bit" or typesize > 1) else None)


def _codec_decode(codec, src, pos, csize, out, dst, nout):
    if codec == "zstd":
        L = _zstd()
        r = L.ZSTD_decompress(_addr(out, dst), nout, _addr(src, pos), csize)
        if L.ZSTD_isError(r) or r != nout:
            raise ValueError("zstd stream does not decode to %d bytes" % nout)
    elif codec == "lz4":
        r = _lz4().LZ4_decompress_safe(_addr(src, pos), _addr(out, dst), csize, nout)
        if r != nout:
            raise ValueError("lz4 stream does not decode to %d bytes" % nout)
    elif codec == "zlib":
        data = zlib.decompress(src[pos:pos + csize].tobytes())
        if len(data) != nout:
            raise ValueError("zlib stream does not decode to %d bytes" % nout)
        out[dst:dst + nout] = np.frombuffer(data, dtype=np.uint8)
    else:
        raise NotImplementedError("Blosc codec %r is not supported" % codec)


def blosc_encode(data, typesize, cname="zstd", clevel=3, shuffle="bit", blocksize=None):
    """uint8 array -> Blosc-1 frame (bytes): one stream per block (flag "do not split")."""
    data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
    nbytes = len(data)
    if cname not in ("zstd", "zlib"):
        # (frames of the other codecs are READ -- blosc_decode -- but the writers of the reference only
        # ever ask for zstd, stitch_patch_graph.py:36)
        raise NotImplementedError("minizarr writes Blosc frames with zstd or zlib, not %r" % (cname,))
    codec_id = {"zstd": 4, "zlib": 3}[cname]
    if blocksize is None:
        blocksize = max(typesize, min(nbytes, 1 << 18) // typesize * typesize) if nbytes else typesize
    nblocks = (nbytes + blocksize - 1) // blocksize if nbytes else 0
    flags = BLOSC_DONT_SPLIT | (codec_id << 5)
    if shuffle == "bit":
        flags |= BLOSC_DOBITSHUFFLE          # (also for typesize 1, like c-blosc)
    elif typesize > 1 and shuffle == "byte":
        flags |= BLOSC_DOSHUFFLE
    streams = []
    for b in range(nblocks):
        blk = data[b * blocksize:(b + 1) * blocksize]
        if flags & BLOSC_DOBITSHUFFLE:
            blk = bit_shuffle(blk, typesize)
        elif flags & BLOSC_DOSHUFFLE:
            blk = byte_shuffle(blk, typesize)
        blk = np.ascontiguousarray(blk)
        if cname == "zstd":
            L = _zstd()
            cap = L.ZSTD_compressBound(len(blk))
            dst = np.empty(cap, dtype=np.uint8)
            r = L.ZSTD_compress(_addr(dst), cap, _addr(blk), len(blk), int(clevel))
            if L.ZSTD_isError(r):
                raise RuntimeError("ZSTD_compress failed")
            comp = dst[:r].tobytes()
        else:
            comp = zlib.compress(blk.tobytes(), int(clevel))
        if len(comp) >= len(blk):
            comp = blk.tobytes()          # stored: compressed size == block size
        streams.append(comp)
    header = 16 + 4 * nblocks
    bstarts, pos = [], header
    for c in streams:
        bstarts.append(pos)
        pos += 4 + len(c)
    out = bytearray()
    out += bytes([BLOSC_VERSION_FORMAT, 1, flags, typesize])
    out += struct.pack("<III", nbytes, blocksize, pos)
    out += struct.pack("<%di" % nblocks, *bstarts)
    for c in streams:
        out += struct.pack("<i", len(c)) + c
    return bytes(out)


# ---------------------------------------------------------------------------------------------
# arrays and groups
# ---------------------------------------------------------------------------------------------
class Attrs(dict):
    def __init__(self, path, writable):
        super().__init__()
        self._file, self._writable = os.path.join(path, ".zattrs"), writable
        if os.path.exists(self._file):
            with _fopen(self._file) as f:
                super().update(json.load(f))

    def __setitem__(self, k, v):
        if not self._writable:
            raise PermissionError("read-only zarr")
        super().__setitem__(k, v.tolist() if isinstance(v, np.ndarray) else
                            (list(v) if isinstance(v, tuple) else v))
        with _fopen(self._file, "w") as f:
            json.dump(dict(self), f, indent=4)


class Array:
    """One zarr array: NumPy-style basic slicing for reading and writing."""

    def __init__(self, path, writable=False):
        self.path, self._writable = path, writable
        with _fopen(os.path.join(path, ".zarray")) as f:
            m = json.load(f)
        if m.get("zarr_format") != 2:
            raise NotImplementedError("zarr format %r" % m.get("zarr_format"))
        if m.get("order", "C") != "C":
            raise NotImplementedError("Fortran-order zarr arrays")
        if m.get("filters"):
            raise NotImplementedError("zarr filters %r" % m.get("filters"))
        self.meta = m
        self.shape = tuple(int(s) for s in m["shape"])
        self.chunks = tuple(int(c) for c in m["chunks"])
        self.dtype = np.dtype(m["dtype"])
        self.fill_value = m.get("fill_value") or 0
        self.compressor = m.get("compressor")
        self.sep = m.get("dimension_separator", ".")
        self.attrs = Attrs(path, writable)
        self.ndim = len(self.shape)

    def __len__(self):
        return self.shape[0]

    # ---- chunk level
    def chunk_file(self, idx):
        return os.path.join(self.path, self.sep.join(str(int(i)) for i in idx))

    def read_chunk_bytes(self, idx):
        fn = self.chunk_file(idx)
        if not os.path.exists(fn):
            return None
        with _fopen(fn, "rb") as f:
            return f.read()

    def decode_chunk(self, raw):
        """bytes of a chunk file -> ndarray of the full chunk shape."""
        c = self.compressor
        if c is None:
            data = np.frombuffer(raw, dtype=np.uint8)
        elif c["id"] == "blosc":
            data = blosc_decode(raw)
        elif c["id"] == "zlib":
            data = np.frombuffer(zlib.decompress(raw), dtype=np.uint8)
        elif c["id"] == "gzip":
            # numcodecs.GZip (the compressor of the reference's own example store,
            # experiments/flylight/JRC_SS05008-20160318_24_B2_crop.zip): a gzip member per chunk
            data = np.frombuffer(zlib.decompress(raw, 16 + zlib.MAX_WBITS), dtype=np.uint8)
        else:
            raise NotImplementedError("zarr compressor %r" % c["id"])
        return data.view(self.dtype).reshape(self.chunks)

    def encode_chunk(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        c = self.compressor
        if c is None:
            return arr.tobytes()
        if c["id"] == "blosc":
            # (-1 = AUTOSHUFFLE: numcodecs takes the bit shuffle for 1-byte types, the byte shuffle otherwise)
            shuffle = {0: None, 1: "byte", 2: "bit", -1: "bit" if self.dtype.itemsize == 1 else "byte"}[
                int(c.get("shuffle", 1))]
            return blosc_encode(arr.view(np.uint8).reshape(-1), self.dtype.itemsize,
                                cname=c.get("cname", "zstd"), clevel=c.get("clevel", 3), shuffle=shuffle,
                                blocksize=c.get("blocksize") or None)
        if c["id"] == "zlib":
            return zlib.compress(arr.tobytes(), int(c.get("level", 1)))
        if c["id"] == "gzip":
            import gzip
            return gzip.compress(arr.tobytes(), int(c.get("level", 1)), mtime=0)   # (numcodecs writes mtime 0 too)
        raise NotImplementedError("zarr compressor %r" % c["id"])

    # ---- selections
    def _normalise(self, sel):
        if not isinstance(sel, tuple):
            sel = (sel,)
        if any(s is Ellipsis for s in sel):
            i = [k for k, s in enumerate(sel) if s is Ellipsis][0]
            sel = sel[:i] + (slice(None),) * (self.ndim - len(sel) + 1) + sel[i + 1:]
        sel = sel + (slice(None),) * (self.ndim - len(sel))
        ranges, squeeze = [], []
        for ax, s in enumerate(sel):
            if isinstance(s, (int, np.integer)):
                s = int(s) + (self.shape[ax] if s < 0 else 0)
                ranges.append((s, s + 1))
                squeeze.append(ax)
            elif isinstance(s, slice):
                start, stop, step = s.indices(self.shape[ax])
                if step != 1:
                    raise NotImplementedError("strided zarr selections")
                ranges.append((start, max(start, stop)))
            else:
                raise NotImplementedError("zarr selection %r" % (s,))
        return ranges, tuple(squeeze)

    def _chunks_of(self, ranges):
        grids = [range(a // c, (b - 1) // c + 1) if b > a else range(0) for (a, b), c in zip(ranges, self.chunks)]
        return [idx for idx in np.ndindex(*[len(g) for g in grids])], grids

    def __getitem__(self, sel):
        ranges, squeeze = self._normalise(sel)
        out = np.full([b - a for a, b in ranges], self.fill_value, dtype=self.dtype)
        self._read(ranges, out)
        return np.squeeze(out, axis=squeeze) if squeeze else out

    def read_into(self, sel, out):
        """The selection decoded chunk by chunk straight into `out` (an array of the selection's
        un-squeezed shape and this array's dtype -- e.g. the NumPy view of a pinned host buffer
        that is then copied to the device: no intermediate array of the whole selection)."""
        ranges, _ = self._normalise(sel)
        if tuple(out.shape) != tuple(b - a for a, b in ranges) or out.dtype != self.dtype:
            raise ValueError("read_into: `out` must have the selection's shape and the array's dtype")
        out[...] = self.fill_value
        self._read(ranges, out)
        return out

    def _read(self, ranges, out):
        pos, grids = self._chunks_of(ranges)

        def work(p):
            idx = tuple(g[i] for g, i in zip(grids, p))
            raw = self.read_chunk_bytes(idx)
            if raw is None:
                return
            chunk = self.decode_chunk(raw)
            src, dst = [], []
            for ax, (a, b) in enumerate(ranges):
                lo, hi = idx[ax] * self.chunks[ax], (idx[ax] + 1) * self.chunks[ax]
                s0, s1 = max(a, lo), min(b, hi)
                src.append(slice(s0 - lo, s1 - lo))
                dst.append(slice(s0 - a, s1 - a))
            out[tuple(dst)] = chunk[tuple(src)]

        if len(pos) > 1:
            with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as pool:
                list(pool.map(work, pos))
        else:
            for p in pos:
                work(p)

    def __array__(self, dtype=None, copy=None):
        a = self[...]
        return a if dtype is None else a.astype(dtype)

    def __setitem__(self, sel, value):
        if not self._writable:
            raise PermissionError("read-only zarr")
        ranges, squeeze = self._normalise(sel)
        value = np.asarray(value, dtype=self.dtype)
        value = np.broadcast_to(np.expand_dims(value, squeeze) if squeeze and value.ndim < self.ndim else value,
                                [b - a for a, b in ranges])
        pos, grids = self._chunks_of(ranges)
        for p in pos:
            idx = tuple(g[i] for g, i in zip(grids, p))
            src, dst, full = [], [], True
            for ax, (a, b) in enumerate(ranges):
                lo, hi = idx[ax] * self.chunks[ax], (idx[ax] + 1) * self.chunks[ax]
                s0, s1 = max(a, lo), min(b, hi)
                dst.append(slice(s0 - lo, s1 - lo))
                src.append(slice(s0 - a, s1 - a))
                full = full and s0 == lo and (s1 == hi or s1 == self.shape[ax])
            if full:
                chunk = np.full(self.chunks, self.fill_value, dtype=self.dtype)
            else:
                raw = self.read_chunk_bytes(idx)
                chunk = self.decode_chunk(raw).copy() if raw is not None else \
                    np.full(self.chunks, self.fill_value, dtype=self.dtype)
            chunk[tuple(dst)] = value[tuple(src)]
            fn = self.chunk_file(idx)
            os.makedirs(os.path.dirname(fn), exist_ok=True)
            with _fopen(fn, "wb") as f:
                f.write(self.encode_chunk(chunk))


class Group:
    def __init__(self, path, writable=False):
        self.path, self._writable = path, writable
        self.attrs = Attrs(path, writable)

    def keys(self):
        return sorted(n for n in os.listdir(self.path)
                      if os.path.isdir(os.path.join(self.path, n)) and
                      (os.path.exists(os.path.join(self.path, n, ".zarray")) or
                       os.path.exists(os.path.join(self.path, n, ".zgroup"))))

    def __contains__(self, key):
        p = os.path.join(self.path, key)
        return os.path.exists(os.path.join(p, ".zarray")) or os.path.exists(os.path.join(p, ".zgroup"))

    def __getitem__(self, key):
        p = os.path.join(self.path, key)
        if os.path.exists(os.path.join(p, ".zarray")):
            return Array(p, self._writable)
        if os.path.exists(os.path.join(p, ".zgroup")):
            return Group(p, self._writable)
        raise KeyError(key)

    def require_group(self, key):
        p = self.path
        for part in key.strip("/").split("/"):
            p = os.path.join(p, part)
            os.makedirs(p, exist_ok=True)
            if not os.path.exists(os.path.join(p, ".zgroup")) and not os.path.exists(os.path.join(p, ".zarray")):
                with _fopen(os.path.join(p, ".zgroup"), "w") as f:
                    json.dump({"zarr_format": 2}, f)
        return Group(p, self._writable)

    def create(self, key, shape, chunks, dtype, compressor="default", fill_value=0, overwrite=False):
        """zarr.Group.create: ``compressor`` is a numcodecs-style config dict, None, or "default"
        = the reference's Blosc(cname='zstd', clevel=3, shuffle=BITSHUFFLE)."""
        if not self._writable:
            raise PermissionError("read-only zarr")
        parts = key.strip("/").split("/")
        parent = self.require_group("/".join(parts[:-1])) if len(parts) > 1 else self
        p = os.path.join(parent.path, parts[-1])
        if os.path.exists(os.path.join(p, ".zarray")) and not overwrite:
            raise ValueError("array %s exists" % key)
        os.makedirs(p, exist_ok=True)
        if compressor == "default":
            compressor = {"id": "blosc", "cname": "zstd", "clevel": 3, "shuffle": 2, "blocksize": 0}
        dt = np.dtype(dtype)
        meta = {"chunks": [int(c) for c in chunks], "compressor": compressor,
                "dtype": dt.str if dt.itemsize > 1 else "|" + dt.str[1:], "fill_value": fill_value,
                "filters": None, "order": "C", "shape": [int(s) for s in shape], "zarr_format": 2}
        with _fopen(os.path.join(p, ".zarray"), "w") as f:
            json.dump(meta, f, indent=4)
        return Array(p, True)

    def create_dataset(self, key, data=None, shape=None, chunks=None, dtype=None, compression=None,
                       compressor="default", **_ignored):
        """h5py / zarr style convenience: whole-array write."""
        data = None if data is None else np.asarray(data)
        shape = shape if shape is not None else data.shape
        dtype = dtype if dtype is not None else data.dtype
        if chunks is None:
            chunks = tuple(min(int(s), 64) for s in shape) if len(shape) else ()
        a = self.create(key, shape, chunks, dtype, compressor=compressor, overwrite=True)
        if data is not None:
            a[...] = data
        return a


def open(path, mode="r"):     # noqa: A001 - mirrors zarr.open
    """zarr.open for a directory store: returns the root Group (or Array)."""
    writable = mode in ("w", "a", "r+", "w-")
    if mode == "w" and os.path.isdir(path):
        import shutil
        shutil.rmtree(path)
    if writable and not os.path.exists(path):
        os.makedirs(path)
        with _fopen(os.path.join(path, ".zgroup"), "w") as f:
            json.dump({"zarr_format": 2}, f)
    if os.path.exists(os.path.join(path, ".zarray")):
        return Array(path, writable)
    if not os.path.exists(os.path.join(path, ".zgroup")):
        if writable:
            with _fopen(os.path.join(path, ".zgroup"), "w") as f:
                json.dump({"zarr_format": 2}, f)
        else:
            raise FileNotFoundError("%s is not a zarr directory store" % path)
    return Group(path, writable)

