#!/usr/bin/env python3
"""The only bytes in the reference's tree that a real zarr / numcodecs wrote: the example store
experiments/flylight/JRC_SS05008-20160318_24_B2_crop.zip (zarr v2, gzip level 1 chunks).  This
script (development container only: it reads /root/reference) copies DATA out of it -- the group
and array metadata, both chunks of `volumes/gt_instances` (|u1, 3 x 50^3) and the two chunks of the
last channel of `volumes/raw` (<u2; chunks 1.0.0.0 and 1.1.0.0; the first channel pair stays out:
243 KB) -- into tests/golden/ref_zarr_fixture/crop.zarr, and records shape / dtype / CRC-32 of the
arrays as decoded by Python's own gzip module (independent of patchperpix_amd.minizarr) in
expected.json.  tests/test_minizarr.py::test_reference_example_store opens the copy with minizarr.
"""
import gzip
import json
import os
import zipfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/experiments/flylight/JRC_SS05008-20160318_24_B2_crop.zip"
ROOT = "JRC_SS05008-20160318_24_B2_crop.zarr/"
KEEP = [".zgroup", "volumes/.zgroup", "volumes/gt_instances/.zarray", "volumes/gt_instances/0.0.0.0",
        "volumes/gt_instances/1.0.0.0", "volumes/raw/.zarray", "volumes/raw/1.0.0.0", "volumes/raw/1.1.0.0"]


def main():
    out = os.path.join(HERE, "ref_zarr_fixture", "crop.zarr")
    z = zipfile.ZipFile(SRC)
    for name in KEEP:
        fn = os.path.join(out, name)
        os.makedirs(os.path.dirname(fn), exist_ok=True)
        with open(fn, "wb") as f:
            f.write(z.read(ROOT + name))
    expected = {}
    for key in ("volumes/gt_instances", "volumes/raw"):
        meta = json.loads(z.read(ROOT + key + "/.zarray"))
        shape, chunks, dt = meta["shape"], meta["chunks"], np.dtype(meta["dtype"])
        full = np.full(shape, meta["fill_value"], dtype=dt)
        grid = [-(-s // c) for s, c in zip(shape, chunks)]
        for idx in np.ndindex(*grid):
            name = key + "/" + ".".join(str(i) for i in idx)
            if name not in KEEP:
                continue                                    # (left out of the fixture: reads as fill value)
            blk = np.frombuffer(gzip.decompress(z.read(ROOT + name)), dtype=dt).reshape(chunks)
            sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
            full[sel] = blk[tuple(slice(0, s.stop - s.start) for s in sel)]
        expected[key] = dict(shape=shape, dtype=dt.str, chunks=chunks, compressor=meta["compressor"],
                             crc32=zlib.crc32(np.ascontiguousarray(full).tobytes()),
                             nonzero=int(np.count_nonzero(full)), max=int(full.max()),
                             sum=int(full.astype(np.int64).sum()))
    with open(os.path.join(HERE, "ref_zarr_fixture", "expected.json"), "w") as f:
        json.dump(expected, f, indent=1, sort_keys=True)
    print(json.dumps(expected, indent=1))


if __name__ == "__main__":
    main()
