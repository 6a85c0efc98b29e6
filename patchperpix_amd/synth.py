"""Synthetic PatchPerPix prediction volumes (host / NumPy side).

The generator follows the recipe fixed in SURVEY.md section 8(d): a label volume of
touching blob instances, and a dense patch prediction

    pred[r][v] = HI if lab[v + r - rad] == lab[v] != 0 else LO,   plus noise,

rounded through float16 (the on-disk dtype of ``volumes/pred_affs``, reference
``experiments/flylight/setups/setup01/predict_no_gp.py:243-257``) and widened to
float32.  Noise is a counter-based integer hash of ``(seed, r, v)`` so the very same
values can be produced tile by tile on the device (``ppp_synth_pred`` in
``csrc/ppp_aux.hip`` uses the identical hash).

Used by the tests, the golden-vector generator and ``bench.py``; it is not part of
the hot path.
"""
import numpy as np

HI = 0.95
LO = 0.05
NOISE = 0.04


def hash_u32(x):
    """lowbias32-style avalanche hash on uint32 arrays (wraps mod 2**32)."""
    m = np.uint64(0xFFFFFFFF)
    x = np.asarray(x, dtype=np.uint64) & m  # counters wrap mod 2**32 first
    x = (x ^ (x >> np.uint64(16))) & m
    x = (x * np.uint64(0x7FEB352D)) & m
    x = (x ^ (x >> np.uint64(15))) & m
    x = (x * np.uint64(0x846CA68B)) & m
    x = (x ^ (x >> np.uint64(16))) & m
    return x.astype(np.uint32)


def cell_labels(shape, cell, seed=0, jitter=True):
    """Label volume: a jittered grid of box-ish blobs that touch their neighbours.

    Voxel v belongs to grid cell ``(v + shift(plane)) // cell``; the label is a hash of
    the cell index (never 0).  A sparse set of cells is dropped to background so that
    foreground is dense (>= 90 %) but not total.
    """
    shape = tuple(int(s) for s in shape)
    cell = np.broadcast_to(np.asarray(cell, dtype=np.int64), (3,))
    zz, yy, xx = np.meshgrid(*[np.arange(s, dtype=np.int64) for s in shape],
                             indexing="ij")
    cz = zz // cell[0]
    if jitter:
        # shift rows of cells by a per-slab offset so that borders are staggered
        sy = (hash_u32(cz.astype(np.uint64) * np.uint64(7919) + np.uint64(seed))
              % np.uint32(max(1, cell[1]))).astype(np.int64)
    else:
        sy = 0
    cy = (yy + sy) // cell[1]
    if jitter:
        sx = (hash_u32((cz * 131 + cy).astype(np.uint64) * np.uint64(104729)
                       + np.uint64(seed + 1))
              % np.uint32(max(1, cell[2]))).astype(np.int64)
    else:
        sx = 0
    cx = (xx + sx) // cell[2]
    key = ((cz * 1000003 + cy) * 1000003 + cx).astype(np.uint64) & np.uint64(0xFFFFFFFF)
    h = hash_u32(key + np.uint64(seed * 2654435761 % (1 << 32)))
    lab = (h % np.uint32(65000)).astype(np.int64) + 1
    # drop ~6 % of the cells to background
    lab[(h >> np.uint32(16)) % np.uint32(16) == 0] = 0
    return lab


def two_blobs(shape):
    """Two touching box instances (the SURVEY probe volume) inside a bg margin."""
    shape = tuple(int(s) for s in shape)
    lab = np.zeros(shape, dtype=np.int64)
    z0, z1 = (0, shape[0]) if shape[0] == 1 else (1, shape[0] - 1)
    ym = shape[1] // 2
    lab[z0:z1, 1:ym, 1:shape[2] - 1] = 1
    lab[z0:z1, ym:shape[1] - 1, 1:shape[2] - 1] = 2
    return lab


def pred_from_labels(lab, patchshape, seed=0, hi=HI, lo=LO, noise=NOISE,
                     dtype=np.float32):
    """Dense patch prediction ``(C, Z, Y, X)`` from a label volume (see module doc)."""
    lab = np.asarray(lab)
    Z, Y, X = lab.shape
    pz, py, px = [int(p) for p in patchshape]
    rz, ry, rx = pz // 2, py // 2, px // 2
    C = pz * py * px
    padded = np.full((Z + 2 * rz, Y + 2 * ry, X + 2 * rx), -1, dtype=np.int64)
    padded[rz:rz + Z, ry:ry + Y, rx:rx + X] = lab
    pred = np.empty((C, Z, Y, X), dtype=np.float32)
    lin = (np.arange(Z * Y * X, dtype=np.uint64)).reshape(Z, Y, X)
    r = 0
    for dz in range(pz):
        for dy in range(py):
            for dx in range(px):
                nb = padded[dz:dz + Z, dy:dy + Y, dx:dx + X]
                same = (nb == lab) & (lab != 0)
                base = np.where(same, np.float32(hi), np.float32(lo))
                h = hash_u32(lin * np.uint64(C) + np.uint64(r)
                             + np.uint64((seed * 0x9E3779B1) % (1 << 32)))
                # 24-bit uniform in [0, 1)
                u = (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / (1 << 24))
                val = base + np.float32(noise) * (np.float32(2.0) * u - np.float32(1.0))
                pred[r] = val.astype(np.float16).astype(np.float32)
                r += 1
    return pred.astype(dtype, copy=False)


def make_case(shape, patchshape, seed=0, kind="cells", cell=None, overlap_frac=0.0,
              noise=NOISE):
    """Returns dict(pred f32 (C,Z,Y,X), foreground bool, numinst u8, labels)."""
    if kind == "two_blobs":
        lab = two_blobs(shape)
    elif kind == "empty":
        lab = np.zeros(tuple(shape), dtype=np.int64)
    else:
        if cell is None:
            cell = [max(1, 3 * int(p)) if int(p) > 1 else 1 for p in patchshape]
        lab = cell_labels(shape, cell, seed=seed)
    pred = pred_from_labels(lab, patchshape, seed=seed, noise=noise)
    fg = lab != 0
    numinst = fg.astype(np.uint8)
    if overlap_frac > 0:
        lin = np.arange(lab.size, dtype=np.uint64).reshape(lab.shape)
        h = hash_u32(lin + np.uint64(seed + 77))
        ov = ((h % np.uint32(10000)) < np.uint32(int(overlap_frac * 10000))) & fg
        numinst[ov] = 2
    return dict(pred=pred, foreground=fg, numinst=numinst, labels=lab)
