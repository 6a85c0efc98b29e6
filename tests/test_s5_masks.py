"""The thinning decisions of the patch intersection made beforehand (ppp_patch_graph_lcg,
csrc/ppp_patch_graph_pa.hip) against a loop-for-loop restatement of computePatchGraph.cu:24-86:
the pair's generator advances on every combination of foreground pixels z1 of A, z2 of B that
both lie in the intersection of the two windows, in the loop order of the kernel, and the
combination is dropped when float(rnd) / 2^32 > 0.2.  The per-patch kernel's use of the masks is
covered end to end (bit-identical affinities, tests/test_gpu_parity.py)."""
import ctypes

import numpy as np
import pytest


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    from patchperpix_amd import backend
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    assert backend.device_count() >= 1
    return torch


def _restated_masks(pred, A, B, ps, th):
    """{(i1, plane of the intersection, chunk): mask of dropped candidates} for one pair."""
    pz, py, px = ps
    rad = (pz // 2, py // 2, px // 2)
    mid = (pz * py * px) // 2
    d = [b - a for a, b in zip(A, B)]
    n = [p - abs(x) for p, x in zip(ps, d)]
    lo1 = [max(x, 0) for x in d]
    lo2 = [max(-x, 0) for x in d]
    rpc = 64 // px
    nch = -(-px // rpc)

    def fg(c, r):
        z, y, x = (c[0] + r[0] - rad[0], c[1] + r[1] - rad[1], c[2] + r[2] - rad[2])
        ch = (r[0] * py + r[1]) * px + r[2]
        return pred[mid, z, y, x] > th and pred[ch, c[0], c[1], c[2]] > th

    rnd = 1
    for v in (A[0], B[0], A[1], B[1], A[2], B[2]):
        rnd = (rnd * v) & 0xFFFFFFFF
    out = {}
    for z1 in range(pz):
        for y1 in range(py):
            for x1 in range(px):
                r1 = (z1, y1, x1)
                if not fg(A, r1):
                    continue
                # pixel 1 inside the window of B
                if any(not (0 <= r1[k] - d[k] < ps[k]) for k in range(3)):
                    continue
                i1 = ((z1 - lo1[0]) * n[1] + (y1 - lo1[1])) * n[2] + (x1 - lo1[2])
                for z2 in range(lo2[0], lo2[0] + n[0]):
                    for c in range(nch):
                        if c * rpc < py:
                            out[(i1, z2 - lo2[0], c)] = 0
                for z2 in range(pz):
                    for y2 in range(py):
                        for x2 in range(px):
                            r2 = (z2, y2, x2)
                            if not fg(B, r2):
                                continue
                            # pixel 2 inside the window of A
                            if any(not (0 <= r2[k] + d[k] < ps[k]) for k in range(3)):
                                continue
                            rnd = (rnd * 1103515245) & 0xFFFFFFFF
                            if float(np.float32(np.uint32(rnd))) / 4294967296.0 > 0.2:
                                c = y2 // rpc
                                out[(i1, z2 - lo2[0], c)] |= 1 << ((y2 - c * rpc) * px + x2)
    return out, n, nch


@pytest.mark.gpu
@pytest.mark.parametrize("shape,ps,seed", [((12, 14, 16), (5, 5, 5), 1), ((16, 17, 18), (9, 9, 9), 2),
                                           ((7, 20, 21), (3, 7, 7), 3)])
def test_thinning_masks_equal_the_reference_loop(shape, ps, seed, torch_cuda):
    torch = torch_cuda
    from patchperpix_amd import backend
    rng = np.random.default_rng(seed)
    C = ps[0] * ps[1] * ps[2]
    pred = rng.random((C,) + shape, dtype=np.float32)
    pred[C // 2] = (rng.random(shape) < 0.7).astype(np.float32)      # the foreground channel
    P = backend.make_params(shape, ps, patch_threshold=0.5)
    rad = [p // 2 for p in ps]
    pairs = []
    while len(pairs) < 70:                                            # more than one wave
        A = [int(rng.integers(rad[k] + 1, shape[k] - rad[k])) for k in range(3)]
        d = [int(rng.integers(-(ps[k] - 1), ps[k])) for k in range(3)]
        B = [a + x for a, x in zip(A, d)]
        if all(rad[k] <= B[k] < shape[k] - rad[k] for k in range(3)) and any(d):
            pairs.append(A + B)
    rows = np.array(pairs, dtype=np.int32)
    d = rows[:, 3:] - rows[:, :3]
    words = backend.lcg_words(d[:, 0], d[:, 1], d[:, 2], P)
    assert np.all(words > 0)
    off = np.concatenate([[0], np.cumsum(words)[:-1]]).astype(np.int64)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    pred_d, rows_d = dev(pred), dev(rows)
    order_d = dev(np.arange(len(rows), dtype=np.int32))
    # lanes of a wave: rows of different offsets next to each other on purpose
    pos_d, off_d = dev(np.arange(len(rows), dtype=np.int64)), dev(off)
    drops = torch.zeros((int(words.sum()),), dtype=torch.int64, device="cuda")
    backend.check(backend.lib().ppp_patch_graph_lcg(
        backend._dev_ptr(pred_d), backend.pred_dtype_code(pred_d), backend._dev_ptr(rows_d),
        backend._dev_ptr(order_d), backend._dev_ptr(pos_d), len(rows), backend._dev_ptr(off_d),
        backend._dev_ptr(drops), ctypes.byref(P), backend._stream()))
    torch.cuda.synchronize()
    got = drops.cpu().numpy().view(np.uint64)
    n_checked = n_dropped = 0
    for k, row in enumerate(rows.tolist()):
        want, n, nch = _restated_masks(pred, row[:3], row[3:], ps, 0.5)
        for (i1, kz, c), m in want.items():
            g = int(got[off[k] + (i1 * n[0] + kz) * nch + c])
            assert g == m, (k, row, i1, kz, c, hex(g), hex(m))
            n_checked += 1
            n_dropped += bin(m).count("1")
    assert n_checked > 1000 and n_dropped > 1000
