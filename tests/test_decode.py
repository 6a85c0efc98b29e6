"""ppp+dec decode step (CPU): decoder shapes of the shipped flylight config and the batched
gather / scatter against a literal restatement of the reference's per-voxel loop
(setup01/decode.py:43-65)."""
import numpy as np
import torch

from patchperpix_amd import decode as dec

AE = dict(activation="relu", num_fmaps=[8, 16], downsample_factors=[[2, 2, 2], [2, 2, 2]],
          upsampling="resize_conv", kernel_size=3, num_repetitions=2, padding="same",
          code_fmaps=22, code_units=176, input_shape_squeezed=(7, 7, 7))


def test_decoder_shapes_flylight():
    torch.manual_seed(0)
    d = dec.PatchDecoder(dict(AE))
    assert d.code_shape == (-1, 22, 2, 2, 2)         # 176 units -> 22 x 2^3
    out = d(torch.randn(5, 176))
    assert out.shape == (5, 1, 7, 7, 7)              # 2^3 -> 4^3 -> 8^3 -> crop 7^3
    assert isinstance(d.up_conv[-1][-1], torch.nn.Conv3d)   # last pass: no activation (logits)


def test_decode_volume_equals_reference_loop():
    torch.manual_seed(1)
    rng = np.random.default_rng(0)
    d = dec.PatchDecoder(dict(AE)).eval()
    shape = (4, 5, 6)
    code = rng.normal(size=(176,) + shape).astype(np.float32)
    numinst = rng.uniform(size=(3,) + shape).astype(np.float32)
    fg = dec.foreground_from_numinst(numinst, 0.5)
    got = dec.decode_volume(d, code, fg, batch_size=7, device="cpu").numpy()
    # literal loop of the reference
    want = np.zeros((343,) + shape, dtype=np.float32)
    with torch.no_grad():
        for c in np.transpose(np.nonzero(fg)):
            v = code[(slice(None),) + tuple(c)].reshape(1, 176)
            p = d(torch.as_tensor(v)).numpy()
            want[(slice(None),) + tuple(c)] = p.reshape(-1)
    assert np.allclose(got, want, rtol=1e-5, atol=1e-6)    # batched vs single-sample convolutions
    assert not got[:, fg == 0].any()
