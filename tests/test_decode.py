"""ppp+dec decode step (CPU): decoder shapes of the shipped flylight config and the batched
gather / scatter against a literal restatement of the reference's per-voxel loop
(setup01/decode.py:43-65)."""
import numpy as np
import pytest
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


def _reference_named_state(d, prefix="decoder."):
    """PatchDecoder weights under the key names of the reference's modules (funlib ConvPass:
    ``conv_pass.<n>``; Upsample resize_conv: one conv, here called ``up.conv_pass.0``) plus the
    encoder half and unrelated U-Net parameters a real checkpoint carries."""
    out = {"unet.l_conv.0.conv_pass.0.weight": torch.zeros(3), prefix + "to_code.conv_pass.0.weight": torch.zeros(2),
           prefix + "down_conv.0.conv_pass.0.bias": torch.zeros(2)}
    for k, v in d.state_dict().items():
        p = k.split(".")
        if p[0] == "from_code":
            out[prefix + "from_code.conv_pass.%s.%s" % (p[1], p[2])] = v
        elif p[0] == "up_conv":
            out[prefix + "up_conv.%s.conv_pass.%s.%s" % (p[1], p[2], p[3])] = v
        else:
            out[prefix + "up.%s.up.conv_pass.0.%s" % (p[1], p[-1])] = v
    return out


def test_decoder_weights_are_mapped_completely_or_not_at_all():
    """strict=False swallowed every key mismatch.  The explicit mapping loads a
    reference-named state dict exactly, and refuses missing / unexpected / mis-shaped keys."""
    import pytest
    torch.manual_seed(3)
    src, dst = dec.PatchDecoder(dict(AE)), dec.PatchDecoder(dict(AE))
    state = _reference_named_state(src)
    dst.load_state_dict(dec.map_decoder_state(state, dst), strict=True)
    x = torch.randn(3, 176)
    assert torch.equal(src(x), dst(x))
    missing = {k: v for k, v in state.items() if "from_code.conv_pass.0.bias" not in k}
    with pytest.raises(KeyError):
        dec.map_decoder_state(missing, dst)
    extra = dict(state)
    extra["decoder.up_conv.0.conv_pass.9.weight"] = torch.zeros(1)
    with pytest.raises(KeyError):
        dec.map_decoder_state(extra, dst)
    bad = dict(state)
    bad["decoder.from_code.conv_pass.0.weight"] = torch.zeros(2, 2)
    with pytest.raises(ValueError):
        dec.map_decoder_state(bad, dst)
    with pytest.raises(KeyError):
        dec.map_decoder_state({"unet.x": torch.zeros(1)}, dst)


@pytest.mark.gpu
def test_decode_then_vote_without_leaving_hbm():
    """BASELINE config [4] (ppp+dec): the decoded (C, Z, Y, X) block stays a device tensor and
    goes straight into to_instance_seg; the result equals voting on a host copy of the same
    decoded block with the CPU oracle.  (Decoder ARITHMETIC is unpinned -- no checkpoint, no
    funlib -- so the block decoded on the GPU is the common input of both sides.)"""
    from oracle import ppp_oracle as orc
    from patchperpix_amd.flags import FLYLIGHT
    from patchperpix_amd.vote_instances import vote_instances as vi
    torch.manual_seed(5)
    rng = np.random.default_rng(5)
    ae = dict(AE, input_shape_squeezed=(5, 5, 5))
    d = dec.PatchDecoder(ae).eval()
    with torch.no_grad():                       # spread the logits so that both classes occur
        d.up_conv[-1][-1].weight.mul_(40.0)
    shape = (14, 16, 18)
    code = rng.normal(size=(176,) + shape).astype(np.float32)
    numinst = np.zeros((3,) + shape, dtype=np.float32)
    numinst[0] = rng.uniform(size=shape) * 0.2            # P(0 instances) < 0.1 -> foreground
    fg = dec.foreground_from_numinst(numinst, 0.5)
    logits = dec.decode_volume(d, code, fg, batch_size=512, device="cuda", out_dtype=torch.float32)
    assert logits.is_cuda and tuple(logits.shape) == (125,) + shape
    sel = torch.as_tensor(fg, device="cuda").bool().expand_as(logits)
    logits = logits - logits[sel].median()      # (random weights: centre the logits so that both classes occur)
    pred = torch.sigmoid(logits)                # loadAffinities applies expit to logits (:249-250)
    pred = pred * torch.as_tensor(fg, device="cuda").float()   # nothing decoded outside the foreground
    pred16 = pred.to(torch.float16)             # decode writes float16 (decode.py:104-109)
    kw = dict(FLYLIGHT, overlapping_inst=False)
    fgb = fg.astype(bool)
    inst, _ = vi.to_instance_seg(pred16, fgb.copy(), fgb.copy(), fgb.astype(np.uint8), [5, 5, 5], **kw)
    ref = orc.to_instance_seg(pred16.float().cpu().numpy(), fgb, fgb.copy(), fgb.astype(np.uint8),
                              [5, 5, 5], **kw)
    assert np.array_equal(inst, ref["instances"])
    assert (pred16 > 0.5).any() and (pred16 < 0.5).any()
