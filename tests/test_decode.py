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


AE_SHIPPED = dict(activation="relu", num_fmaps=[64, 128], downsample_factors=[[2, 2, 2], [2, 2, 2]],
                  upsampling="resize_conv", kernel_size=3, num_repetitions=2, padding="same",
                  code_fmaps=22, code_units=176, input_shape_squeezed=(7, 7, 7))


def test_head_and_tail_compose_and_fused_tail_is_recognised():
    torch.manual_seed(2)
    d = dec.PatchDecoder(dict(AE_SHIPPED)).eval()
    x = torch.randn(3, 176)
    with torch.no_grad():
        assert torch.equal(d(x), d.tail(d.head(x)))
        assert tuple(d.head(x).shape) == (3, 64, 4, 4, 4)
    assert d.fused_tail_params() is not None                       # the shipped shape
    assert dec.PatchDecoder(dict(AE)).fused_tail_params() is None  # 8 feature maps: torch ops
    assert dec.PatchDecoder(dict(AE_SHIPPED, input_shape_squeezed=(5, 5, 5))).fused_tail_params() is None


@pytest.mark.parametrize("cfg", ["shipped", "small"])
def test_dense_head_equals_convolution_head(cfg):
    """enable_dense_head: every convolution of the head (the first with the upsampling in front
    of it) as x @ W + b, W made by pushing the identity through the layer.  Same weights, another
    summation order: |diff| <= 1e-5 * max|value| in float32."""
    torch.manual_seed(5)
    d = dec.PatchDecoder(dict(AE_SHIPPED if cfg == "shipped" else AE)).eval()
    x = torch.randn(9, d.code_units)
    with torch.no_grad():
        want = d.head(x)
        assert d.enable_dense_head()
        assert len(d._dense["stages"]) == 3 and d._dense["stages"][-1][0].shape[1] == int(np.prod(want.shape[1:]))
        got = d.head(x)
        assert got.shape == want.shape
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
        # the whole decoder through the dense head
        assert float((d(x) - d.tail(want)).abs().max()) <= 1e-5 * max(1.0, float(d.tail(want).abs().max()))
    # a matrix above the limit: nothing changes
    d2 = dec.PatchDecoder(dict(AE_SHIPPED)).eval()
    assert not d2.enable_dense_head(max_bytes=1 << 20) and getattr(d2, "_dense", None) is None


@pytest.mark.gpu
def test_fused_tail_kernel_matches_torch_tail():
    """csrc/ppp_decode.hip (f32 MFMA over the channels, LDS gathers, crop, float16, scatter) against
    the torch restatement of the same layers on seeded weights, 7^3 patches.  Tolerance: both sides
    round to float16 at the end; before that the two float32 summation orders differ by a few
    ulp, so a value may land on the neighbouring float16: |diff| <= 1 float16 ulp (2^-10 relative)
    + 1e-4 absolute, and at least 99 % of the values identical."""
    torch.manual_seed(11)
    d = dec.PatchDecoder(dict(AE_SHIPPED)).cuda().eval()
    shape = (6, 9, 11)
    V = int(np.prod(shape))
    g = torch.Generator(device="cpu").manual_seed(3)
    codes = torch.randn((401, 176), generator=g).cuda()
    dst = torch.randperm(V, generator=g)[:401].sort().values.cuda()
    want = torch.zeros((343,) + shape, dtype=torch.float16, device="cuda")
    got = torch.zeros_like(want)
    dec.decode_into(d, codes, dst, want, batch_size=128, fused=False)
    dec.decode_into(d, codes, dst, got, batch_size=128, fused=True)
    w, g2 = want.float().reshape(343, -1), got.float().reshape(343, -1)
    assert not g2[:, torch.ones(V, dtype=torch.bool, device="cuda").index_fill(0, dst, False)].any()
    diff = (w - g2).abs()
    assert bool((diff <= w.abs() * 2.0 ** -10 + 1e-4).all()), float(diff.max())
    assert float((diff[:, dst] == 0).float().mean()) > 0.99
    assert float(w[:, dst].abs().max()) > 0        # the decoder produced something
    # float32 destination block as well
    got32 = torch.zeros((343,) + shape, dtype=torch.float32, device="cuda")
    dec.decode_into(d, codes, dst, got32, batch_size=512, fused=True)
    assert torch.equal(got32.to(torch.float16), got)


@pytest.mark.gpu
def test_decode_provider_votes_like_the_decoded_volume_p7():
    """BASELINE config [4] at 7^3: decode -> vote.  (a) the whole volume decoded once into a
    resident float16 block (fused tail kernel), then voted; (b) the prediction decoded TILE BY
    TILE on demand inside the tiled assembly (DecodeProvider: a frame per tile and pass), never
    materialised.  Same instance map, and equal to the CPU oracle voting on (a)'s block."""
    from oracle import ppp_oracle as orc
    from patchperpix_amd import tiling
    from patchperpix_amd.flags import FLYLIGHT
    from patchperpix_amd.vote_instances import vote_instances as vi
    torch.manual_seed(7)
    rng = np.random.default_rng(7)
    d = dec.PatchDecoder(dict(AE_SHIPPED)).cuda().eval()
    shape = (22, 24, 26)
    code = torch.from_numpy(rng.normal(size=(176,) + shape).astype(np.float32)).cuda().half()
    fg = rng.uniform(size=shape) < 0.9
    # random weights: spread and centre the logits so that both classes occur
    probe = dec.decode_volume(d, code, fg, batch_size=2048, out_dtype=torch.float32)
    vals = probe[:, torch.as_tensor(fg, device="cuda")]
    with torch.no_grad():
        scale = 8.0 / float(vals.std())
        d.up_conv[-1][-1].weight.mul_(scale)
        d.up_conv[-1][-1].bias.mul_(scale).sub_(float(vals.median()) * scale)
    del probe, vals
    logits16 = dec.decode_volume(d, code, fg, batch_size=2048, out_dtype=torch.float16)
    fg_t = torch.as_tensor(fg, device="cuda")
    # loadAffinities' expit of the float16 logits: float64 evaluation, narrowed to float32
    pred16 = torch.where(fg_t.expand_as(logits16), torch.sigmoid(logits16.double()).float(),
                         torch.zeros(logits16.shape, dtype=torch.float32, device="cuda"))
    assert (pred16 > 0.5).any() and (pred16 < 0.5).any()
    kw = dict(FLYLIGHT, overlapping_inst=False)
    ps = [7, 7, 7]
    want, _ = vi.to_instance_seg(pred16, fg.copy(), fg.copy(), fg.astype(np.uint8), ps, **dict(kw, _n_slabs=1))
    ref = orc.to_instance_seg(pred16.float().cpu().numpy(), fg, fg.copy(), fg.astype(np.uint8), ps, **kw)
    assert np.array_equal(want, ref["instances"]) and want.any()
    prov = dec.DecodeProvider(d, code, fg, batch_size=2048, expit=True)
    got, _ = tiling.assemble(prov, 0, shape, fg.copy(), fg.copy(), fg.astype(np.uint8), ps,
                             tiling.plan_slabs(shape[0], 2), _yx_tiles=(1, 2), **kw)
    assert np.array_equal(got, want)
    assert prov.voxels_decoded > int(fg.sum())        # halos are decoded again per tile


AE_2D_25 = dict(activation="relu", num_fmaps=[8, 16, 32], downsample_factors=[[2, 2], [2, 2], [2, 2]],
                upsampling="resize_conv", kernel_size=3, num_repetitions=2, padding="same",
                code_fmaps=16, code_units=256, input_shape_squeezed=(25, 25))


def test_decoder_shapes_2d_25():
    """the 2-d Autoencoder (torch_model.py:452-544 is shape-generic: spatial_dims =
    len(input_shape_squeezed)): 256 units -> 16 x 4^2 -> 8^2 -> 16^2 -> 32^2 -> crop 25^2"""
    torch.manual_seed(0)
    d = dec.PatchDecoder(dict(AE_2D_25))
    assert d.nd == 2 and d.code_shape == (-1, 16, 4, 4) and len(d.up) == 3
    out = d(torch.randn(3, 256))
    assert out.shape == (3, 1, 25, 25)
    assert d.fused_tail_params() is None            # (the fused tail kernel is the 7^3 one)
    # the dense-GEMM head does not pay at 8^2 / 16^2 grids: the stages stay convolutions
    assert d.enable_dense_head() is False


@pytest.mark.gpu
def test_decode_2d_25_then_vote():
    """BASELINE config [4], variant (ii) of SURVEY 8(d): the 2-d decoder with 25 x 25 patches on a
    stack of slices, patch shape (1, 25, 25) -- decode_volume against the reference's per-voxel
    loop (decode.py:43-65) on the same decoder, then the vote on the decoded block against the
    oracle.  (Decoder ARITHMETIC unpinned: no funlib, no checkpoint.)"""
    from oracle import ppp_oracle as orc
    from patchperpix_amd.flags import FLYLIGHT
    from patchperpix_amd.vote_instances import vote_instances as vi
    torch.manual_seed(11)
    rng = np.random.default_rng(11)
    d = dec.PatchDecoder(dict(AE_2D_25)).cuda().eval()
    with torch.no_grad():
        d.up_conv[-1][-1].weight.mul_(40.0)
    shape = (2, 40, 44)
    code = rng.normal(size=(256,) + shape).astype(np.float32)
    fg = rng.uniform(size=shape) < 0.85
    logits = dec.decode_volume(d, code, fg, batch_size=256, device="cuda", out_dtype=torch.float32)
    assert tuple(logits.shape) == (625,) + shape
    # literal loop of the reference on a few voxels
    scale = float(logits.abs().max())
    with torch.no_grad():
        for c in np.transpose(np.nonzero(fg))[::97]:
            v = torch.as_tensor(code[(slice(None),) + tuple(c)].reshape(1, 256)).cuda()
            want = d(v).reshape(-1)
            got = logits[(slice(None),) + tuple(int(x) for x in c)]
            # batched vs single-sample convolutions (MIOpen picks its algorithm per shape): 1e-3 of
            # the logits' range
            assert torch.allclose(got, want, rtol=1e-3, atol=1e-3 * scale)
    assert not logits[:, torch.as_tensor(~fg, device="cuda")].any()
    sel = torch.as_tensor(fg, device="cuda").expand_as(logits)
    pred = torch.sigmoid(logits - logits[sel].median()) * torch.as_tensor(fg, device="cuda").float()
    pred16 = pred.to(torch.float16)
    kw = dict(FLYLIGHT, overlapping_inst=False, patch_threshold=0.5)
    ps = [1, 25, 25]
    # 2-d patches are for 2-d data (Z = 1): a stack is refused (pairs across slices are undefined
    # in the reference), the slices are voted one by one
    with pytest.raises(ValueError, match="2-d patches"):
        vi.to_instance_seg(pred16, fg.copy(), fg.copy(), fg.astype(np.uint8), ps, **kw)
    for z in range(shape[0]):
        p16, f = pred16[:, z:z + 1].contiguous(), fg[z:z + 1]
        inst, _ = vi.to_instance_seg(p16, f.copy(), f.copy(), f.astype(np.uint8), ps, **kw)
        ref = orc.to_instance_seg(p16.float().cpu().numpy(), f, f.copy(), f.astype(np.uint8), ps, **kw)
        assert np.array_equal(inst, ref["instances"]) and inst.max() > 0


def _golden_decode_sample(name):
    import json
    import os
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    ae = json.loads(str(f["ae_json"]))
    ae["input_shape_squeezed"] = tuple(ae["input_shape_squeezed"])
    d = dec.PatchDecoder(ae).eval()
    d.load_state_dict({k[2:]: torch.as_tensor(f[k]) for k in f.files if k.startswith("w:")}, strict=True)
    return f, d


DECODE_SAMPLE_CASES = ["ds_p7_numinst3", "ds_p7_fg1", "ds_p5x5_numinst3"]


@pytest.mark.parametrize("name", DECODE_SAMPLE_CASES)
def test_decode_volume_equals_the_references_decode_sample(name):
    """The reference's OWN decode_sample (setup01/decode.py:16-66), run in the development container by
    tests/golden/gen_golden_decode_sample.py with this decoder (weights in the fixture) as
    ``model.decoder``: foreground rule, visiting order, (B, 1, units) batches and the layout of the
    (prod(patchshape), Z, Y, X) block are the reference's.  Same network and same batch size on both
    sides, float32 on the CPU: equal to rounding of the convolutions (tolerance 1e-6 absolute)."""
    f, d = _golden_decode_sample(name)
    fg = dec.foreground_from_numinst(f["numinst"], float(f["fg_thresh"]))
    want = f["output"]
    assert fg.shape == want.shape[1:]
    assert np.array_equal(fg != 0, np.any(want != 0, axis=0))          # decoded exactly on the foreground
    got = dec.decode_volume(d, f["code"], fg, batch_size=int(f["batch"]), device="cpu", fused=False).numpy()
    assert got.shape == want.shape and got.dtype == want.dtype
    assert np.allclose(got, want, rtol=0, atol=1e-6)
    assert not got[:, fg == 0].any()


@pytest.mark.gpu
@pytest.mark.parametrize("name", DECODE_SAMPLE_CASES)
def test_decode_volume_on_the_gpu_equals_the_references_decode_sample(name):
    """The same fixtures through the device path (GEMM / MIOpen head, torch or fused tail): float32
    accumulation in another order, tolerance 1e-4 absolute on logits of magnitude ~1."""
    f, d = _golden_decode_sample(name)
    fg = dec.foreground_from_numinst(f["numinst"], float(f["fg_thresh"]))
    got = dec.decode_volume(d, f["code"], fg, batch_size=int(f["batch"]), device="cuda").cpu().numpy()
    assert np.allclose(got, f["output"], rtol=0, atol=1e-4)
    assert not got[:, fg == 0].any()


@pytest.mark.parametrize("name", ["ae_forward_p7", "ae_forward_p5x5"])
def test_patch_decoder_forward_equals_the_references_autoencoder_forward(name):
    """``Autoencoder.forward`` of the reference (setup01/torch_model.py:523-544, with the real
    ``PatchPerPix.util.crop``, util/train_util.py:55-69) run in the development container on an instance
    carrying THIS decoder's layers (gen_golden_decode_sample.py: its ``__init__`` needs funlib): the
    reshape of the (B, 1, units) code, the order of the stages and the centre crop -- which plane of the
    8 per axis a 7-wide patch drops -- are the reference's.  Same layers, same batch, float32 on the CPU:
    bit-identical."""
    f, d = _golden_decode_sample(name)
    with torch.no_grad():
        got = d(torch.as_tensor(f["code"])).numpy()
    assert got.shape == f["output"].shape
    assert np.array_equal(got, f["output"])
