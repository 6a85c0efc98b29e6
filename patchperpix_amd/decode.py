"""ppp+dec: decode a per-voxel code into the patch prediction before voting
(reference: experiments/flylight/setups/setup01/decode.py:16-130 and the decoder half of
``Autoencoder``, setup01/torch_model.py:452-544).

The decoder is the only dense-contraction piece next to the hot path.  Its head -- the 1x1
``from_code`` and the first upsampling stage, 28 of the 30 M multiply-adds per voxel: dense 64 /
128-channel convolutions at 4^3 -- runs as torch-ROCm convolutions (MIOpen / rocBLAS); its tail --
the last stage, the crop, the conversion to float16 and the scatter into the (C, Z, Y, X)
prediction block -- is one hand-written HIP kernel (csrc/ppp_decode.hip: f32 MFMA + LDS), so a
batch goes from codes to float16 patch values in the block S1 reads with no float32 patch
tensors and no per-voxel Python loop (decode.py:43-65).  ``decode_volume`` fills a resident float16
block batch by batch; ``DecodeProvider`` decodes tile by tile on demand inside the tiled assembly
(``patchperpix_amd.tiling``), for volumes whose decoded prediction would not fit.

PARITY UNPINNED for the decoder arithmetic: the reference builds it from
``funlib.learn.torch`` (``ConvPass``, ``Upsample``; git dependency, branch ``ppp``, not vendored
and not importable here) and ships no checkpoint.  ``PatchDecoder`` restates the published
structure of those blocks (conv stack with "same" padding and an activation after every conv;
``resize_conv`` = nearest-neighbour upsampling followed by one conv pass).  The gather / scatter
semantics ARE pinned: tests/golden/ds_*.npz hold outputs of the reference's own ``decode_sample``
run with a PatchDecoder as ``model.decoder`` (tests/golden/gen_golden_decode_sample.py).
"""
import logging
import os

import numpy as np

logger = logging.getLogger(__name__)


def _torch():
    import torch
    return torch


def _conv_pass(nd, c_in, c_out, kernel_sizes, activation):
    torch = _torch()
    conv = {2: torch.nn.Conv2d, 3: torch.nn.Conv3d}[nd]
    layers = []
    for ks in kernel_sizes:
        layers.append(conv(c_in, c_out, ks, padding=tuple(k // 2 for k in ks)))
        if activation is not None:
            layers.append(getattr(torch.nn, activation)())
        c_in = c_out
    return torch.nn.Sequential(*layers)


class PatchDecoder(_torch().nn.Module):
    """Decoder half of the reference's Autoencoder (torch_model.py:497-544): reshape the code to
    (code_fmaps, s, .., s), 1x1 conv ``from_code``, then per stage [upsample -> conv pass]; the
    last conv pass has no activation (logits); centre crop to the patch shape."""

    def __init__(self, config):
        super().__init__()
        torch = _torch()
        act = {"relu": "ReLU", "sigmoid": "Sigmoid"}.get(config["activation"], config["activation"])
        self.patchshape = tuple(int(p) for p in config["input_shape_squeezed"])
        nd = self.nd = len(self.patchshape)
        ks = [[int(config["kernel_size"])] * nd] * int(config["num_repetitions"])
        self.code_fmaps = int(config["code_fmaps"])
        self.code_units = int(config["code_units"])
        s = round((self.code_units / self.code_fmaps) ** (1.0 / nd))
        assert s ** nd * self.code_fmaps == self.code_units, \
            "size of spatially reshaped code has to add up to code units"
        self.code_shape = (-1, self.code_fmaps) + (s,) * nd
        nf_prev = int(config["num_fmaps"][-1])
        self.from_code = _conv_pass(nd, self.code_fmaps, nf_prev, [[1] * nd], act)
        ups, convs = [], []
        stages = list(reversed(config["num_fmaps"]))[1:] + [1]
        for idx, nf in enumerate(stages):
            factor = tuple(int(f) for f in config["downsample_factors"][-idx])
            if config.get("upsampling", "resize_conv") != "resize_conv":
                raise NotImplementedError("only resize_conv upsampling is restated")
            ups.append(torch.nn.Sequential(
                torch.nn.Upsample(scale_factor=factor, mode="nearest"),
                _conv_pass(nd, nf_prev, nf, [[int(config["kernel_size"])] * nd], act)))
            convs.append(_conv_pass(nd, nf, nf, ks, None if nf == 1 else act))
            nf_prev = nf
        self.up = torch.nn.ModuleList(ups)
        self.up_conv = torch.nn.ModuleList(convs)

    def head(self, code):
        """Everything but the last stage: (B, code_units) -> the features the last stage upsamples
        (B, num_fmaps[0], s', ..) -- the dense 64 / 128-channel convolutions.  As convolutions
        (MIOpen), or -- after ``enable_dense_head`` -- as one library GEMM per convolution."""
        torch = _torch()
        out = self.from_code(torch.reshape(code, self.code_shape))
        dense = getattr(self, "_dense", None)
        first = 0
        if dense is not None:
            shape = dense["out_shape"]
            out = out.reshape(out.shape[0], -1)
            for W, b, act in dense["stages"]:
                out = torch.addmm(b, out, W)
                if act is not None:
                    out = act(out)
            out = out.reshape((-1,) + shape)
            first = dense["n_stages"]
        for up, conv in zip(self.up[first:-1], self.up_conv[first:-1]):
            out = conv(up(out))
        return out

    def enable_dense_head(self, max_bytes=1 << 30, max_flop_ratio=3.0):
        """Turn the head's convolutions into plain GEMMs.  At the 2^3 -> 4^3 grids of the head a
        "same"-padded 3^3 convolution is a DENSE linear map between (channels x positions)
        vectors -- every output position sees most input positions -- so each convolution
        (the first one together with the nearest-neighbour upsampling in front of it) becomes
        ``relu(x @ W + b)`` with W of shape (C_in * positions_in, C_out * positions_out): 1024 x
        4096 and twice 4096 x 4096 for the shipped decoder.  1.33x the multiply-adds of the direct
        form (the upsampled convolution gets 3.4x cheaper, the others 2.4x dearer), but as square
        float32 library GEMMs instead of 4^3-sized convolutions.  W is made by pushing the
        identity through the layer itself, so it holds exactly the layer's weights; the
        summation order differs from the convolution's (tests/test_decode.py states the
        tolerance).  Stages are converted from the code outwards while that pays: a stage whose
        matrices would exceed max_bytes, or whose dense form costs more than `max_flop_ratio` times
        the multiply-adds of its convolutions (grids beyond ~4^3 / 8^2: the 2-d 25 x 25 decoder's
        16^2 stage would cost 28x), stays a convolution, and so do the stages after it.  Returns
        False (and changes nothing) when no stage qualifies."""
        torch = _torch()
        if len(self.up) < 2:
            return False
        p0 = next(self.parameters())
        dev, s0 = p0.device, self.code_shape[2:]
        in_shape = (int(self.from_code[0].out_channels),) + tuple(int(v) for v in s0)
        stages = []
        done_stages, done_units, done_shape = 0, 0, in_shape
        with torch.no_grad():
            for up, conv in zip(self.up[:-1], self.up_conv[:-1]):
                # multiply-adds of the stage as convolutions / as dense maps
                direct = dense_cost = 0
                shp = in_shape
                z = torch.zeros((1,) + shp, device=dev, dtype=p0.dtype)
                for m in [up[0]] + list(up[1]) + list(conv):
                    if isinstance(m, (torch.nn.Conv2d, torch.nn.Conv3d)):
                        direct += int(np.prod(z.shape[2:])) * m.in_channels * m.out_channels * int(np.prod(m.kernel_size))
                    z = m(z)
                    if isinstance(m, (torch.nn.Conv2d, torch.nn.Conv3d)):
                        dense_cost += int(np.prod(shp)) * int(np.prod(z.shape[1:]))
                        shp = tuple(int(v) for v in z.shape[1:])
                if dense_cost > max_flop_ratio * direct:
                    break
                units = []              # (linear part, activation or None)
                mods = [up[0]] + list(up[1]) + list(conv)
                lin = []
                for m in mods:
                    if isinstance(m, (torch.nn.Upsample, torch.nn.Conv2d, torch.nn.Conv3d)):
                        if lin and isinstance(lin[-1], (torch.nn.Conv2d, torch.nn.Conv3d)):
                            units.append((lin, None))
                            lin = []
                        lin.append(m)
                    else:
                        units.append((lin, m))
                        lin = []
                if lin:
                    units.append((lin, None))
                for lin, act in units:
                    k = int(np.prod(in_shape))
                    zero = torch.zeros((1,) + in_shape, device=dev, dtype=p0.dtype)
                    y0 = zero
                    for m in lin:
                        y0 = m(y0)
                    out_shape = tuple(int(v) for v in y0.shape[1:])
                    n = int(np.prod(out_shape))
                    if k * n * 4 > max_bytes:
                        stages = stages[:done_units]
                        in_shape = done_shape
                        break
                    W = torch.empty((k, n), device=dev, dtype=p0.dtype)
                    step = max(1, (1 << 26) // max(n, k))
                    for a in range(0, k, step):
                        b = min(k, a + step)
                        eye = torch.zeros((b - a, k), device=dev, dtype=p0.dtype)
                        eye[torch.arange(b - a, device=dev), torch.arange(a, b, device=dev)] = 1
                        y = eye.reshape((b - a,) + in_shape)
                        for m in lin:
                            y = m(y)
                        W[a:b] = (y - y0).reshape(b - a, n)
                    stages.append((W, y0.reshape(1, n).clone(), act))
                    in_shape = out_shape
                else:
                    done_stages, done_units, done_shape = done_stages + 1, len(stages), in_shape
                    continue
                break
        if done_stages == 0:
            return False
        self._dense = {"stages": stages[:done_units], "out_shape": done_shape, "n_stages": done_stages}
        return True

    def tail(self, feats):
        """The last stage + centre crop (PatchPerPix/util: crop) as torch ops (the restatement the
        fused kernel ppp_decode_tail is checked against)."""
        out = self.up_conv[-1](self.up[-1](feats))
        sl = [slice(None), slice(None)]
        for have, want in zip(out.shape[2:], self.patchshape):
            o = (have - want) // 2
            sl.append(slice(o, o + want))
        return out[tuple(sl)]

    def forward(self, code):
        return self.tail(self.head(code))

    def fused_tail_params(self):
        """(w1, b1, w2, b2, w3, b3) when the last stage is the shipped one -- 3-d, nearest x2, one
        3^3 convolution to a single map + ReLU, two 3^3 single-map convolutions without
        activation, 8^3 -> 7^3 crop -- which csrc/ppp_decode.hip runs fused with the scatter into
        the prediction block; None otherwise (torch ops do the tail then)."""
        torch = _torch()
        if self.nd != 3 or self.patchshape != (7, 7, 7) or len(self.up) < 1:
            return None
        up, conv = self.up[-1], self.up_conv[-1]
        ups, first = up[0], up[1]
        convs = [m for m in first if isinstance(m, torch.nn.Conv3d)]
        acts = [m for m in first if not isinstance(m, torch.nn.Conv3d)]
        tail = [m for m in conv]
        sf = ups.scale_factor if isinstance(ups.scale_factor, (tuple, list)) else (ups.scale_factor,) * 3
        if tuple(float(f) for f in sf) != (2.0, 2.0, 2.0) or len(convs) != 1 or \
                not (len(acts) == 1 and isinstance(acts[0], torch.nn.ReLU)) or \
                len(tail) != 2 or not all(isinstance(m, torch.nn.Conv3d) for m in tail):
            return None
        c1 = convs[0]
        if c1.in_channels != 64 or c1.out_channels != 1 or tuple(c1.kernel_size) != (3, 3, 3) or \
                any(tuple(m.kernel_size) != (3, 3, 3) or m.in_channels != 1 or m.out_channels != 1 for m in tail):
            return None
        return (c1.weight, float(c1.bias.item()), tail[0].weight, float(tail[0].bias.item()),
                tail[1].weight, float(tail[1].bias.item()))


def foreground_from_numinst(pred_numinstfg, fg_thresh):
    """decode.py:33-37: numinst probabilities [0, 1, 2] -> fg where P(0) < 0.1, else threshold."""
    if pred_numinstfg.shape[0] > 1:
        return np.array(pred_numinstfg[0] < 0.1).astype(np.uint8)
    return np.squeeze((pred_numinstfg >= fg_thresh).astype(np.uint8))


def decode_into(decoder, codes, dst, pred, batch_size=1024, fused=None):
    """Decode `codes` (B, code_units; device) and write patch k into pred[:, dst[k]] -- pred is the
    (C, ...) prediction block (float16 or float32, device), dst int64 linear voxel indices.
    The decoder's head runs as torch convolutions; the tail (last stage, crop, conversion to
    pred's dtype, scatter) in ONE fused HIP kernel when the decoder has the shipped shape
    (ppp_decode_tail, csrc/ppp_decode.hip), else as torch ops + an index assignment.
    fused: None = the kernel when possible; False = torch ops; True = the kernel or an error."""
    torch = _torch()
    from . import backend
    C = int(np.prod(decoder.patchshape))
    tp = decoder.fused_tail_params() if fused is not False else None
    if fused and tp is None:
        raise RuntimeError("this decoder's tail has no fused kernel")
    flat = pred.reshape(C, -1)
    if getattr(decoder, "_dense", None) is None and pred.is_cuda and \
            os.environ.get("PPP_DECODE_HEAD", "dense") != "conv" and not getattr(decoder, "_dense_tried", False):
        decoder._dense_tried = True
        decoder.enable_dense_head()
    # Convolutions go through MIOpen, which searches for an algorithm the first time it sees a
    # problem SHAPE -- seconds per new batch size.  A ragged last batch (a different size for every
    # call: every slice, every tile) is therefore padded to the full batch whenever a convolution
    # will see it; the padded rows are decoded and dropped.  (GEMM head + fused tail: no MIOpen.)
    convs_run = tp is None or getattr(decoder, "_dense", None) is None or \
        decoder._dense["n_stages"] < len(decoder.up) - 1
    with torch.no_grad():
        for s in range(0, int(dst.numel()), int(batch_size)):
            sel = dst[s:s + batch_size]
            cb = codes[s:s + batch_size]
            n = int(sel.numel())
            if convs_run and n < int(batch_size) and pred.is_cuda:
                cb = torch.cat([cb, cb.new_zeros((int(batch_size) - n,) + tuple(cb.shape[1:]))], 0)
            feats = decoder.head(cb)
            if tp is not None:
                backend.decode_tail(feats[:n].contiguous() if feats.shape[0] != n else feats,
                                    tp[0], tp[1], tp[2], tp[3], tp[4], tp[5], sel, pred,
                                    decoder.patchshape)
            else:
                flat[:, sel] = decoder.tail(feats)[:n].reshape(n, C).t().to(pred.dtype)
    return pred


def decode_volume(decoder, pred_code, pred_fg, batch_size=1024, device="cuda", out_dtype=None,
                  fused=None):
    """decode_sample (decode.py:16-66) on the device: returns the (C, *spatial) prediction
    tensor (float32 unless out_dtype; float16 = the dtype it is written in, decode.py:104-109),
    zero outside the foreground.  Batches of `batch_size` voxels go through the decoder and
    straight into the block in its final dtype: no float32 (C, Z, Y, X) intermediate when
    out_dtype is float16."""
    torch = _torch()
    out_dtype = out_dtype or torch.float32
    code = torch.as_tensor(np.asarray(pred_code) if not torch.is_tensor(pred_code) else pred_code, device=device)
    fg = torch.as_tensor(np.asarray(pred_fg) != 0 if not torch.is_tensor(pred_fg) else pred_fg != 0, device=device)
    units = code.shape[0]
    C = int(np.prod(decoder.patchshape))
    flat_code = code.reshape(units, -1)
    idx = torch.nonzero(fg.reshape(-1)).reshape(-1)
    out = torch.zeros((C,) + tuple(fg.shape), dtype=out_dtype, device=device)
    decoder = decoder.to(device).eval()
    # codes are gathered slab by slab (a transposed float32 copy of ALL codes would be as large
    # as a third of the prediction)
    chunk = max(int(batch_size), 1 << 18)
    for s in range(0, int(idx.numel()), chunk):
        sel = idx[s:s + chunk]
        decode_into(decoder, flat_code[:, sel].t().float().contiguous(), sel, out, batch_size, fused)
    return out


class DecodeProvider:
    """Prediction provider for patchperpix_amd.tiling.assemble in ppp+dec mode: pred_box() DECODES
    the float16 prediction of a box on demand from the per-voxel code (code_units, Z, Y, X) -- a
    rank never holds the decoded (C, Z, Y, X) volume (343 x 2 bytes per voxel at 7^3, 31 KB at
    25^3), only the code (2 x code_units bytes per voxel) and one tile.  The price is decoding the
    halo of every tile again (x2-x3 at 128^3 tiles), which is why volumes whose float16 prediction
    fits HBM are decoded once (decode_volume) instead."""

    def __init__(self, decoder, code, fg, batch_size=4096, device="cuda", fused=None, expit=False):
        """expit: pass the decoded logits through the logistic function the way loadAffinities
        does with the float16 array it reads (utilVoteInstances.py:249-250: scipy.special.expit;
        with the scipy of this image -- loops d, f, g -- a float16 array is evaluated in float64,
        the result narrowed to float32 later): the tile is then float32, like ZarrProvider's.
        PARITY UNPINNED in the last bit against another scipy / libm."""
        torch = _torch()
        self.expit = bool(expit)
        self.decoder = decoder.to(device).eval()
        self.code = torch.as_tensor(code, device=device)
        self.fg = torch.as_tensor(np.asarray(fg) != 0 if not torch.is_tensor(fg) else fg != 0, device=device)
        self.batch_size, self.device, self.fused = int(batch_size), device, fused
        self.voxels_decoded = 0

    def pred_box(self, box):
        torch = _torch()
        z0, z1, y0, y1, x0, x1 = [int(v) for v in box]
        C = int(np.prod(self.decoder.patchshape))
        pred = torch.zeros((C, z1 - z0, y1 - y0, x1 - x0), dtype=torch.float16, device=self.device)
        fg = self.fg[z0:z1, y0:y1, x0:x1]
        dst = torch.nonzero(fg.reshape(-1)).reshape(-1)
        if dst.numel():
            codes = self.code[:, z0:z1, y0:y1, x0:x1].reshape(self.code.shape[0], -1)[:, dst].t().float().contiguous()
            decode_into(self.decoder, codes, dst, pred, self.batch_size, self.fused)
            self.voxels_decoded += int(dst.numel())
        if self.expit:
            # float16 logits -> float64 logistic -> float32 (background voxels stay 0, as decode
            # leaves them: the reference applies expit to the whole array, where logit 0 -> 0.5;
            # those voxels are outside the foreground and never read by the vote)
            out = torch.zeros(pred.shape, dtype=torch.float32, device=self.device)
            if dst.numel():
                flat, oflat = pred.reshape(C, -1), out.reshape(C, -1)
                step = max(1, (1 << 24) // max(1, int(dst.numel())))
                for c0 in range(0, C, step):
                    oflat[c0:c0 + step, dst] = torch.sigmoid(flat[c0:c0 + step, dst].double()).float()
            return out
        return pred


def map_decoder_state(state, decoder):
    """Reference checkpoint -> PatchDecoder state dict, explicitly and completely.

    The reference keeps the decoder as ``UnetModelWrapper.decoder = Autoencoder(...)``
    (torch_model.py:145): its parameters are the checkpoint keys that contain ``decoder.``.  Blocks
    are funlib.learn.torch ``ConvPass`` modules (parameters ``<block>.conv_pass.<n>.{weight,bias}``,
    n counting convolutions AND activations, exactly like the ``Sequential`` built here) and
    ``Upsample`` modules in ``resize_conv`` mode (one convolution each; its sub-module names on the
    ``ppp`` branch are not known here, so the single weight / bias pair under ``up.<i>.`` is taken
    whatever it is called).  The encoder half (``down_conv``, ``down``, ``to_code``) is not used by
    decode and is dropped by name.  Anything else that is left over, missing, or of the wrong
    shape raises: a decoder must never run on partly random weights."""
    dec = {k.split("decoder.", 1)[1]: v for k, v in state.items() if "decoder." in k}
    if not dec:
        raise KeyError("checkpoint holds no `decoder.` parameters")
    dec = {k: v for k, v in dec.items() if not k.startswith(("down_conv.", "down.", "to_code."))}
    own = decoder.state_dict()
    out, used = {}, set()

    def take(src, dst):
        if src not in dec:
            raise KeyError("decoder parameter %s (for %s) is not in the checkpoint" % (src, dst))
        if tuple(dec[src].shape) != tuple(own[dst].shape):
            raise ValueError("decoder parameter %s has shape %s, expected %s" %
                             (src, tuple(dec[src].shape), tuple(own[dst].shape)))
        out[dst] = dec[src]
        used.add(src)

    for dst in own:
        parts = dst.split(".")
        if parts[0] == "from_code":                       # from_code.<n>.weight
            take("from_code.conv_pass.%s.%s" % (parts[1], parts[2]), dst)
        elif parts[0] == "up_conv":                       # up_conv.<i>.<n>.weight
            take("up_conv.%s.conv_pass.%s.%s" % (parts[1], parts[2], parts[3]), dst)
        elif parts[0] == "up":                            # up.<i>.1.<n>.weight: the stage's one conv
            cands = [k for k in dec if k.startswith("up.%s." % parts[1]) and k.endswith("." + parts[-1])]
            if len(cands) != 1:
                raise KeyError("expected one %s under decoder.up.%s, found %s" % (parts[-1], parts[1], cands))
            take(cands[0], dst)
        else:
            raise KeyError("unexpected parameter %s in PatchDecoder" % dst)
    left = sorted(set(dec) - used)
    if left:
        raise KeyError("checkpoint decoder parameters without a counterpart: %s" % left)
    return out


def decode(**config):
    """decode(**cfg) of the reference (decode.py:69-130): load the decoder weights, decode every
    sample, write ``aff_key`` as float16."""
    torch = _torch()
    from .vote_instances import io_hdflike
    if not torch.cuda.is_available():
        raise RuntimeError("patchperpix_amd.decode needs a GPU (there is no CPU fallback)")
    device = torch.device("cuda")
    ae = dict(config.get("autoencoder") or config.get("included_ae_config") or {})
    ae.setdefault("code_units", config["code_units"])
    ae.setdefault("input_shape_squeezed",
                  tuple(int(p) for p in config["patchshape"] if int(p) > 1))
    if ae.get("padding", "same") != "same":
        raise NotImplementedError("decoder padding %r: only 'same' is restated" % ae.get("padding"))
    decoder = PatchDecoder(ae)
    ckpt = torch.load(config["checkpoint_file"], map_location=device)
    state = ckpt["swa_model_state_dict" if config.get("use_swa") else "model_state_dict"]
    if config.get("use_swa"):    # AveragedModel prefixes every key with `module.` and adds n_averaged
        state = {k.split("module.", 1)[1]: v for k, v in state.items() if k.startswith("module.")}
    decoder.load_state_dict(map_decoder_state(state, decoder), strict=True)
    for sample in config["samples"]:
        with io_hdflike.open_container(sample, "r") as f:
            code = np.array(f[config["code_key"]])
            numinstfg = np.array(f[config.get("numinst_key", config.get("fg_key"))])
        fg = foreground_from_numinst(numinstfg, config.get("fg_thresh", 0.5))
        # decoded batch by batch straight into the float16 block that is written (decode.py:104-109)
        pred = decode_volume(decoder, code, fg, config.get("decode_batch_size", 1024), device,
                             out_dtype=torch.float16)
        name = os.path.basename(sample).split(".")[0]
        outfn = os.path.join(config["output_folder"], name + "." + config["output_format"])
        data = pred.cpu().numpy()
        if config["output_format"] == "zarr":
            with io_hdflike.open_container(outfn, "a" if os.path.exists(outfn) else "w") as f:
                # decode.py:104-109: no chunks / compressor given -> the container's defaults
                ds = f.create(config["aff_key"], shape=data.shape, dtype=np.float16,
                              chunks=(data.shape[0],) + tuple(min(int(s), 64) for s in data.shape[1:]))
                ds.attrs["offset"] = [0] * len(config.get("voxel_size", [1] * (data.ndim - 1)))
                ds.attrs["resolution"] = list(config.get("voxel_size", [1] * (data.ndim - 1)))
                ds[:] = data
        elif config["output_format"] == "hdf":
            with io_hdflike.open_container(outfn, "a" if os.path.exists(outfn) else "w") as f:
                f.create_dataset(config["aff_key"], data=data, compression="gzip")
        else:
            raise NotImplementedError(config["output_format"])
