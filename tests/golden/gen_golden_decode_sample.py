#!/usr/bin/env python3
"""Golden vectors of the reference's ``decode_sample`` (the function
experiments/flylight/setups/setup01/decode.py:16-66).  DEVELOPMENT CONTAINER ONLY: the reference's
module is imported in place from /root/reference and ITS function is called.

What this pins and what it cannot: ``decode_sample`` is the host side of the decode step -- which
voxels are foreground (numinst P(0) < 0.1 with several channels, ``>= fg_thresh`` with one), the order
they are visited in, how a voxel's code reaches the decoder ((B, 1, code_units) batches) and where the
decoded patch lands in the (prod(patchshape), Z, Y, X) float32 block.  The decoder NETWORK the
reference hands to it (``UnetModelWrapper.decoder``, an ``Autoencoder`` of funlib.learn.torch's
ConvPass / Upsample) cannot be built here -- that package is a git dependency the image lacks --
so ``model.decoder`` is this repository's PatchDecoder with seeded weights, which are stored in the
fixture: both sides of the test evaluate the SAME network, the reference's loop decides everything
else.  The decoder's arithmetic stays unpinned.

Import-time stubs only (none of them is called): ``toml``, ``h5py``, the sibling ``torch_model``
module and ``PatchPerPix.visualize``; ``zarr.open`` is replaced by a reader of in-memory arrays, the
one container access ``decode_sample`` makes.

A second kind of fixture (``ae_forward_*``) runs the reference's ``Autoencoder.forward``
(setup01/torch_model.py:523-544) with the real ``PatchPerPix.util.crop`` on an instance carrying this
repository's layers: see ``autoencoder_forward``.

  python tests/golden/gen_golden_decode_sample.py [case ...]
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_DECODE = "/root/reference/experiments/flylight/setups/setup01/decode.py"
REF_MODEL = "/root/reference/experiments/flylight/setups/setup01/torch_model.py"
REF_TRAIN_UTIL = "/root/reference/PatchPerPix/util/train_util.py"
sys.path.insert(0, REPO)

from patchperpix_amd import decode as dec  # noqa: E402

AE3 = dict(activation="relu", num_fmaps=[8, 16], downsample_factors=[[2, 2, 2], [2, 2, 2]],
           upsampling="resize_conv", kernel_size=3, num_repetitions=2, padding="same",
           code_fmaps=22, code_units=176, input_shape_squeezed=(7, 7, 7))
AE2 = dict(activation="relu", num_fmaps=[6, 12], downsample_factors=[[2, 2], [2, 2]],
           upsampling="resize_conv", kernel_size=3, num_repetitions=2, padding="same",
           code_fmaps=9, code_units=36, input_shape_squeezed=(5, 5))

CASES = {
    # name: (decoder config, patchshape, volume shape, numinst channels, fg_thresh, batch size, seed)
    # three numinst channels: foreground where P(0 instances) < 0.1 (decode.py:33-34)
    "ds_p7_numinst3": (AE3, (7, 7, 7), (4, 5, 6), 3, 0.5, 7, 11),
    # one foreground channel: >= fg_thresh, squeezed (decode.py:35-37); a batch larger than the foreground
    "ds_p7_fg1": (AE3, (7, 7, 7), (3, 4, 5), 1, 0.6, 1024, 12),
    # 2-d patches on a one-slice volume (patchshape[patchshape > 1], decode.py:20-22)
    "ds_p5x5_numinst3": (AE2, (1, 5, 5), (1, 9, 11), 3, 0.5, 16, 13),
}


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference_decode(store):
    """the reference's decode.py as module `ppp_ref_setup01.decode`; zarr.open(sample) -> store[sample]"""
    def unused(*a, **k):
        raise RuntimeError("stubbed function called")
    _stub("toml", load=unused)
    _stub("h5py", File=unused)
    _stub("zarr", open=lambda path, mode="r": store[path])
    pkg = _stub("ppp_ref_setup01")
    pkg.__path__ = []
    pkg.torch_model = _stub("ppp_ref_setup01.torch_model", UnetModelWrapper=unused)
    _stub("PatchPerPix")
    _stub("PatchPerPix.visualize", visualize_patches=unused)
    spec = importlib.util.spec_from_file_location("ppp_ref_setup01.decode", REF_DECODE)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


class Model:
    def __init__(self, decoder):
        self.decoder = decoder


def load_reference_autoencoder():
    """The reference's ``Autoencoder`` class (setup01/torch_model.py:452-544) with the REAL
    ``PatchPerPix.util.crop`` (util/train_util.py:55-69).  Import-time stubs, never called: monai,
    torchinfo, gunpowder and the NAMES torch_model imports from funlib.learn.torch.models."""
    def unused(*a, **k):
        raise RuntimeError("stubbed name used")
    _stub("gunpowder")
    spec = importlib.util.spec_from_file_location("ppp_ref_train_util", REF_TRAIN_UTIL)
    tu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tu)
    names = ("crop crop_to_factor gather_nd_torch gather_nd_torch_no_batch "
             "seg_to_affgraph_3d_multi_torch_code seg_to_affgraph_2d_multi_torch_code "
             "seg_to_affgraph_3d_multi_torch seg_to_affgraph_2d_multi_torch seg_to_affgraph_3d_torch "
             "seg_to_affgraph_2d_torch seg_to_affgraph_3d_torch_code seg_to_affgraph_2d_torch_code").split()
    _stub("PatchPerPix")
    _stub("PatchPerPix.util", **{n: getattr(tu, n) for n in names})
    _stub("monai")
    _stub("torchinfo")
    _stub("funlib")
    _stub("funlib.learn")
    _stub("funlib.learn.torch")
    _stub("funlib.learn.torch.models", UNet=unused, ConvPass=unused, Downsample=unused, Upsample=unused)
    spec = importlib.util.spec_from_file_location("ppp_ref_torch_model", REF_MODEL)
    tm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tm)
    return tm.Autoencoder


AE_FORWARD_CASES = {
    # name: (decoder config, batch, seed) -- 2^3 -> 4^3 -> 8^3 -> crop 7^3;  2^2 -> 4^2 -> 8^2 -> crop 5^2
    "ae_forward_p7": (AE3, 5, 21),
    "ae_forward_p5x5": (AE2, 6, 22),
}


def autoencoder_forward(names):
    """``Autoencoder.forward`` of the reference on an instance whose LAYERS are this repository's
    (``__init__`` builds them from funlib's ConvPass / Upsample, which the image lacks, so it is not
    run): the reshape to ``code_shape``, the order from_code -> [up_i -> up_conv_i] and the centre
    crop (which of the 8 planes per axis is dropped for a 7-wide patch) are the reference's."""
    Autoencoder = load_reference_autoencoder()
    for name in names:
        ae, batch, seed = AE_FORWARD_CASES[name]
        torch.manual_seed(seed)
        mine = dec.PatchDecoder(dict(ae)).eval()
        ref = Autoencoder.__new__(Autoencoder)
        torch.nn.Module.__init__(ref)
        ref.config = dict(ae)
        ref.code_shape = mine.code_shape
        ref.from_code, ref.up, ref.up_conv = mine.from_code, mine.up, mine.up_conv
        code = torch.randn(batch, 1, ae["code_units"])       # (B, 1, units): what decode_sample hands over
        with torch.no_grad():
            out = Autoencoder.forward(ref, code)
        weights = {"w:" + k: v.numpy() for k, v in mine.state_dict().items()}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), code=code.numpy(), output=out.numpy(),
                            ae_json=np.array(json.dumps(ae)), **weights)
        print("%-20s code %s -> %s" % (name, tuple(code.shape), tuple(out.shape)))


def main(argv):
    fwd = [n for n in (argv or list(AE_FORWARD_CASES)) if n in AE_FORWARD_CASES]
    if fwd:
        autoencoder_forward(fwd)
    store = {}
    ref = load_reference_decode(store)
    for name in [n for n in (argv or list(CASES)) if n in CASES]:
        ae, ps, shape, nch, fg_thresh, batch, seed = CASES[name]
        torch.manual_seed(seed)
        rng = np.random.default_rng(seed)
        decoder = dec.PatchDecoder(dict(ae)).eval()
        code = rng.normal(size=(ae["code_units"],) + shape).astype(np.float32)
        numinst = rng.uniform(size=(nch,) + shape).astype(np.float32)
        if nch > 1:
            numinst[0] *= 0.2          # P(0) < 0.1 on about half the voxels
        store["sample"] = {"code": code, "numinst": numinst}
        cfg = dict(decode_batch_size=batch, code_units=ae["code_units"], patchshape=list(ps),
                   output_format="zarr", code_key="code", numinst_key="numinst", fg_thresh=fg_thresh)
        with torch.no_grad():
            out = ref.decode_sample(cfg, Model(decoder), "sample", torch.device("cpu"))
        assert out.dtype == np.float32 and out.shape[0] == int(np.prod(ps))
        weights = {"w:" + k: v.numpy() for k, v in decoder.state_dict().items()}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), code=code, numinst=numinst,
                            patchshape=np.array(ps), fg_thresh=np.array(fg_thresh), batch=np.array(batch),
                            ae_json=np.array(json.dumps(ae)), output=out, **weights)
        nfg = int(np.count_nonzero(np.any(out != 0, axis=0)))
        print("%-20s out %s  decoded voxels %d of %d" % (name, out.shape, nfg, int(np.prod(shape))))


if __name__ == "__main__":
    main(sys.argv[1:])
