#!/usr/bin/env python3
"""`label` and `decode` tasks with the reference driver's interface
(reference: experiments/run_ppp.py -- argument names :157-267, task dispatch :1974-2293,
``decode`` :682-746, ``vote_instances`` / ``vote_instances_sample`` :1054-1190).

Only the two tasks on this repository's path are provided; training, prediction and evaluation
stay with the reference.  Usage, as in the reference README:

    python -m patchperpix_amd.run_ppp --setup setup01 --config default.toml --do label \
        --pred-folder <dir with *.zarr|*.hdf|*.npy> --output-folder <dir> [--sample NAME]

Config: the reference's TOML files are read unchanged ([vote_instances], [model], [prediction],
[visualize], [general], [data], [evaluation]).
"""
import argparse
import glob
import logging
import os
import sys
import time

logger = logging.getLogger(__name__)


def load_config(paths):
    try:
        import tomllib as toml_reader   # Python >= 3.11
    except ImportError:
        import tomli as toml_reader
    config = {}
    from .vote_instances.vote_instances import merge_dicts
    for p in paths:
        with open(p, "rb") as f:
            config = merge_dicts(config, toml_reader.load(f))
    return config


def get_list_samples(pred_folder, fmt, only=None):
    names = sorted(os.path.splitext(os.path.basename(p))[0]
                   for p in glob.glob(os.path.join(pred_folder, "*." + fmt)))
    return [n for n in names if only is None or only in n]


def vote_instances_sample(config, pred_folder, output_folder, sample):
    """run_ppp.py:1119-1190."""
    from . import vote_instances as vi
    cfg = config["vote_instances"]
    cfg["result_folder"] = output_folder
    cfg["check_required"] = False
    out_fmt = cfg.get("output_format", "hdf")
    output_fn = os.path.join(output_folder, os.path.basename(sample) + "." + out_fmt)
    if not config.get("general", {}).get("overwrite", False) and os.path.exists(output_fn):
        logger.info("Skipping vote instances for %s. Already exists!", output_fn)
        return
    pred_fmt = config["prediction"]["output_format"]
    pred_file = os.path.join(pred_folder, sample + "." + pred_fmt)
    pred = config["prediction"]
    if cfg.get("blockwise", False):
        vi.stitch_patch_graph.main(
            pred_file, **cfg, **config["model"], **config.get("visualize", {}),
            aff_key=pred.get("aff_key"), numinst_key=pred.get("numinst_key"),
            fg_key=pred.get("fg_key"), fg_folder=pred.get("fg_folder"),
            fg_thresh=pred.get("fg_thresh"))
    else:
        cfg["affinities"] = pred_file
        vi.vote_instances.main(**cfg, **config["model"], numinst_key=pred.get("numinst_key"),
                               aff_key=pred.get("aff_key"), fg_key=pred.get("fg_key"))


def label(args, config):
    samples = get_list_samples(args.pred_folder, config["prediction"]["output_format"],
                               args.sample)
    os.makedirs(args.output_folder, exist_ok=True)
    for idx, sample in enumerate(samples):
        t0 = time.time()
        print("labelling {}/{}: {}".format(idx, len(samples), sample))
        vote_instances_sample(config, args.pred_folder, args.output_folder, sample)
        logger.info("time vote_instances_sample: %.2fs", time.time() - t0)


def decode(args, config):
    from . import decode as dec
    fmt = config["prediction"]["output_format"]
    samples = [os.path.join(args.pred_folder, s + "." + fmt)
               for s in get_list_samples(args.pred_folder, fmt, args.sample)]
    os.makedirs(args.output_folder, exist_ok=True)
    dec.decode(checkpoint_file=args.checkpoint, output_folder=args.output_folder, samples=samples,
               included_ae_config=config.get("autoencoder") or config["model"].get("autoencoder"),
               **config["model"], **config["prediction"], **config.get("data", {}))


def main(argv=None):
    from patchperpix_amd import backend as _backend
    _backend.tune_host_allocator(cli=True)     # (main(argv) IS the program, also behind a console script)
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config", action="append", required=True)
    ap.add_argument("-a", "--app", default="flylight")
    ap.add_argument("-s", "--setup", default="setup01")
    ap.add_argument("-d", "--do", nargs="+", default=["label"], choices=["label", "decode"])
    ap.add_argument("--pred-folder", required=True)
    ap.add_argument("--output-folder", required=True)
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--sample", default=None)
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    config = load_config(args.config)
    for task in args.do:
        {"label": label, "decode": decode}[task](args, config)


if __name__ == "__main__":
    main(sys.argv[1:])
