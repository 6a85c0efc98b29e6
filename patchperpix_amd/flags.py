"""Flag sets for ``to_instance_seg`` taken from the reference's shipped flylight config
(experiments/flylight/setups/setup01/default.toml:114-169, plus ``overlapping_inst`` from
[model]).  Three named sets, so that every caller says which one it runs:

``FLYLIGHT``         the [vote_instances] section as shipped: ``mws = true``,
                     ``skipThinCover = false`` (default.toml:134,141).  Deviations, each forced
                     by this package's scope and stated wherever the set is reported:
                     ``blockwise = false`` (whole-volume assembly; the tiled path reproduces the
                     whole-volume result instead of the per-block cover, DESIGN.md section 6;
                     the per-block scheme is ``blockwise_semantics="reference"`` of the
                     blockwise driver), ``skeletonize_foreground`` not set in the benchmark sets
                     (the synthetic volumes are dense cells, not tubes; the option itself is
                     served -- scikit-image, else the library's own 3-d thinning),
                     ``removeIntersection`` kept but without effect (the reference reads it in
                     its NumPy branch only, aff_patch_graph.py:244).
``FLYLIGHT_CC``      the same with ``mws = false`` -- the value the config's own validation sweep
                     uses (default.toml:99): connected components instead of the mutex watershed.
``FLYLIGHT_NOTHIN_CC`` additionally ``skipThinCover = true``: the kernels-only pipeline
                     (S1, S2, greedy cover, pairs, S5, union-find); what round 1 benchmarked.
"""
_COMMON = dict(
    patch_threshold=0.5, fc_threshold=0.5, cuda=True, blockwise=False,
    select_patches_for_sparse_data=True, includeSinglePatchCCS=True,
    removeIntersection=True,
    consensus_interleaved_cnt=False, consensus_norm_prob_product=True,
    consensus_prob_product=True, consensus_norm_aff=True,
    vi_bg_use_inv_th=False, vi_bg_use_half_th=False, vi_bg_use_less_than_th=True,
    rank_norm_patch_score=True, rank_int_counter=False, patch_graph_norm_aff=True,
    flip_cons_arr_axes=False, pad_with_ps=False, overlapping_inst=True,
    max_total_patch_distance_in_ps_multiples=2,
    # bookkeeping keys the reference's stage functions expect in kwargs
    debug=False, isbiHack=False, save_no_intermediates=True, sample=1.0,
    result_folder="/tmp", affinities="synthetic.zarr",
)
FLYLIGHT = dict(_COMMON, mws=True, skipThinCover=False)
FLYLIGHT_CC = dict(_COMMON, mws=False, skipThinCover=False)
FLYLIGHT_NOTHIN_CC = dict(_COMMON, mws=False, skipThinCover=True)
FLAG_SETS = {"shipped": FLYLIGHT, "cc": FLYLIGHT_CC, "nothin_cc": FLYLIGHT_NOTHIN_CC}
# the entries in which the sets differ from each other / from default.toml, for reports
REPORTED = ("mws", "skipThinCover", "blockwise", "removeIntersection", "patch_threshold",
            "fc_threshold", "select_patches_for_sparse_data", "includeSinglePatchCCS",
            "overlapping_inst", "consensus_norm_prob_product", "consensus_norm_aff",
            "vi_bg_use_less_than_th", "rank_norm_patch_score", "patch_graph_norm_aff")


def describe(kw):
    """The flags worth printing next to a measurement."""
    d = {k: kw.get(k) for k in REPORTED}
    d["skeletonize_foreground"] = bool(kw.get("skeletonize_foreground", False))
    return d
