"""The [vote_instances] flag set of the reference's shipped flylight config
(experiments/flylight/setups/setup01/default.toml:114-169 plus overlapping_inst from [model]):
the default used by bench.py, the smoke test and the tests."""
FLYLIGHT = dict(
    patch_threshold=0.5, fc_threshold=0.5, cuda=True, blockwise=False,
    select_patches_for_sparse_data=True, includeSinglePatchCCS=True,
    removeIntersection=False, mws=False, skipThinCover=True,
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
