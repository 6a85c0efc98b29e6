cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 bash tools/profile_round.sh r03_x --steps 2 --warmup 1 > gpurun_out/r03x_profile.log 2>&1
tail -2 gpurun_out/r03x_profile.log | cut -c1-400
cat gpurun_out/r03_x/meta.json
grep -E "calib_|consensus_v3|rank_wg|patch_graph_pa" gpurun_out/r03_x/pmc_fetch_write.txt
head -8 gpurun_out/r03_x/kernel_stats.txt
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/r03x_all_gpu_tests.txt 2>&1
tail -3 gpurun_out/r03x_all_gpu_tests.txt
