cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PPP_RANK_WG=1
bash tools/pmc_kernel.sh "rank_wg" r03e_s2_b32 tools/time_s2.py --case 128p9 --reps 1 > gpurun_out/r03e_pmc_s2_b32.txt 2>&1
export PPP_RANK_WG=2
bash tools/pmc_kernel.sh "rank_wg" r03e_s2_b64 tools/time_s2.py --case 128p9 --reps 1 > gpurun_out/r03e_pmc_s2_b64.txt 2>&1
cat gpurun_out/r03e_pmc_s2_b32.txt gpurun_out/r03e_pmc_s2_b64.txt
