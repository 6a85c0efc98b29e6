cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in 16x8x16 8x16x16 8x8x16; do echo -n "$t "; PPP_RANK_WG_TILE=$t python3 tools/time_s2.py --case 176p9 --reps 2 2>/dev/null | tail -1; done > gpurun_out/r04_k_s2_tiles_176.txt
cat gpurun_out/r04_k_s2_tiles_176.txt
