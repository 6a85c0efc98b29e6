cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PPP_COVER_TRACE=1 PPP_COVER_SPARSE_DIV=0 timeout 600 python3 bench.py --workload synth256_p9 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> gpurun_out/r03r_trace256.err
grep "cover rounds" gpurun_out/r03r_trace256.err | awk 'NR%4==1' | head -60
