#!/bin/bash
# SQ counters of the one-wave and the two-wave S1 kernel on the 9^3 slab.  usage: tools/s1_v4_sq.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
bash tools/pmc_kernel.sh "consensus_v3_kernel" ${tag}_v3 tools/time_s1.py --case slab9 --reps 1 > gpurun_out/${tag}_v3.txt 2>&1
PPP_S1_V4=1 bash tools/pmc_kernel.sh "consensus_v4_kernel" ${tag}_v4 tools/time_s1.py --case slab9 --reps 1 > gpurun_out/${tag}_v4.txt 2>&1
cat gpurun_out/${tag}_v3.txt gpurun_out/${tag}_v4.txt
