#!/bin/bash
# What a resident wave per SIMD is worth to S1: the shipped kernel with unused dynamic LDS that lowers
# the waves a CU holds (8 -> 6 -> 4 at 9^3).  usage: tools/s1_occupancy.sh <tag> [time_s1_tile args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
out=gpurun_out/${tag}.txt
: > $out
for dyn in 0 7000 20000; do
  PPP_S1_DYNLDS=$dyn python3 tools/time_s1_tile.py "$@" 2>/dev/null | tail -1 | sed "s#^{#{\"PPP_S1_DYNLDS\": $dyn, #" >> $out
done
cat $out
