#!/bin/bash
# S1 on one tile box of the resident 512^3 volume: the general kernel against the short classification
# of a clean prediction (ppp_pred_check, round 6).  usage: tools/s1_clean_ab.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
out=gpurun_out/${tag}.txt
: > $out
for mode in 0 1; do
  PPP_S1_CLEAN=$mode python3 tools/time_s1_tile.py "$@" 2>/dev/null | tail -1 | sed "s#^{#{\"PPP_S1_CLEAN\": $mode, #" >> $out
done
cat $out
