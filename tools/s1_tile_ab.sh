#!/bin/bash
# S1 on one tile box of the resident 512^3 volume: this tree vs the trees under variants/wt_*.  usage: tools/s1_tile_ab.sh <tag> [time_s1_tile args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
out=gpurun_out/${tag}.txt
: > $out
python3 tools/time_s1_tile.py "$@" 2>/dev/null | tail -1 >> $out
for l in variants/*.so; do
  [ -f $l ] && PPP_LIB=$GRAFT_REPO_ROOT/$l python3 tools/time_s1_tile.py "$@" 2>/dev/null | tail -1 | sed "s#\"tree\": \"[^\"]*\"#\"tree\": \"$(basename $l)\"#" >> $out
done
for t in variants/wt_*; do
  [ -d $t ] && PPP_TREE=$GRAFT_REPO_ROOT/$t python3 tools/time_s1_tile.py "$@" 2>/dev/null | tail -1 >> $out
done
cat $out
