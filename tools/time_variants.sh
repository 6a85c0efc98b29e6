#!/bin/bash
# time one tool under every library in variants/ (+ the shipped one).  usage: tools/time_variants.sh <tag> <tool.py> [args...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; tool=$2; shift 2
out=gpurun_out/${tag}.txt
: > $out
python3 $tool "$@" 2>/dev/null | tail -1 >> $out
for lib in variants/*.so; do
  PPP_LIB=$GRAFT_REPO_ROOT/$lib python3 $tool "$@" 2>/dev/null | tail -1 >> $out
done
cat $out
