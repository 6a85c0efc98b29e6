#!/bin/bash
# S1 two-wave kernel: parity tests of the consensus stage + timing against the one-wave kernel.
# usage: tools/s1_v4_check.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
out=gpurun_out/${tag}_s1_v4.txt
: > $out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "kernels_match or golden or reference_layout or consensus" > gpurun_out/${tag}_s1_tests.txt 2>&1
tail -5 gpurun_out/${tag}_s1_tests.txt >> $out
for c in slab9 140p7 96p5; do
  timeout 600 python3 tools/time_s1.py --case $c 2>/dev/null | tail -1 >> $out
  PPP_S1_V4=1 timeout 600 python3 tools/time_s1.py --case $c 2>/dev/null | tail -1 >> $out
done
cat $out
