#!/bin/bash
# S2 one-bit masks: parity tests + timing against the two-bit masks.  usage: tools/s2_p1_check.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_large_golden.py -q -m gpu -x -k "rank or kernels_match or golden or scale or large" > gpurun_out/${tag}_tests.txt 2>&1
tail -4 gpurun_out/${tag}_tests.txt
for c in 176p9 140p7; do
  python3 tools/time_s2.py --case $c 2>/dev/null | tail -1
  PPP_RANK_P1=0 python3 tools/time_s2.py --case $c 2>/dev/null | tail -1
done
