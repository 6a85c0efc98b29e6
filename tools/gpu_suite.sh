#!/bin/bash
# The GPU test suite with per-test durations.  usage: tools/gpu_suite.sh <tag> [pytest args...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
timeout 2400 python -m pytest tests -q -m gpu --durations=15 "$@" > gpurun_out/${tag}_gpu_tests.txt 2>&1
tail -40 gpurun_out/${tag}_gpu_tests.txt
