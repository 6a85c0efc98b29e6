#!/bin/bash
# PMC passes of the S2 kernel alone (tools/time_s2.py) with counter sets given as arguments (one quoted set per pass).
# usage: tools/pmc_s2_sets.sh <tag> <case> "SET A ..." "SET B ..." ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; case=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out; : > $out/summary.txt
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 tools/time_s2.py --case $case --reps 1 > /dev/null 2>$out/err$i.txt
  python3 tools/summarize_prof.py $out/p$i $out/p$i.txt | grep rank_wg_kernel >> $out/summary.txt
  rm -rf $out/p$i
done
awk '{print $(NF-2), $(NF-1)}' $out/summary.txt
