#!/bin/bash
# Run on the GPU box (gpurun): bench line + rocprofv3 kernel stats + PMC FETCH/WRITE passes of
# the same command; writes everything under gpurun_out/<tag>/ (copy the summaries to profiles/).
# usage: tools/profile_round.sh <tag> [bench args...]
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py "$@" > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py "$@" --no-cpu-baseline > $out/bench_under_rocprof.json 2>> $out/bench.err
python3 tools/summarize_prof.py $out/stats $out/kernel_stats.txt > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>> $out/bench.err
done
mkdir -p $out/pmc && cp -r $out/pmc_FETCH_SIZE/* $out/pmc/ 2>/dev/null; 
python3 - <<PY
import glob, os, shutil
# merge both PMC passes into one summary
import sys
sys.path.insert(0, "tools")
import summarize_prof
tmp = "$out/pmc_all"
os.makedirs(tmp, exist_ok=True)
for i, d in enumerate(["$out/pmc_FETCH_SIZE", "$out/pmc_WRITE_SIZE"]):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        shutil.copy(f, os.path.join(tmp, "%d_%s" % (i, os.path.basename(f))))
summarize_prof.main(tmp, "$out/pmc_fetch_write.txt")
PY
rm -rf $out/pmc $out/pmc_all $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/stats/*/*kernel_trace.csv
head -12 $out/kernel_stats.txt
cat $out/bench.json
