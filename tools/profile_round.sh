#!/bin/bash
# Run on the GPU box (gpurun): bench line + rocprofv3 kernel stats + PMC FETCH/WRITE passes of
# the same command; writes everything under gpurun_out/<tag>/ (copy the summaries to profiles/):
#   bench.json               the JSON line of `python3 bench.py <args>`
#   kernel_stats.txt         rocprofv3 --kernel-trace --stats summary of the same command
#   pmc_fetch_write.txt      FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes, KiB)
#   meta.json                fingerprint of the kernel sources the profile was taken from + the
#                            true byte counts of the transpose kernel (counter calibration)
# usage: tools/profile_round.sh <tag> [bench args...]
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py "$@" > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py "$@" --no-cpu-baseline --no-north-star --no-variants > $out/bench_under_rocprof.json 2>> $out/bench.err
python3 tools/summarize_prof.py $out/stats $out/kernel_stats.txt > /dev/null
# (the PMC passes also run the library's counter-calibration kernels: a known byte count, read and
# written with the access widths of the hot kernels -- MI355X_MICROARCH.md, "calibrate on a known
# byte count in your own access pattern")
export PPP_BENCH_CALIBRATE=1
export PPP_BENCH_STAGES=0        # (no extra staged step in the counter passes)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-north-star --no-variants > $out/bench_pmc_$c.json 2>> $out/bench.err
done
unset PPP_BENCH_CALIBRATE
# SQ counters of the same command (two more passes: instruction counts / issue, waits + LDS)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/pmc_sq1 -- python3 bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-north-star --no-variants > /dev/null 2>> $out/bench.err
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $out/pmc_sq2 -- python3 bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-north-star --no-variants > /dev/null 2>> $out/bench.err
python3 - <<PY
import glob, os, shutil, sys
sys.path.insert(0, "tools")
import summarize_prof
tmp = "$out/pmc_sq_all"
os.makedirs(tmp, exist_ok=True)
for i, d in enumerate(["$out/pmc_sq1", "$out/pmc_sq2"]):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        shutil.copy(f, os.path.join(tmp, "%d_%s" % (i, os.path.basename(f))))
summarize_prof.main(tmp, "$out/pmc_sq.txt")
PY
rm -rf $out/pmc_sq_all $out/pmc_sq1 $out/pmc_sq2
python3 - <<PY
import glob, json, os, shutil, sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import summarize_prof, bench
tmp = "$out/pmc_all"
os.makedirs(tmp, exist_ok=True)
for i, d in enumerate(["$out/pmc_FETCH_SIZE", "$out/pmc_WRITE_SIZE"]):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        shutil.copy(f, os.path.join(tmp, "%d_%s" % (i, os.path.basename(f))))
summarize_prof.main(tmp, "$out/pmc_fetch_write.txt")
line = json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
ps, vol = line["config"]["patchshape"], line["config"]["volume"]
W = (2 * ps[0] - 1) * (2 * ps[1] - 1) * (2 * ps[2] - 1)
BV = vol[0] * vol[1] * vol[2]
cal = {}
try:
    cal = json.loads(open("$out/bench_pmc_FETCH_SIZE.json").read().strip().splitlines()[-1]).get("counter_calibration_bytes", {})
except Exception:
    pass
json.dump({"src_sha16": bench.source_sha16(), "command": "python3 bench.py $*",
           "workload": line["config"]["workload"], "flag_set": line["config"]["flag_set"],
           "calib_read_bytes": cal.get("read"), "calib_write_bytes": cal.get("write"),
           "calib_read_f32_bytes": cal.get("read_f32"),
           "transpose_true_read_bytes": (W - 1) // 2 * BV * 4.0, "transpose_true_write_bytes": W * BV * 4.0},
          open("$out/meta.json", "w"), indent=1)
PY
rm -rf $out/pmc_all $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/stats/*/*kernel_trace.csv $out/bench_pmc_*.json
unset PPP_BENCH_STAGES
head -12 $out/kernel_stats.txt
cat $out/bench.json
