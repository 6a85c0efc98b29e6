#!/bin/bash
# bench.py on a list of workloads (1 warm-up + N steps each), one summary line per workload.
# usage: tools/run_workloads.sh <tag> <steps> <workload> [workload ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; steps=$2; shift 2
for w in "$@"; do
  timeout 1500 python3 bench.py --workload $w --steps $steps --warmup 1 --no-cpu-baseline --no-north-star --no-variants 2>gpurun_out/${tag}_$w.err | tail -1 > gpurun_out/${tag}_bench_$w.json
  python3 - $w gpurun_out/${tag}_bench_$w.json <<'PY'
import json, sys
w, f = sys.argv[1:3]
try:
    d = json.load(open(f))
    print(w, round(d["ms_per_step"], 1), "ms", round(d["value"], 3), "Mvox/s", d["config"]["instances_found"],
          {k: round(v) for k, v in d["stage_wall_ms"].items() if v >= 1})
except Exception as e:
    print(w, "ERR", e)
PY
done
