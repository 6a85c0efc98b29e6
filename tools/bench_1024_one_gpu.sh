#!/bin/bash
# 1024^3 / 9^3 on ONE GPU (one step, no warm-up): the stage times that every rank of an 8-GPU job
# would repeat.  Run on the GPU box: tools/bench_1024_one_gpu.sh <tag>
tag=${1:-r04_zd}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1400 python3 bench.py --workload synth1024_p9 --steps 1 --warmup 0 --no-cpu-baseline \
    2>gpurun_out/${tag}.err | tail -1 > gpurun_out/${tag}_bench_synth1024_p9.json
python3 - "$tag" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/{sys.argv[1]}_bench_synth1024_p9.json"))
print(d["ms_per_step"], d["value"])
for k, v in sorted(d.get("stage_wall_ms", {}).items(), key=lambda kv: -kv[1])[:24]:
    print(f"{k:40s} {v:10.1f}")
PY
