#!/bin/bash
# 512^3 step with and without the ring sweep.  usage: tools/ring_check.sh <tag> [workload]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; wl=${2:-synth512_p9}
for mode in 1 0; do
  PPP_RING=$mode timeout 1200 python bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline --no-variants --no-north-star > gpurun_out/${tag}_ring_$mode.json 2> gpurun_out/${tag}_ring_$mode.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${tag}_ring_$mode.json").read().strip().splitlines()[-1])
    print("ring=$mode", round(d["ms_per_step"]), d["config"].get("instances_crc32"), {k: round(v) for k, v in d["kernel_ms"].items()}, d["workload_stats"].get("ring_z"), d["workload_stats"].get("s1_base_voxels"), d["config"].get("per_rank_peak_hbm_gb"))
    print({k: round(v) for k, v in d["stage_wall_ms"].items()})
except Exception as e:
    print("ring=$mode failed", e); print(open("gpurun_out/${tag}_ring_$mode.err").read()[-1500:])
PY
done
