cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in 1 0; do
  PPP_CONS_CACHE=$mode timeout 900 python bench.py --workload synth256_p9 --steps 1 --warmup 1 --no-cpu-baseline --no-variants --slabs 3 --yx 1 2 > gpurun_out/r05_e_grid_$mode.json 2> gpurun_out/r05_e_grid_$mode.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r05_e_grid_$mode.json").read().strip().splitlines()[-1])
    print("cache=$mode", d["ms_per_step"], d["config"].get("instances_crc32"), d.get("kernel_ms"))
except Exception as e:
    print("cache=$mode failed", e); print(open("gpurun_out/r05_e_grid_$mode.err").read()[-1500:])
PY
done
