cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PPP_BENCH_DUMP_STAGES=1 timeout 600 python3 bench.py --workload dec32x256_p25 --steps 1 --warmup 0 --no-cpu-baseline 2>gpurun_out/r04_o.err | tail -1 > gpurun_out/r04_o_dec32.json
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r04_o_dec32.json"))
print(d["ms_per_step"], {k: round(v) for k,v in d["stage_wall_ms"].items() if v>=1})
for k,v in d["stage_lists_ms"].items():
    if k in ("decode","s5_patch_graph","s1_consensus","s6_label_paint"): print(k, [round(x) for x in v])
PY
