cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python3 bench.py > gpurun_out/r04_l_bench_default.json 2> gpurun_out/r04_l_bench_default.err
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r04_l_bench_default.json"))
print(d["value"], d["unit"], d["ms_per_step"], d["config"]["workload"], d["config"]["instances_found"], d["config"]["instances_crc32"])
print({k: round(v) for k,v in d["stage_wall_ms"].items()})
print({k: round(v) for k,v in d["kernel_ms"].items()})
print(d["roofline"]["frac"], d["roofline"]["avg_ms"], d["roofline_other_kernels"]["rank_patches"]["avg_ms"], d["roofline_other_kernels"]["patch_graph"]["avg_ms"])
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"])
PY
