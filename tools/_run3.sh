cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 700 python3 bench.py --workload synth512_p9 --flags shipped --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03c_synth512_shipped.json 2> gpurun_out/r03c_synth512_shipped.err
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r03c_synth512_shipped.json"))
print(d["ms_per_step"], d["config"]["instances_found"], d["config"]["instances_crc32"])
print(json.dumps(d["stage_wall_ms"]))
print(json.dumps(d["kernel_ms"]))
PY
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03c_stats -- python3 bench.py --workload synth512_p9 --flags shipped --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03c_under_rocprof.json 2>> gpurun_out/r03c_synth512_shipped.err
python3 tools/summarize_prof.py gpurun_out/r03c_stats gpurun_out/r03c_kernel_stats.txt | head -30
rm -rf gpurun_out/r03c_stats
