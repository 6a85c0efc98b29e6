cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r06_final_gpu_tests.txt 2>&1; tail -3 gpurun_out/r06_final_gpu_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python3 bench.py > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err; python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r06_final_bench.json"))
print(d["metric"], d["value"], d["unit"], d["n_gpus"], d["steps"], d["warmup"], d["ms_per_step"], d["scaling"], d["config"]["workload"], d["config"]["flag_set"])
print("roofline", {k: d["roofline"].get(k) for k in ("bound","achieved","peak","frac","traffic","traffic_corrected")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["sample"])
PY
