cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/r03p_all_gpu_tests.txt 2>&1
tail -4 gpurun_out/r03p_all_gpu_tests.txt
timeout 600 python3 bench.py --workload flylight140_p7 --steps 20 --warmup 5 > gpurun_out/r03p_flylight140.json 2> gpurun_out/r03p_flylight140.err
timeout 300 python3 bench.py --workload worm2d_p25 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03p_worm2d.json 2> gpurun_out/r03p_worm2d.err
python3 - <<'PY'
import json
for f in ("r03p_flylight140","r03p_worm2d"):
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); c=d["config"]
        print(f, d["ms_per_step"], d["value"], c["instances_found"], c["instances_crc32"]); print(json.dumps(d.get("kernel_ms")))
    except Exception as e: print(f, "ERR", e)
PY
timeout 2400 bash tools/profile_round.sh r03_p --steps 2 --warmup 1 > gpurun_out/r03p_profile.log 2>&1
tail -30 gpurun_out/r03p_profile.log
