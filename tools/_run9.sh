cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_decode.py -x -q -m gpu > gpurun_out/r03i_decode_tests.txt 2>&1
tail -15 gpurun_out/r03i_decode_tests.txt
timeout 600 python3 bench.py --workload dec96_p7 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r03i_dec96.json 2> gpurun_out/r03i_dec96.err
tail -3 gpurun_out/r03i_dec96.err
PPP_DECODE_FUSED=0 timeout 600 python3 bench.py --workload dec96_p7 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r03i_dec96_torchtail.json 2> gpurun_out/r03i_dec96_torchtail.err
python3 - <<'PY'
import json
for f in ("r03i_dec96","r03i_dec96_torchtail"):
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); c=d["config"]
        print(f, d["ms_per_step"], d["value"], c["instances_found"], c["instances_crc32"], d["stage_wall_ms"].get("decode"), d["workload_stats"].get("n_selected"), d["workload_stats"].get("n_pairs"))
    except Exception as e: print(f, "ERR", e)
PY
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r03i_all_gpu_tests.txt 2>&1
tail -5 gpurun_out/r03i_all_gpu_tests.txt
