cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_blockwise.py tests/test_decode.py tests/test_cli_gpu.py tests/test_integration_binding.py -x -q -m gpu > gpurun_out/r03j_tests.txt 2>&1
tail -15 gpurun_out/r03j_tests.txt
timeout 1200 python3 bench.py --workload dec256_p7 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03j_dec256.json 2> gpurun_out/r03j_dec256.err
python3 - <<'PY'
import json
for f in ("r03j_dec256",):
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); c=d["config"]
        print(f, d["ms_per_step"], d["value"], c["instances_found"], c["instances_crc32"], c["parallelism"], c["per_rank_peak_hbm_gb"])
        print(json.dumps(d["stage_wall_ms"])); print(json.dumps(d["workload_stats"]))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 gpurun_out/r03j_dec256.err
