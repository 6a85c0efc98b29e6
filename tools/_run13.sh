cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_tiling.py -q -m gpu -k "cover or provider or sharded" > gpurun_out/r03m_tests.txt 2>&1
tail -5 gpurun_out/r03m_tests.txt
timeout 1800 python3 bench.py --workload synth1024_p9 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03m_synth1024.json 2> gpurun_out/r03m_synth1024.err
python3 - <<'PY'
import json
for f in ("r03m_synth1024",):
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); c=d["config"]
        print(f, d["ms_per_step"], d["value"], c["instances_found"], c["instances_crc32"], c["parallelism"], c["per_rank_peak_hbm_gb"])
        print(json.dumps(d["stage_wall_ms"])); print(json.dumps(d["workload_stats"]))
    except Exception as e: print(f, "ERR", e)
PY
tail -5 gpurun_out/r03m_synth1024.err
