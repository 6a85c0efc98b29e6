cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_tiling.py -q -m gpu -x -k "rows_box or tiles_reproduce or 128_cubed" > gpurun_out/r03y_tests.txt 2>&1
tail -3 gpurun_out/r03y_tests.txt
for rb in 1 0; do
PPP_ROWS_BOX=$rb timeout 900 python3 bench.py --workload synth512_p9 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03y_s512_rb$rb.json 2> gpurun_out/r03y_s512_rb$rb.err
done
python3 - <<'PY'
import json
for f in ("r03y_s512_rb1","r03y_s512_rb0"):
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); c=d["config"]
        print(f, round(d["ms_per_step"],1), c["instances_found"], c["instances_crc32"], {k: round(v) for k,v in d["stage_wall_ms"].items() if k in ("s1_consensus","s5b_consensus","s5c_patch_graph")})
    except Exception as e: print(f, "ERR", e)
PY
