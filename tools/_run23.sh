cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_tiling.py tests/test_many_ids.py -q -m gpu -x -k "rows_box or tiles or tiled or provider or many or large or sharded or ranks" > gpurun_out/r03w_tests.txt 2>&1
tail -6 gpurun_out/r03w_tests.txt
for rb in 0 1; do
PPP_ROWS_BOX=$rb timeout 600 python3 bench.py --workload synth256_p9 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03w_s256_rb$rb.json 2> gpurun_out/r03w_s256_rb$rb.err
done
python3 - <<'PY'
import json
for f in ("r03w_s256_rb0","r03w_s256_rb1"):
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); c=d["config"]
        print(f, round(d["ms_per_step"],1), c["instances_found"], c["instances_crc32"], c["parallelism"], {k: round(v) for k,v in d["stage_wall_ms"].items() if k in ("s1_consensus","s5b_consensus","s5c_patch_graph")})
    except Exception as e: print(f, "ERR", e)
PY
