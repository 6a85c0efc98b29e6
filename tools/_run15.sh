cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_decode.py -q -m gpu -x -k "fresh or 128_cubed or decode or dense or tail or provider" > gpurun_out/r03o_tests.txt 2>&1
tail -8 gpurun_out/r03o_tests.txt
for b in 4096 16384; do
PPP_DECODE_BATCH=$b timeout 600 python3 bench.py --workload dec96_p7 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03o_dec96_dense_b$b.json 2> gpurun_out/r03o_dec96_dense_b$b.err
done
timeout 1200 python3 bench.py --workload dec256_p7 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03o_dec256_dense.json 2> gpurun_out/r03o_dec256_dense.err
python3 - <<'PY'
import json
for f in ("r03o_dec96_dense_b4096","r03o_dec96_dense_b16384","r03o_dec256_dense"):
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); c=d["config"]
        print(f, d["ms_per_step"], d["value"], c["instances_found"], c["instances_crc32"], d["stage_wall_ms"].get("decode"))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 gpurun_out/r03o_dec256_dense.err
