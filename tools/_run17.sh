cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_tiling.py tests/test_many_ids.py -q -m gpu -x -k "cover or end_to_end or tiles or fresh or many or provider or large" > gpurun_out/r03q_tests.txt 2>&1
tail -6 gpurun_out/r03q_tests.txt
for div in 0 64 256 1024; do
PPP_COVER_SPARSE_DIV=$div timeout 600 python3 bench.py --workload synth256_p9 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r03q_s256_div$div.json 2> gpurun_out/r03q_s256_div$div.err
PPP_COVER_SPARSE_DIV=$div timeout 600 python3 bench.py --workload flylight140_p7 --steps 10 --warmup 3 --no-cpu-baseline --no-variants > gpurun_out/r03q_f140_div$div.json 2> gpurun_out/r03q_f140_div$div.err
done
python3 - <<'PY'
import json
for wl in ("s256","f140"):
  for div in (0,64,256,1024):
    f="r03q_%s_div%d"%(wl,div)
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); c=d["config"]
        print(f, round(d["ms_per_step"],1), c["instances_found"], c["instances_crc32"], "cover", round(d["stage_wall_ms"].get("s3_cover",0),1), d["workload_stats"].get("cover_rounds"))
    except Exception as e: print(f, "ERR", e)
PY
