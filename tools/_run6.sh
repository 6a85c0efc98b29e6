cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rank or end_to_end or fresh or cover or thin" > gpurun_out/r03f_parity.txt 2>&1
tail -3 gpurun_out/r03f_parity.txt
for c in 128p9 140p7 96p9; do
  for wg in 3 1; do
    PPP_RANK_WG=$wg timeout 300 python3 tools/time_s2.py --case $c >> gpurun_out/r03f_s2.txt 2>&1
  done
done
PPP_RANK_WG_TILE=8x8x16 timeout 300 python3 tools/time_s2.py --case 128p9 >> gpurun_out/r03f_s2.txt 2>&1
PPP_RANK_WG_TILE=8x16x16 timeout 300 python3 tools/time_s2.py --case 140p7 >> gpurun_out/r03f_s2.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03f_s2.txt
timeout 700 python3 bench.py --workload synth512_p9 --flags shipped --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03f_synth512_shipped.json 2> gpurun_out/r03f_synth512_shipped.err
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r03f_synth512_shipped.json"))
print(d["ms_per_step"], d["config"]["instances_found"], d["config"]["instances_crc32"])
print(json.dumps(d["stage_wall_ms"]))
print(json.dumps(d["kernel_ms"]))
PY
tail -3 gpurun_out/r03f_synth512_shipped.err
