cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rank or end_to_end or fresh" > gpurun_out/r03b_parity.txt 2>&1
tail -5 gpurun_out/r03b_parity.txt
for c in 140p7 96p9 128p9; do
  for wg in 1 0; do
    PPP_RANK_WG=$wg timeout 300 python3 tools/time_s2.py --case $c >> gpurun_out/r03b_s2.txt 2>&1
  done
done
PPP_RANK_WG_TILE=8x8x16 timeout 300 python3 tools/time_s2.py --case 128p9 >> gpurun_out/r03b_s2.txt 2>&1
PPP_RANK_WG_TILE=8x16x16 timeout 300 python3 tools/time_s2.py --case 140p7 >> gpurun_out/r03b_s2.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03b_s2.txt
timeout 600 python3 bench.py --workload synth512_p9 --flags shipped --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03b_synth512_shipped.json 2> gpurun_out/r03b_synth512_shipped.err
tail -c 2500 gpurun_out/r03b_synth512_shipped.json
tail -5 gpurun_out/r03b_synth512_shipped.err
