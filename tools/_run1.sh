cd $GRAFT_REPO_ROOT
python -m pytest tests/test_many_ids.py -x -q -m gpu > gpurun_out/r03a_ids.txt 2>&1
tail -3 gpurun_out/r03a_ids.txt
PPP_COVER_TRACE=1 timeout 900 python3 bench.py --workload synth512_p9 --flags shipped --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03a_synth512_shipped.json 2> gpurun_out/r03a_synth512_shipped.err
tail -c 3000 gpurun_out/r03a_synth512_shipped.json
tail -5 gpurun_out/r03a_synth512_shipped.err
