cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/r03u_all_gpu_tests.txt 2>&1
tail -4 gpurun_out/r03u_all_gpu_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 2400 bash tools/profile_round.sh r03_u --steps 2 --warmup 1 > gpurun_out/r03u_profile.log 2>&1
tail -3 gpurun_out/r03u_profile.log | cut -c1-600
cat gpurun_out/r03_u/meta.json
