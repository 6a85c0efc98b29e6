cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/print_s1_kernels.py > gpurun_out/r03n_s1_kernels.txt 2>&1
cat gpurun_out/r03n_s1_kernels.txt | tail -10
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_cli_gpu.py tests/test_minihdf5.py tests/test_blockwise.py -q -m gpu -x > gpurun_out/r03n_tests.txt 2>&1
tail -8 gpurun_out/r03n_tests.txt
