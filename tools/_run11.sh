cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_blockwise.py tests/test_decode.py tests/test_cli_gpu.py tests/test_integration_binding.py -q -m gpu > gpurun_out/r03k_tests.txt 2>&1
tail -15 gpurun_out/r03k_tests.txt
