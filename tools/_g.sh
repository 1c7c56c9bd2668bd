cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_bf16_storage.py -q -s -k "unit_backward_against" 2>&1 | grep -a "passed\|failed\|unit-local\|Error\|assert" | cut -c1-1500 > gpurun_out/t8.log
cat gpurun_out/t8.log
