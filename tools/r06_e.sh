#!/bin/bash
# round 6: backward-data of the 232-channel convs on the split-precision one-tile kernel: op / engine tests, same-box A/B, parity cases
mkdir -p gpurun_out/r06e
python -m pytest tests/test_gpu_ops.py -q -m gpu -k "pwconv_x3 or pwconv_bn_bwd" > gpurun_out/r06e/ops.log 2>&1; echo "ops rc=$?" >> gpurun_out/r06e/ops.log
python -m pytest tests/test_gpu_learner.py tests/test_gpu_paths.py -q -m gpu -x -k "determinism or full_size or paths or consistent or identical or fused or trunk_forward" > gpurun_out/r06e/eng.log 2>&1; echo "eng rc=$?" >> gpurun_out/r06e/eng.log
python -m pytest tests/test_gpu_bf16_storage.py -q -m gpu -x -k "every_unit" > gpurun_out/r06e/unit.log 2>&1; echo "unit rc=$?" >> gpurun_out/r06e/unit.log
bash tools/ab_env.sh "CDRL_PW_X3_WIDE_BWD=0" > gpurun_out/r06e/ab.log 2>&1
cat gpurun_out/r06e/ab.log
python -m pytest tests/test_gpu_learner.py -q -m gpu -k "pinned_decisions and (seed or 135 or 360)" > gpurun_out/r06e/pinned.log 2>&1; echo "pinned rc=$?" >> gpurun_out/r06e/pinned.log
tail -n 3 gpurun_out/r06e/ops.log; tail -n 3 gpurun_out/r06e/eng.log; tail -n 3 gpurun_out/r06e/unit.log; tail -n 4 gpurun_out/r06e/pinned.log
