#!/bin/bash
# round 6: 232-channel forward convs on the split-precision kernel (pw_x3_wide_kernel): op / engine tests, parity cases, same-box A/B
mkdir -p gpurun_out/r06c
python -m pytest tests/test_gpu_ops.py -q -m gpu -k "pwconv_x3" -x > gpurun_out/r06c/ops.log 2>&1; echo "ops rc=$?" >> gpurun_out/r06c/ops.log
python -m pytest tests/test_gpu_learner.py tests/test_gpu_paths.py -q -m gpu -x -k "determinism or full_size or paths or consistent or identical or fused or trunk_forward" > gpurun_out/r06c/eng.log 2>&1; echo "eng rc=$?" >> gpurun_out/r06c/eng.log
bash tools/ab_env.sh "CDRL_PW_X3_WIDE=0" > gpurun_out/r06c/ab.log 2>&1
cat gpurun_out/r06c/ab.log
python -m pytest tests/test_gpu_learner.py -q -m gpu -k "pinned_decisions and (seed or 135 or 360)" --durations=10 > gpurun_out/r06c/pinned.log 2>&1; echo "pinned rc=$?" >> gpurun_out/r06c/pinned.log
tail -n 3 gpurun_out/r06c/ops.log; tail -n 3 gpurun_out/r06c/eng.log; tail -n 12 gpurun_out/r06c/pinned.log
