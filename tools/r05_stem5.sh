#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05_stem5; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_paths.py -q -x -s -k "coefficient_free" 2>&1 | tail -4 | tee $o/paths.txt
timeout 900 python -m pytest tests/test_gpu_learner.py -q -x -k "full_size_properties" 2>&1 | tail -3 | tee $o/learner.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x 2>&1 | tail -3 | tee $o/ops.txt
bash tools/ab_env.sh CDRL_STEM_RAW=1 2>&1 | tee $o/ab.txt
