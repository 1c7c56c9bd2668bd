#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05_stem4; mkdir -p $o
bash tools/ab_env.sh CDRL_STEM_RAW=0 2>&1 | tee $o/ab.txt
timeout 900 python -m pytest tests/test_gpu_learner.py -q -x 2>&1 | tail -3 | tee $o/learner.txt
bash tools/timeline_bench.sh stemraw2 > /dev/null 2>&1
