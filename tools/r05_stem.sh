#!/bin/bash
# GPU box: coefficient-free stem filter gradient -- op test, parity, A/B of the update-step, smoke
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05_stem; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -s -k "stem or maxpool or pool" 2>&1 | tail -12 > $o/ops.txt; cat $o/ops.txt
timeout 1200 python -m pytest tests/test_gpu_learner.py tests/test_gpu_paths.py -q -x 2>&1 | tail -6 > $o/learner.txt; cat $o/learner.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $o/smoke.txt
bash tools/ab_env.sh CDRL_STEM_RAW=0 2>&1 | tee $o/ab.txt
