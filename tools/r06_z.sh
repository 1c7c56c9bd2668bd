#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06z; mkdir -p $o
python -m pytest tests/test_gpu_paths.py tests/test_gpu_learner.py tests/test_gpu_dp_rccl.py -q -m gpu -x -k "synchronisation or sequence or nccl or two_ranks or determinism" --durations=5 > $o/t.log 2>&1; echo "rc=$?" >> $o/t.log
tail -n 14 $o/t.log
