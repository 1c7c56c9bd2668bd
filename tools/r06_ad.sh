#!/bin/bash
# round 6: small launches folded (one pack launch per pass incl. the W^T transposes, norm fold + head tick inside the optimizer kernels, activation in the split-K reduce)
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06ad; mkdir -p $o
python -m pytest tests/test_gpu_learner.py tests/test_gpu_paths.py tests/test_gpu_update_loop.py tests/test_gpu_ops.py -q -m gpu -x -k "not pinned" > $o/t.log 2>&1; echo "rc=$?" >> $o/t.log
tail -n 5 $o/t.log
bash tools/ab_libs.sh scratch/ab/prev.so scratch/ab/new.so > $o/ab.log 2>&1
cat $o/ab.log
