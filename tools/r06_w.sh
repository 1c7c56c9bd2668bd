#!/bin/bash
# round 6: weight packing of a pass on the side stream beside the stem (CDRL_PACK_SIDE), one stem reduce, wave-per-tensor norm fold, tick folded
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06w; mkdir -p $o
bash tools/ab_multi2.sh "CDRL_PACK_SIDE=1" "CDRL_PACK_SIDE=0" > $o/ab.log 2>&1
cat $o/ab.log
python -m pytest tests/test_gpu_learner.py tests/test_gpu_paths.py tests/test_gpu_update_loop.py -q -m gpu -x -k "not pinned" > $o/eng.log 2>&1; echo "eng rc=$?" >> $o/eng.log
tail -n 4 $o/eng.log
