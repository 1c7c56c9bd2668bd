#!/bin/bash
# round 6: numbered side-stream records, the critical stream waits for a record CDRL_SIDE_LAG behind the newest one (-1: every claim waits, rounds 1-5)
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06p; mkdir -p $o
bash tools/ab_multi2.sh "CDRL_SIDE_LAG=-1" "CDRL_SIDE_LAG=2" "CDRL_SIDE_LAG=4" "CDRL_SIDE_LAG=6" "CDRL_SIDE_LAG=0" > $o/ab.log 2>&1
cat $o/ab.log
python -m pytest tests/test_gpu_learner.py tests/test_gpu_paths.py -q -m gpu -x -k "not pinned" > $o/eng.log 2>&1; echo "eng rc=$?" >> $o/eng.log
tail -n 4 $o/eng.log
