#!/bin/bash
# round 6: learned tail events on / off (internal events without the system-scope fence in both), same box
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06n; mkdir -p $o
bash tools/ab_multi2.sh "CDRL_TAIL_EVENTS=0" "CDRL_TAIL_EVENTS=1" "CDRL_TAIL_EVENTS=0 CDRL_EVENT_FENCE=1" > $o/ab.log 2>&1
cat $o/ab.log
python -m pytest tests/test_gpu_learner.py tests/test_gpu_paths.py -q -m gpu -x -k "not pinned" > $o/eng.log 2>&1; echo "eng rc=$?" >> $o/eng.log
tail -n 4 $o/eng.log
