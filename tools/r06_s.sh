#!/bin/bash
# round 6: one hand-over between the caller's stream and the engine's per update-step (cdrl_learner_sequence_begin / _end) instead of four
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06s; mkdir -p $o
bash tools/ab_multi2.sh "CDRL_SEQUENCE=1" "CDRL_SEQUENCE=0" > $o/ab.log 2>&1
cat $o/ab.log
python -m pytest tests/test_gpu_learner.py tests/test_gpu_agent.py tests/test_gpu_dp_rccl.py -q -m gpu -x -k "not pinned" > $o/eng.log 2>&1; echo "eng rc=$?" >> $o/eng.log
tail -n 4 $o/eng.log
