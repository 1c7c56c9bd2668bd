#!/bin/bash
# round 6: the 58-channel forward convs of stage 0 on the split-precision kernel (dword-aligned 16-byte buffer loads; CDRL_PW_X3_UNALIGNED)
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06ae; mkdir -p $o
bash tools/ab_multi2.sh "CDRL_PW_X3_UNALIGNED=1" "CDRL_PW_X3_UNALIGNED=0" > $o/ab.log 2>&1
cat $o/ab.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v Warning | tail -n 2 | cut -c1-400 | tee $o/smoke.log
python -m pytest tests/test_gpu_learner.py tests/test_gpu_paths.py tests/test_gpu_update_loop.py -q -m gpu -x -k "not (pinned and 90)" --durations=4 > $o/t.log 2>&1; echo "rc=$?" >> $o/t.log
tail -n 9 $o/t.log
