#!/bin/bash
# round 6: forks through stop events bound to the critical stream's kernels (CDRL_TAIL_EVENTS): determinism / paths tests, same-box A/B, smoke
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06l; mkdir -p $o
bash tools/ab_env.sh "CDRL_TAIL_EVENTS=0" > $o/ab.log 2>&1
cat $o/ab.log
python -m pytest tests/test_gpu_learner.py tests/test_gpu_paths.py -q -m gpu -x -k "not pinned" > $o/eng.log 2>&1; echo "eng rc=$?" >> $o/eng.log
tail -n 4 $o/eng.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v Warning | tail -n 3 | tee $o/smoke.log
