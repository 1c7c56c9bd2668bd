#!/bin/bash
# round 6: shortcut branch of the stride-2 units on the fused conv ops: same-box A/B first, then the full GPU suite + smoke
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06h; mkdir -p $o
bash tools/ab_env.sh "CDRL_FUSED_SC=0" > $o/ab.log 2>&1
cat $o/ab.log
python -m pytest tests -q -m gpu -x --durations=15 > $o/gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $o/gpu_tests.log
tail -n 22 $o/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v Warning | tail -n 3 | tee $o/smoke.log
