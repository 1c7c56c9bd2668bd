#!/bin/bash
# round 6, second half (stream synchronisation without system-scope fences / stop events / lagged waits / call sequences): full GPU suite + smoke at HEAD, then the profile set
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06v; mkdir -p $o
python -m pytest tests -q -m gpu --durations=12 > $o/gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $o/gpu_tests.log
tail -n 18 $o/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v Warning | tail -n 3 | tee $o/smoke.log
bash tools/final_round.sh r06 > gpurun_out/final_r06.log 2>&1; tail -n 34 gpurun_out/final_r06.log
