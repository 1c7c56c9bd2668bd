#!/bin/bash
# round 6: stem filter gradient recomputing y from its patch tile (CDRL_STEM_RECOMP): op tests, isolated kernel times, same-box A/B, smoke
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06k; mkdir -p $o
python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16_storage.py -q -m gpu -k "stem" > $o/ops.log 2>&1; echo "ops rc=$?" >> $o/ops.log
tail -n 5 $o/ops.log
bash tools/ab_env.sh "CDRL_STEM_RECOMP=0" > $o/ab.log 2>&1
cat $o/ab.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v Warning | tail -n 3 | tee $o/smoke.log
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d $o/prof -o t -- python3 bench.py --no-cpu-baseline --no-kernel-rooflines --no-secondary --steps 30 > $o/prof.log 2>&1
python3 tools/rocpd_summary.py $o/prof/t_results.db 100 2>/dev/null | grep -n "stem\|maxpool\|pool_bn" | head
rm -f $o/prof/t_results.db
