#!/bin/bash
# round 6: small-M split-precision GEMM for the dense layers behind the tower: op / engine tests, same-box A/B, smoke
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06j; mkdir -p $o
python -m pytest tests/test_gpu_ops.py -q -m gpu -k "gemm_x3_rows or gru" > $o/ops.log 2>&1; echo "ops rc=$?" >> $o/ops.log
python -m pytest tests/test_gpu_learner.py tests/test_gpu_paths.py tests/test_gpu_agent.py tests/test_gpu_dp_rccl.py -q -m gpu -k "not pinned and not policy_then" > $o/eng.log 2>&1; echo "eng rc=$?" >> $o/eng.log
tail -n 3 $o/ops.log $o/eng.log
bash tools/ab_env.sh "CDRL_X3_ROWS=0" > $o/ab.log 2>&1
cat $o/ab.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v Warning | tail -n 3 | tee $o/smoke.log
