cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_update_loop.py::test_update_loop_matches_oracle_step_by_step tests/test_gpu_dp_rccl.py -x -q -s 2>&1 | tail -25 > gpurun_out/t1.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 >> gpurun_out/t1.log
CDRL_DIAG_SKIP_TN=1 python bench.py --steps 20 --no-cpu-baseline --no-kernel-rooflines 2>&1 | tail -1 | cut -c1-300 >> gpurun_out/t1.log
CDRL_DIAG=1 CDRL_DIAG_SKIP_TN=1 python bench.py --steps 20 --no-cpu-baseline --no-kernel-rooflines 2>&1 | tail -2 | cut -c1-300 >> gpurun_out/t1.log
cat gpurun_out/t1.log
