cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > gpurun_out/t42.log
bash tools/final_round.sh r04 > gpurun_out/final_r04_stdout.log 2>&1
tail -3 gpurun_out/final_r04_stdout.log | cut -c1-300 >> gpurun_out/t42.log
cat gpurun_out/t42.log
