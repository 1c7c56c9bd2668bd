cd $GRAFT_REPO_ROOT
out=gpurun_out/t3.log
timeout 900 python -m pytest tests/test_gpu_learner.py tests/test_gpu_paths.py -x -q -k "not 256" 2>&1 | tail -15 > $out
bash tools/ab_env.sh "CDRL_FUSED_BWD=0" >> $out 2>&1
cat $out
