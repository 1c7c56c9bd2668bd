cd $GRAFT_REPO_ROOT
out=gpurun_out/t9.log
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "pwconv_bwd_fused or pwconv_x3" 2>&1 | tail -3 > $out
timeout 300 python tools/iso_pwb.py 2>&1 | tail -4 >> $out
bash tools/ab_env.sh "CDRL_FUSED_BWD=0" >> $out 2>&1
cat $out
