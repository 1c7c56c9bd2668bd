cd $GRAFT_REPO_ROOT
out=gpurun_out/iso_pwb.txt
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "pwconv_bwd_fused" 2>&1 | tail -3 > $out
timeout 300 python tools/iso_pwb.py 2>&1 | tail -4 >> $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for d in 0 8; do
  echo "== CDRL_DIAG_PWB=$d" >> $out
  export CDRL_DIAG=1 CDRL_DIAG_PWB=$d PWB_NEW_ONLY=1
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_pwb -o p -- python3 tools/iso_pwb.py > /dev/null 2>&1
  python3 tools/rocpd_summary.py gpurun_out/prof_pwb/p_results.db 30 2>&1 | grep "pwb_" >> $out
  rm -f gpurun_out/prof_pwb/*.db
done
cat $out
