#!/bin/bash
# round 6: what the side-stream work costs the critical chain now (racy / wrong-result diagnostics, tools/diag_step.py)
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06x; mkdir -p $o
for i in 1 2; do
for e in "CDRL_DIAG=0" "CDRL_DIAG=1 CDRL_DIAG_SKIP_TN=1" "CDRL_DIAG=1 CDRL_DIAG_SKIP_AUX=3" "CDRL_DIAG=1 CDRL_DIAG_SKIP_STEMF=1" "CDRL_DIAG=1 CDRL_DIAG_SKIP_FIN=7"; do
echo "$e: $(env $e python tools/diag_step.py 150 2>/dev/null | tail -1)"
done; done > $o/diag.log 2>&1
cat $o/diag.log
