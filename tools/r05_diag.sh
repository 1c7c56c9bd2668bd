#!/bin/bash
# timing diagnostics (wrong results by design): what the cross-stream events, the side stream's filter gradients and the aux nets cost
cd $GRAFT_REPO_ROOT
run() { echo "$1: $(env $1 python tools/diag_step.py 100 2>/dev/null | tail -1)"; }
for i in 1 2; do
run A=1
run "CDRL_DIAG=1 CDRL_DIAG_NOEV=7"
run "CDRL_DIAG=1 CDRL_DIAG_SKIP_TN=1"
run "CDRL_DIAG=1 CDRL_DIAG_SKIP_TN=1 CDRL_DIAG_NOEV=7"
run "CDRL_DIAG=1 CDRL_DIAG_SKIP_AUX=3"
run "CDRL_SIDE_STREAM=0"
done
