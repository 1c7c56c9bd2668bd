#!/bin/bash
# what the stem filter gradient (exposed tail of every pass) costs the update-step: timing only, no stem weight gradient
cd $GRAFT_REPO_ROOT
run() { echo "$1: $(env $1 python tools/diag_step.py 100 2>/dev/null | tail -1)"; }
for i in 1 2 3; do
run A=1
run "CDRL_DIAG=1 CDRL_DIAG_SKIP_STEMF=1"
done
