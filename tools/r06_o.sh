#!/bin/bash
# round 6: what the remaining event traffic costs (racy diagnostics, tools/diag_step.py): waits of the critical stream / forks / side records skipped;
# and the bench A/B of the round's event work against rounds 1-5's events on the same box
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06o; mkdir -p $o
for i in 1 2 3; do
for e in "CDRL_DIAG=0" "CDRL_DIAG=1 CDRL_DIAG_NOEV=1" "CDRL_DIAG=1 CDRL_DIAG_NOEV=2" "CDRL_DIAG=1 CDRL_DIAG_NOEV=4" "CDRL_DIAG=1 CDRL_DIAG_NOEV=7"; do
echo "$e: $(env $e python tools/diag_step.py 150 2>/dev/null | tail -1)"
done; done > $o/diag.log 2>&1
cat $o/diag.log
bash tools/ab_multi2.sh "CDRL_X=0" "CDRL_TAIL_EVENTS=0 CDRL_EVENT_FENCE=1" > $o/ab.log 2>&1
cat $o/ab.log
