#!/bin/bash
# upper bound of what removing the forward BatchNorm finalize launches could give (timing only: stale statistics)
cd $GRAFT_REPO_ROOT
run() { echo "$1: $(env $1 python tools/diag_step.py 100 2>/dev/null | tail -1)"; }
for i in 1 2; do
run A=1
run "CDRL_DIAG=1 CDRL_DIAG_SKIP_FIN=1"
run "CDRL_DIAG=1 CDRL_DIAG_SKIP_FIN=2"
run "CDRL_DIAG=1 CDRL_DIAG_SKIP_FIN=4"
run "CDRL_DIAG=1 CDRL_DIAG_SKIP_FIN=7"
done
bash tools/timeline_bench.sh r05y > /dev/null 2>&1
python tools/timeline_step.py gpurun_out/tl_r05y/timeline.tsv 40 > gpurun_out/tl_r05y/step.txt
sed -n '/queue 3/,/queue 4/p' gpurun_out/tl_r05y/step.txt | head -40
