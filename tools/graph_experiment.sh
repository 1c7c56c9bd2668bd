#!/bin/bash
# hipGraph replay vs eager launches of the update-step, with and without the side / aux streams inside the capture
# (stored-action loss: the re-sampling step takes a per-call Philox offset and is never captured).  -> gpurun_out/graph_experiment.txt
out=$GRAFT_REPO_ROOT/gpurun_out/graph_experiment.txt
: > $out
run() { echo "== $*" >> $out; env "$@" timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --stored-actions 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'host enqueue ms/step', d['host_enqueue_ms_per_step'])" >> $out 2>&1; }
run CDRL_GRAPH=0
run CDRL_GRAPH=1
run CDRL_GRAPH=0 CDRL_SIDE_STREAM=0
run CDRL_GRAPH=1 CDRL_SIDE_STREAM=0
cat $out
