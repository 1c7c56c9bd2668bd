#!/bin/bash
# K concurrent engines of B = 256 / K (tools/chain_proxy.py), eager / single-stream graph / three-stream graph -> gpurun_out/chain_proxy.txt
out=$GRAFT_REPO_ROOT/gpurun_out/chain_proxy.txt
: > $out
run() { echo "== $*" >> $out; env "${@:2}" timeout 300 python tools/chain_proxy.py $1 40 2>&1 | tail -1 >> $out; }
run 1 CDRL_GRAPH=0
run 2 CDRL_GRAPH=0
run 4 CDRL_GRAPH=0
run 1 CDRL_GRAPH=1 CDRL_SIDE_STREAM=0
run 2 CDRL_GRAPH=1 CDRL_SIDE_STREAM=0
run 4 CDRL_GRAPH=1 CDRL_SIDE_STREAM=0
run 8 CDRL_GRAPH=1 CDRL_SIDE_STREAM=0
run 2 CDRL_GRAPH=1
run 4 CDRL_GRAPH=1
run 2 CDRL_GRAPH=0 CDRL_SIDE_STREAM=0
run 4 CDRL_GRAPH=0 CDRL_SIDE_STREAM=0
cat $out
