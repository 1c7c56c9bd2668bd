#!/bin/bash
# round 6: hipGraph replay of the four bodies of an update-step, re-measured on the round-6 engine (ROCm 7.2 graph options beside it)
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06r; mkdir -p $o
bash tools/ab_multi2.sh "CDRL_X=0" "CDRL_GRAPH=1" "CDRL_GRAPH=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "CDRL_GRAPH=1 DEBUG_HIP_GRAPH_BATCH_SIZE=256" > $o/ab.log 2>&1
cat $o/ab.log
