#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/timeline_bench.sh r06t > gpurun_out/r06t_tl.log 2>&1
python tools/timeline_step.py gpurun_out/tl_r06t/timeline.tsv 100 > gpurun_out/tl_r06t/timeline_step.txt
python tools/timeline_idle.py gpurun_out/tl_r06t/timeline.tsv > gpurun_out/tl_r06t/timeline_idle.txt 2>&1
head -30 gpurun_out/tl_r06t/timeline_step.txt
