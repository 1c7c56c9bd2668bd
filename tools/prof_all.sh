#!/bin/bash
# usage: tools/prof_all.sh <tag>: kernel trace summary + timeline + PMC traffic + full bench line (with cpu baseline)
tag=$1
cd $GRAFT_REPO_ROOT
bash tools/prof_bench.sh $tag > gpurun_out/prof_$tag.log 2>&1
bash tools/timeline_bench.sh $tag > gpurun_out/tl_$tag.log 2>&1
python3 tools/timeline_step.py gpurun_out/tl_$tag/timeline.tsv > gpurun_out/timeline_$tag.txt 2>&1
bash tools/pmc_bench.sh $tag > gpurun_out/pmc_$tag.log 2>&1
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_$tag.json
tail -3 gpurun_out/pmc_$tag.log; head -c 600 gpurun_out/bench_$tag.json; head -20 gpurun_out/timeline_$tag.txt
