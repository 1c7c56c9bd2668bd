#!/bin/bash
# usage (GPU box): tools/timeline_bench.sh <tag> [bench args] -> gpurun_out/tl_<tag>/timeline.tsv (last update-step)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/tl_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace -d $out -o trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-rooflines "$@" > $out/run.log 2>&1
python3 tools/rocpd_timeline.py $out/trace_results.db 3600 > $out/timeline.tsv 2> $out/cols.txt
rm -f $out/trace_results.db
cat $out/cols.txt; wc -l $out/timeline.tsv
