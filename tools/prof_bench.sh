#!/bin/bash
# usage (on the GPU box): tools/prof_bench.sh <tag> [bench args]  -> gpurun_out/prof_<tag>/{summary.md,bench.log}
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | tail -1 > $out/bench.log
timeout 400 rocprofv3 --kernel-trace --stats -d $out -o trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-rooflines "$@" > $out/run.log 2>&1
python3 tools/rocpd_summary.py $out/trace_results.db 45 > $out/summary.md
rm -f $out/trace_results.db
python3 -c "
import json;d=json.loads(open('$out/bench.log').read());print('ms/step',d['ms_per_step'],'value',d['value'],'roofline frac',d['roofline']['frac'])"
head -30 $out/summary.md
