#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) + per-stream timeline of one update-step for a configuration-3 run.
# usage (GPU box): tools/c3_traffic.sh <tag> <bench args...>  -> gpurun_out/c3t_<tag>/
tag=$1; shift
cd $GRAFT_REPO_ROOT
out=gpurun_out/c3t_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $ctr -d $out -o step_$ctr -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-rooflines "$@" > $out/step_$ctr.log 2>&1
  python3 tools/rocpd_pmc.py $out/step_${ctr}_results.db cdrl > $out/step_$ctr.txt 2>&1
  rm -f $out/*_results.db
done
tail -n 2 $out/step_FETCH_SIZE.txt $out/step_WRITE_SIZE.txt
bash tools/timeline_bench.sh c3t_$tag "$@" > /dev/null 2>&1
python3 tools/timeline_step.py gpurun_out/tl_c3t_$tag/timeline.tsv 14 > $out/timeline_step.txt 2>&1
head -60 $out/timeline_step.txt
