#!/bin/bash
# round-5 first GPU call: baseline bench of the round-4 kernels on this box, SQ counters, F5 width traffic (item 7d)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
python bench.py --no-cpu-baseline --no-kernel-rooflines --steps 150 2>/dev/null | tail -1 > gpurun_out/r05a/bench_base.json
bash tools/pmc_sq.sh r05a > gpurun_out/r05a/pmc_sq.log 2>&1
bash tools/c3_traffic.sh w360 --width 360 > gpurun_out/r05a/w360_traffic.log 2>&1
python bench.py --no-cpu-baseline --no-kernel-rooflines --width 360 --steps 30 2>/dev/null | tail -1 > gpurun_out/r05a/bench_w360.json
cat gpurun_out/r05a/bench_base.json | cut -c1-400
