#!/bin/bash
# round 6: kernel trace + per-stream timeline of the step after the wide convs; collective tax at world 1 before / after coalescing
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06f; mkdir -p $o
python -m pytest tests/test_gpu_ops.py -q -m gpu -k "pwconv_x3_wide_bwd" > $o/ops.log 2>&1; echo "ops rc=$?" >> $o/ops.log
python -m pytest tests/test_gpu_dp_rccl.py -q -m gpu > $o/dp.log 2>&1; echo "dp rc=$?" >> $o/dp.log
tail -n 2 $o/ops.log $o/dp.log
fc() { RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$1 CDRL_FORCE_COLLECTIVES=1 python bench.py --no-cpu-baseline --no-kernel-rooflines --no-secondary --steps 150 2>/dev/null | grep '^{"metric' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
pl() { python bench.py --no-cpu-baseline --no-kernel-rooflines --no-secondary --steps 150 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do echo "plain: $(pl)   forced collectives (world 1): $(fc $((29520+i)))" | tee -a $o/collectives.txt; done
bash tools/prof_bench.sh r06f > $o/prof.log 2>&1
cp gpurun_out/prof_r06f/summary.md $o/kernel_trace_summary.md
bash tools/timeline_bench.sh r06f > $o/timeline.log 2>&1
python tools/timeline_step.py gpurun_out/tl_r06f/timeline.tsv 40 > $o/timeline_step.txt
head -50 $o/timeline_step.txt
