#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-kernel-rooflines --steps 150 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], d['roofline'])"
bash tools/timeline_bench.sh r05x > /dev/null 2>&1
python tools/timeline_step.py gpurun_out/tl_r05x/timeline.tsv 70 > gpurun_out/tl_r05x/step.txt
head -50 gpurun_out/tl_r05x/step.txt
