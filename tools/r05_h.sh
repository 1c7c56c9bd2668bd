#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -k "pwconv_bwd_fused" 2>&1 | tail -2
bash tools/iso.sh tools/iso_pwb.py pwbr 2>&1 | grep -E "pwb"
bash tools/ab_bench.sh 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-kernel-rooflines --steps 150 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
