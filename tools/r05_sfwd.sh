#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05_sfwd; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_paths.py -q -x -k "band_staged" 2>&1 | tail -4 | tee $o/paths.txt
bash tools/ab_env.sh CDRL_STEM_FWD_BAND=0 2>&1 | tee $o/ab.txt
bash tools/timeline_bench.sh sfwd > /dev/null 2>&1
grep "stem_fwd" gpurun_out/tl_sfwd/timeline.tsv | tail -4
