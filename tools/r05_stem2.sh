#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05_stem2; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -s -k "stem" 2>&1 | tail -8 | tee $o/ops.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace -d $o/iso -o t -- python3 tools/iso_stem.py > $o/iso_run.log 2>&1
tail -2 $o/iso_run.log
python3 tools/rocpd_timeline.py $o/iso/t_results.db 20000 2>/dev/null | awk -F'\t' 'NR>1{a[$7]+=$2; n[$7]++} END{for(k in a) printf "%8.1f us x%3d  %s\n", a[k]/n[k], n[k], k}' | sort -k4 | tee $o/iso.txt
rm -f $o/iso/t_results.db
bash tools/ab_env.sh CDRL_STEM_RAW=0 2>&1 | tee $o/ab.txt
