#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05_sb3; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "stem" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace -d $o/iso -o t -- python3 tools/iso_stem.py > $o/iso_run.log 2>&1
python3 tools/rocpd_timeline.py $o/iso/t_results.db 20000 2>/dev/null | awk -F'\t' 'NR>1{a[$7]+=$2; n[$7]++} END{for(k in a) printf "%8.1f us x%3d  %s\n", a[k]/n[k], n[k], k}' | grep "stem_bwd_mfma"
rm -rf $o/iso
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-kernel-rooflines --steps 150 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
