#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05_stem3; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for bits in 0 1 2 3 4 7; do
  CDRL_DIAG=1 CDRL_DIAG_STEMXT=$bits timeout 300 rocprofv3 --kernel-trace -d $o/iso$bits -o t -- python3 tools/iso_stem.py > $o/iso_run$bits.log 2>&1
  echo "== bits $bits"
  python3 tools/rocpd_timeline.py $o/iso$bits/t_results.db 20000 2>/dev/null | awk -F'\t' 'NR>1{a[$7]+=$2; n[$7]++} END{for(k in a) printf "%8.1f us x%3d  %s\n", a[k]/n[k], n[k], k}' | grep stem_xt
  rm -rf $o/iso$bits
done 2>&1 | tee $o/diag.txt
