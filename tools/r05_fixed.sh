#!/bin/bash
cd $GRAFT_REPO_ROOT
export PWB_NEW_ONLY=1 PWB_FIXED=1 CDRL_DIAG=1
for bits in 0 16 32 48 63; do
  CDRL_DIAG_PWB=$bits bash tools/iso.sh tools/iso_pwb.py fx$bits > /dev/null 2>&1
  echo "== bits $bits"
  grep "pwb_kernel" gpurun_out/iso_fx$bits/kernels.txt | awk '{a[int((NR-1)/28)]+=$1; n[int((NR-1)/28)]++; g[int((NR-1)/28)]=$2} END{for(i=0;i<10;i++) printf "run %d grid %s: %.1f us\n", i, g[i], a[i]/n[i]}'
done
