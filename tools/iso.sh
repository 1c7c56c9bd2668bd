#!/bin/bash
# usage (GPU box): tools/iso.sh <script.py> <tag> -> per-kernel isolated durations
out=$GRAFT_REPO_ROOT/gpurun_out/iso_$2
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace -d $out -o t -- python3 $1 > $out/run.log 2>&1
python3 tools/rocpd_timeline.py $out/t_results.db 400 2>/dev/null | awk -F'\t' 'NR>1{printf "%8.1f %8s %5s %s\n",$2,$5,$6,$7}' > $out/kernels.txt
rm -f $out/t_results.db
grep -E "dwf|tn_direct|pw_nn|gemm" $out/kernels.txt | sort -k4 | uniq -c -f3 | head -40
