#!/bin/bash
# usage (GPU box): tools/iso.sh <script.py> <tag> [script args] -> per-kernel isolated durations (mean us, launches, grid, workgroup x, name)
s=$1; tag=$2; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/iso_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace -d $out -o t -- python3 $s "$@" > $out/run.log 2>&1
python3 tools/rocpd_timeline.py $out/t_results.db 20000 2>/dev/null | awk -F'\t' 'NR>1{printf "%8.1f %8s %5s %s\n",$2,$5,$6,$7}' > $out/kernels.txt
rm -f $out/t_results.db
grep -E "dw[fs]2?_|tn_direct|pw_nn|pw_x3|pww|pwb|gemm" $out/kernels.txt | awk '{k=$2" "$3" "$4; for(i=5;i<=NF;i++) k=k" "$i; a[k]+=$1; n[k]++} END{for(k in a) printf "%8.1f us x%3d  %s\n", a[k]/n[k], n[k], k}' | sort -k5
