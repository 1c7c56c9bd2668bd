#!/bin/bash
# HBM bytes and MFMA utilisation of the isolated 1x1-conv kernels (float32 MFMA / split precision / bf16), PMC passes of their own.
# usage (GPU box): tools/pmc_pw_kernels.sh <tag> -> gpurun_out/pmc_pw_<tag>/summary.txt
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_pw_$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 tools/bench_pw_kernels.py > $out/timing.txt 2>&1
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  name=$(echo $ctr | tr ' ' '+')
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $out -o k_$name -- python3 tools/bench_pw_kernels.py 8 > $out/k_$name.log 2>&1
  python3 tools/rocpd_pmc.py $out/k_${name}_results.db pw_ > $out/k_$name.txt 2>&1
  rm -f $out/*_results.db
done
{ echo "== timing (un-profiled, cold buffers)"; cat $out/timing.txt; for f in $out/k_*.txt; do echo "== $(basename $f)"; grep -v TOTAL $f; done; } > $out/summary.txt
cat $out/summary.txt
