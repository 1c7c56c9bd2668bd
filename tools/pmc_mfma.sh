#!/bin/bash
# MFMA utilisation of one update-step from PMC counters (own passes, --kernel-trace only, program directly after `--`).
# usage (GPU box): tools/pmc_mfma.sh <tag> [bench args] -> gpurun_out/pmc_mfma_<tag>/{mfma.json,*.txt}
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_mfma_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_F32" "SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  name=$(echo $ctr | tr ' ' '+')
  timeout 400 rocprofv3 --kernel-trace --pmc $ctr -d $out -o step_$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-rooflines "$@" > $out/step_$name.log 2>&1
  python3 tools/rocpd_pmc.py $out/step_${name}_results.db cdrl > $out/step_$name.txt 2>&1
  rm -f $out/*_results.db
done
python3 tools/pmc_mfma_summary.py $out "$@" > $out/mfma.json
cat $out/mfma.json | head -60
