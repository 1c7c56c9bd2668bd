#!/bin/bash
# SQ issue / stall counters of one update-step per kernel (own passes, --kernel-trace only, program directly after `--`).
# usage (GPU box): tools/pmc_sq.sh <tag> [bench args] -> gpurun_out/pmc_sq_<tag>/{sq.json,step_*.txt}
# Passes are kept small (<= 4 SQ counters, 8 slots per pass on gfx950) so that one unknown counter name costs one pass only.
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -o 'SQ_[A-Z0-9_]*' | sort -u > $out/sq_counters_available.txt
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  name=$(echo $ctr | tr ' ' '+')
  timeout 400 rocprofv3 --kernel-trace --pmc $ctr -d $out -o step_$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-rooflines "$@" > $out/step_$name.log 2>&1
  python3 tools/rocpd_pmc.py $out/step_${name}_results.db cdrl > $out/step_$name.txt 2>&1
  rm -f $out/*_results.db
done
python3 tools/pmc_sq_summary.py $out "$@" > $out/sq.json
head -80 $out/sq.json
