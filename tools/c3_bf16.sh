#!/bin/bash
# Configuration 3 (BASELINE.json configs[2]: bf16 MFMA 1x1-conv GEMMs, batch 1024) evidence: bench lines (f32 / bf16-operand mode
# at B = 1024 and 256), kernel trace, HBM traffic (FETCH_SIZE / WRITE_SIZE) and MFMA counters of the bf16 run.
# usage (GPU box): tools/c3_bf16.sh <rNN> -> gpurun_out/c3_<rNN>/
r=$1
cd $GRAFT_REPO_ROOT
out=gpurun_out/c3_$r
mkdir -p $out
for a in "--dtype f32 --batch 1024 --steps 8" "--dtype bf16 --batch 1024 --steps 8" "--dtype f32 --steps 30" "--dtype bf16 --steps 30"; do
  echo "== $a" >> $out/bench_lines.txt
  timeout 600 python bench.py --no-cpu-baseline $a 2>/dev/null | tail -1 >> $out/bench_lines.txt
done
bash tools/prof_bench.sh c3$r --dtype bf16 --batch 1024 > $out/prof.log 2>&1
cp gpurun_out/prof_c3$r/summary.md $out/kernel_trace_summary.md
bash tools/pmc_mfma.sh c3$r --dtype bf16 --batch 1024 > $out/pmc_mfma.log 2>&1
cp gpurun_out/pmc_mfma_c3$r/mfma.json $out/pmc_mfma.json
mkdir -p $out/pmc && cp gpurun_out/pmc_mfma_c3$r/*.txt $out/pmc/
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $ctr -d $out -o step_$ctr -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-rooflines --dtype bf16 --batch 1024 > $out/step_$ctr.log 2>&1
  python3 tools/rocpd_pmc.py $out/step_${ctr}_results.db cdrl > $out/pmc/step_$ctr.txt 2>&1
  rm -f $out/*_results.db
done
tail -n 2 $out/pmc/step_FETCH_SIZE.txt $out/pmc/step_WRITE_SIZE.txt
cat $out/bench_lines.txt | cut -c1-400
head -40 $out/kernel_trace_summary.md
