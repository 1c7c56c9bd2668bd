#!/bin/bash
# HBM traffic of one update-step from PMC counters (separate passes, as MI355X_MICROARCH.md prescribes).
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $out -o cal_$ctr -- python3 tools/pmc_calibrate.py > $out/cal_$ctr.log 2>&1
  python3 tools/rocpd_pmc.py $out/cal_${ctr}_results.db gather_rows > $out/cal_$ctr.txt 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc $ctr -d $out -o step_$ctr -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-rooflines > $out/step_$ctr.log 2>&1
  python3 tools/rocpd_pmc.py $out/step_${ctr}_results.db cdrl > $out/step_$ctr.txt 2>&1
  rm -f $out/*_results.db
done
for f in cal_FETCH_SIZE cal_WRITE_SIZE step_FETCH_SIZE step_WRITE_SIZE; do echo == $f; tail -n 3 $out/$f.txt; done
