#!/bin/bash
# usage (GPU box): tools/final_round.sh <rNN>  -> gpurun_out/final_<rNN>/ : everything profiles/<rNN>_* is made from
r=$1
cd $GRAFT_REPO_ROOT
out=gpurun_out/final_$r
mkdir -p $out
timeout 900 python bench.py > $out/bench_full.log 2>$out/bench_full.err
tail -1 $out/bench_full.log > $out/bench.json
bash tools/prof_bench.sh $r > $out/prof.log 2>&1
cp gpurun_out/prof_$r/summary.md $out/kernel_trace_summary.md
bash tools/timeline_bench.sh $r > $out/timeline.log 2>&1
python tools/timeline_step.py gpurun_out/tl_$r/timeline.tsv 18 > $out/timeline_step.txt
bash tools/pmc_bench.sh $r > $out/pmc.log 2>&1
mkdir -p $out/pmc && cp gpurun_out/pmc_$r/*.txt $out/pmc/
bash tools/pmc_mfma.sh $r > $out/pmc_mfma.log 2>&1
cp gpurun_out/pmc_mfma_$r/mfma.json $out/pmc_mfma.json
cp gpurun_out/pmc_mfma_$r/*.txt $out/pmc/
# SQ issue / stall counters per kernel and the idle-time itemisation of the critical stream (item 5)
bash tools/pmc_sq.sh $r > $out/pmc_sq.log 2>&1
cp gpurun_out/pmc_sq_$r/sq.json $out/pmc_sq.json
mkdir -p $out/pmc_sq && cp gpurun_out/pmc_sq_$r/step_*.txt $out/pmc_sq/
python tools/timeline_idle.py gpurun_out/tl_$r/timeline.tsv > $out/timeline_idle.txt 2>&1
# F5 three-camera width (90x360): HBM traffic + per-stream timeline + bench line
bash tools/c3_traffic.sh w360$r --width 360 > $out/w360_traffic.log 2>&1
mkdir -p $out/w360 && cp gpurun_out/c3t_w360$r/step_FETCH_SIZE.txt gpurun_out/c3t_w360$r/step_WRITE_SIZE.txt gpurun_out/c3t_w360$r/timeline_step.txt $out/w360/
timeout 600 python bench.py --no-cpu-baseline --no-kernel-rooflines --width 360 --steps 30 2>/dev/null | tail -1 > $out/w360/bench.json
cfg=$out/final_configs.txt
echo "secondary configurations (python bench.py --no-cpu-baseline ...): ms_per_step / update-steps/s / roofline" > $cfg
run() { echo "== $*" >> $cfg; timeout 600 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline'])" >> $cfg 2>&1; }
run --steps 30 --warmup 5
run --steps 30 --warmup 5 --stored-actions
run --width 360 --steps 10
run --height 135 --width 180 --steps 10
run --batch 1024 --steps 8
run --batch 64 --steps 30
echo "== RCCL path (early buckets on the communication stream) forced on 1 GPU (CDRL_FORCE_COLLECTIVES=1, torchrun-style env)" >> $cfg
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 CDRL_FORCE_COLLECTIVES=1 timeout 600 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | grep '^{"metric' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $cfg 2>&1
echo "== isolated kernels (scratch-free: bench.py kernel_rooflines of the default run)" >> $cfg
python3 -c "
import json
d=json.load(open('$out/bench.json'))
for k in d['kernel_rooflines']: print(k['kernel'][:70], k['shape'], k['us'], 'us', k['achieved_GBs'], 'GB/s', k['frac'])
print('cpu_baseline', d.get('cpu_baseline'))
" >> $cfg 2>&1
# configuration 3 (bf16 storage, B = 1024): bench lines, kernel trace, per-stream timeline, HBM traffic
c3=$out/c3
mkdir -p $c3
for a in "--dtype bf16s --batch 1024 --steps 30" "--dtype bf16 --batch 1024 --steps 30" "--dtype f32 --batch 1024 --steps 30" "--dtype bf16s --batch 256 --steps 100"; do
  echo "== $a" >> $c3/bench_lines.txt
  timeout 600 python bench.py --no-cpu-baseline --no-kernel-rooflines $a 2>/dev/null | tail -1 >> $c3/bench_lines.txt
done
bash tools/prof_bench.sh c3$r --dtype bf16s --batch 1024 > $c3/prof.log 2>&1
cp gpurun_out/prof_c3$r/summary.md $c3/kernel_trace_summary.md
bash tools/c3_traffic.sh $r --dtype bf16s --batch 1024 > $c3/traffic.log 2>&1
cp gpurun_out/c3t_$r/step_FETCH_SIZE.txt gpurun_out/c3t_$r/step_WRITE_SIZE.txt gpurun_out/c3t_$r/timeline_step.txt $c3/
python3 bench.py --rollout-rows 2>/dev/null | tail -1 > $out/rollout_rows.json
echo "== smoke" >> $cfg
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v Warning | tail -3 >> $cfg
cat $cfg
