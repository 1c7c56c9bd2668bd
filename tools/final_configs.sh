#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/final_configs_v31.txt
echo "final secondary configurations (python bench.py --no-cpu-baseline ...), ms_per_step / update-steps/s / roofline" > $out
run() { echo "== $*" >> $out; timeout 600 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline'])" >> $out 2>&1; }
run --steps 30 --warmup 5
run --width 360 --steps 10
run --height 135 --width 180 --steps 10
run --batch 1024 --steps 8
run --batch 64 --steps 30
echo "== RCCL path forced on 1 GPU (CDRL_FORCE_COLLECTIVES=1, torchrun-style env)" >> $out
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 CDRL_FORCE_COLLECTIVES=1 timeout 600 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | grep '^{"metric' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $out 2>&1
echo "== smoke" >> $out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $out 2>&1
cat $out
