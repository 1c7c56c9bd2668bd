#!/bin/bash
# usage (GPU box): tools/quick_bench.sh [ENV=VAL ...] -> ms/update-step of the four bench configurations
for a in "--dtype bf16s --batch 1024 --steps 40" "--dtype bf16s --batch 256 --steps 100" "--dtype f32 --steps 100" "--dtype f32 --batch 1024 --steps 30"; do
echo "== $* bench $a: $(env "$@" python bench.py --no-cpu-baseline --no-kernel-rooflines $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
done
