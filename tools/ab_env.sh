#!/bin/bash
# usage (GPU box): tools/ab_env.sh "<ENV=VAL ...>" [bench args]  -- alternates default / with the environment, 3 times each, ONE box
e="$1"; shift
for i in 1 2 3; do
  echo "default: $(python bench.py --no-cpu-baseline --no-kernel-rooflines --steps 150 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
  echo "$e: $(env $e python bench.py --no-cpu-baseline --no-kernel-rooflines --steps 150 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
done
