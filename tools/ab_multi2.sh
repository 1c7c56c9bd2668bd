#!/bin/bash
# usage (GPU box): tools/ab_multi2.sh "<ENV ...>" "<ENV ...>" ... -- rotates through the environments (use "X=0" for a no-op default), 3 rounds, ONE box
for i in 1 2 3; do
  for e in "$@"; do
    echo "$e: $(env $e python bench.py --no-cpu-baseline --no-kernel-rooflines --no-secondary --steps 150 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
  done
done
