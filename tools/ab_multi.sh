#!/bin/bash
# usage (GPU box): tools/ab_multi.sh "<bench args>" ENV1=VAL ENV2=VAL ...  -- default and every variant in turn, 3 rounds, ONE box
args="$1"; shift
for i in 1 2 3; do
  for e in A=1 "$@"; do
    echo "$e: $(env $e python bench.py --no-cpu-baseline --no-kernel-rooflines --steps 120 $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
  done
done | sort | awk '{k=$1; v=$2; s[k]+=v; n[k]++; l[k]=l[k]" "v} END {for (k in s) printf "%-32s mean %.3f  (%s )\n", k, s[k]/n[k], l[k]}' | sort
