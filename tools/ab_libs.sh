#!/bin/bash
# usage (GPU box): tools/ab_libs.sh <a.so> <b.so> [bench args] -- alternates two builds of the library (staged under scratch/ab/), 3 rounds, ONE box
a=$1; b=$2; shift; shift
cp carla-driving-rl-agent_amd/libcdrl_hip.so /tmp/cur.so
for i in 1 2 3; do
for v in $a $b; do
  cp $v carla-driving-rl-agent_amd/libcdrl_hip.so
  echo "$v: $(python bench.py --no-cpu-baseline --no-kernel-rooflines --no-secondary --steps 150 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
done; done
cp /tmp/cur.so carla-driving-rl-agent_amd/libcdrl_hip.so
