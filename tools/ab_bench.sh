#!/bin/bash
# usage (GPU box): tools/ab_bench.sh [bench args]  -- alternates ./_old_lib.so (a previous build copied there) and the current library, ONE box
cp carla-driving-rl-agent_amd/libcdrl_hip.so /tmp/new.so
for i in 1 2 3; do
for v in old new; do
  if [ $v = old ]; then cp _old_lib.so carla-driving-rl-agent_amd/libcdrl_hip.so; else cp /tmp/new.so carla-driving-rl-agent_amd/libcdrl_hip.so; fi
  echo "$v: $(python bench.py --no-cpu-baseline --no-kernel-rooflines --steps 100 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
done; done
cp /tmp/new.so carla-driving-rl-agent_amd/libcdrl_hip.so
