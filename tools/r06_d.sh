#!/bin/bash
# round 6: full GPU suite with durations after the switch prune + the wide forward conv, then the bench line with kernel rooflines
mkdir -p gpurun_out/r06d
python -m pytest tests -q -m gpu -x --durations=40 > gpurun_out/r06d/gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> gpurun_out/r06d/gpu_tests.log
tail -n 5 gpurun_out/r06d/gpu_tests.log
python bench.py --no-cpu-baseline --no-secondary --steps 200 > gpurun_out/r06d/bench.log 2>&1
tail -n 1 gpurun_out/r06d/bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['ms_per_step'], d['value'])
for k in d.get('kernel_rooflines', []):
    print(k['us'], k['frac'], k['kernel'][:110])
"
