#!/bin/bash
# round 6, first GPU call: the new pinned-parity cases (three seeds at B = 256, Fake-env spaces at 90x360, config 5's 135x180),
# then the headline bench on this box
mkdir -p gpurun_out/r06a
python -m pytest tests/test_gpu_learner.py -q -m gpu -k "pinned_decisions" --durations=15 > gpurun_out/r06a/pinned.log 2>&1
echo "pinned rc=$?" >> gpurun_out/r06a/pinned.log
python bench.py --no-cpu-baseline --no-kernel-rooflines --no-secondary --steps 200 > gpurun_out/r06a/bench.log 2>&1
tail -3 gpurun_out/r06a/pinned.log; tail -1 gpurun_out/r06a/bench.log | cut -c1-400
