#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06q; mkdir -p $o
python bench.py --no-cpu-baseline --no-kernel-rooflines --no-secondary --steps 150 2>/dev/null | tail -1 > $o/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06q/bench.json'))
print({k:d[k] for k in ('ms_per_step','device_ms_per_step','host_enqueue_ms_per_step','device_ms_per_step_blocks')})
PY
python -m pytest tests -q -m gpu -x --durations=15 > $o/gpu_tests.log 2>&1; echo "rc=$?" >> $o/gpu_tests.log
tail -n 25 $o/gpu_tests.log
