#!/bin/bash
# round 6: collectives 7 -> 4 per update-step (world-1 tax), size rule of the wide convs; engine + DP tests
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06g; mkdir -p $o
python -m pytest tests/test_gpu_dp_rccl.py tests/test_gpu_learner.py tests/test_gpu_paths.py -q -m gpu -k "dp or rccl or world or ranks or determinism or full_size or paths or identical or fused or bench" > $o/tests.log 2>&1; echo "tests rc=$?" >> $o/tests.log
tail -n 3 $o/tests.log
fc() { RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$1 CDRL_FORCE_COLLECTIVES=1 python bench.py --no-cpu-baseline --no-kernel-rooflines --no-secondary --steps 150 2>/dev/null | grep '^{"metric' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
pl() { python bench.py --no-cpu-baseline --no-kernel-rooflines --no-secondary --steps 150 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do echo "plain: $(pl)   forced collectives (world 1): $(fc $((29530+i)))" | tee -a $o/collectives.txt; done
echo "wide bwd off: $(CDRL_PW_X3_WIDE_BWD=0 pl)  both wide off: $(CDRL_PW_X3_WIDE=0 pl)  default: $(pl)" | tee -a $o/collectives.txt
