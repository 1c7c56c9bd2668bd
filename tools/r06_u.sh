#!/bin/bash
# round 6: 16 scratch slots, the critical stream waits for a record CDRL_SIDE_LAG behind the newest one
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06u; mkdir -p $o
bash tools/ab_multi2.sh "CDRL_SIDE_LAG=4" "CDRL_SIDE_LAG=6" "CDRL_SIDE_LAG=8" "CDRL_SIDE_LAG=10" "CDRL_SIDE_LAG=13" > $o/ab.log 2>&1
cat $o/ab.log
python -c "
from carla_driving_rl_agent_amd.engine import LearnerEngine
e = LearnerEngine(256, device='cuda:0', T=4, H=90, W=120)
print('workspace GB', e.workspace.numel() * e.workspace.element_size() / 1e9 if hasattr(e, 'workspace') else '?')
" 2>&1 | tail -2
