#!/usr/bin/env python3
"""Times the update-step of the engine under whatever CDRL_* switches are in the environment -- INCLUDING the wrong-result
diagnostic ones that bench.py refuses (CDRL_DIAG=1 CDRL_DIAG_*): timing only, never a benchmark line.  Usage: tools/diag_step.py [steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carla_driving_rl_agent_amd.engine import LearnerEngine          # noqa: E402
from carla_driving_rl_agent_amd.init import init_engine_parameters   # noqa: E402
from carla_driving_rl_agent_amd import synthetic                     # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B, T, H, W = int(os.environ.get('DS_B', 256)), 4, 90, 120
eng = LearnerEngine(B, device='cuda:0', T=T, H=H, W=W, compute=os.environ.get('DS_COMPUTE', 'f32'))
init_engine_parameters(eng, seed=42)
r = synthetic.make_rollout(B, T=T, H=H, W=W, seed=7)
states = {k: torch.as_tensor(v).cuda() for k, v in r['states'].items()}
speed = (torch.as_tensor(r['speed'][:, 0]) / 100.0).cuda().contiguous()
sim = torch.as_tensor(r['similarity'][:, 0]).cuda().contiguous()
adv = torch.as_tensor(np.random.default_rng(1).standard_normal(B).astype(np.float32)).cuda()
pol = dict(states=states, advantages=adv, old_log_prob=torch.as_tensor(r['old_log_prob']).cuda(), speed=speed, similarity=sim,
           u=torch.as_tensor(r['action']).cuda(), du_da=None, du_db=None)
val = dict(states=states, returns=torch.as_tensor(r['value']).cuda().contiguous(), speed=speed, similarity=sim)


def step(i):
    eng.policy_forward_backward_resample(pol, seed=3, offset=i + 1)
    eng.policy_apply()
    eng.value_forward_backward(val)
    eng.value_apply()


for i in range(5):
    step(i)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(steps):
    step(i)
b.record()
torch.cuda.synchronize()
print('ms_per_step %.3f' % (a.elapsed_time(b) / steps))
