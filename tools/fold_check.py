#!/usr/bin/env python3
"""Bit-identity check of an engine switch: three update-steps, then a checksum of every parameter / Adam / moving-statistics arena and
of the losses.  Usage: [ENV=...] tools/fold_check.py  (compare the printed lines of two runs)"""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carla_driving_rl_agent_amd.engine import LearnerEngine          # noqa: E402
from carla_driving_rl_agent_amd.init import init_engine_parameters   # noqa: E402
from carla_driving_rl_agent_amd import synthetic                     # noqa: E402

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
losses = []
for i in range(3):
    eng.policy_forward_backward_resample(pol, seed=3, offset=i + 1)
    losses.append(eng.metrics('policy')['loss'])
    eng.policy_apply()
    eng.value_forward_backward(val)
    losses.append(eng.metrics('value')['loss'])
    eng.value_apply()
torch.cuda.synchronize()
h = hashlib.sha256()
for t in (eng.params, eng.adam_m, eng.adam_v, eng.grads):
    h.update(t.cpu().numpy().tobytes())
print('losses', ' '.join(repr(x) for x in losses))
print('sha256', h.hexdigest())
