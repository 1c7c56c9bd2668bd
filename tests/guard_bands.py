"""Helper of tests/test_gpu_guard.py: one full-size update-step (policy pass + apply, value pass + apply, predict) of an engine created
with CDRL_GUARD=1 in THIS process, then the canary check; prints `bands <bad> <first offset>`.  `poke` additionally overwrites four
bytes of one band first (the detector must see it)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carla_driving_rl_agent_amd.engine import LearnerEngine          # noqa: E402
from carla_driving_rl_agent_amd.init import init_engine_parameters   # noqa: E402
from carla_driving_rl_agent_amd import synthetic                     # noqa: E402

mode, compute = sys.argv[1], sys.argv[2]
B, T = int(os.environ.get('GB_B', 256)), 4
H, W = int(os.environ.get('GB_H', 90)), int(os.environ.get('GB_W', 120))
plain_bytes = LearnerEngine(B, device=None, T=T, H=H, W=W, compute=compute).workspace_bytes if os.environ.get('CDRL_GUARD') != '1' else None
eng = LearnerEngine(B, device='cuda:0', T=T, H=H, W=W, compute=compute)
init_engine_parameters(eng, seed=42)
r = synthetic.make_rollout(B, T=T, H=H, W=W, seed=7)
states = {k: torch.as_tensor(v).cuda() for k, v in r['states'].items()}
adv = torch.as_tensor(np.random.default_rng(1).standard_normal(B).astype(np.float32)).cuda()
speed = (torch.as_tensor(r['speed'][:, 0]) / 100.0).cuda().contiguous()
sim = torch.as_tensor(r['similarity'][:, 0]).cuda().contiguous()
pol = dict(states=states, advantages=adv, old_log_prob=torch.as_tensor(r['old_log_prob']).cuda(), speed=speed, similarity=sim,
           u=torch.as_tensor(r['action']).cuda(), du_da=None, du_db=None)
val = dict(states=states, returns=torch.as_tensor(r['value']).cuda().contiguous(), speed=speed, similarity=sim)      # (B, 2) targets (base, exp)
for _ in range(2):
    eng.policy_forward_backward_resample(pol, seed=3, offset=1)
    eng.policy_apply()
    eng.value_forward_backward(val)
    eng.value_apply()
eng.predict(states)
torch.cuda.synchronize()
assert np.isfinite(eng.metrics('policy')['loss']) and np.isfinite(eng.metrics('value')['loss'])
poked = -1
if mode == 'poke':
    # find a band by its first word (pattern ^ 0) in the first 256 MB of the workspace, write four bytes into its middle
    n = min(eng.workspace.numel(), 2 ** 28) // 4
    words = eng.workspace[:4 * n].view(torch.int32)
    start = int((words == -1513908706).nonzero()[int(sys.argv[3])].item()) * 4          # 0xA5C3961E as int32
    poked = start + 30000
    eng.workspace[poked:poked + 4] = 0
bad, first = eng.check_guards()
print('bands', bad, first, eng.workspace_bytes, poked)
