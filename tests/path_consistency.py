"""Helper of tests/test_gpu_paths.py: runs one policy pass of the engine at the benchmark shape in THIS process (the
fusion switches CDRL_FUSED_* are read once per process) and saves gradients / outputs; `cmp` compares two such files."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == 'run':
    from carla_driving_rl_agent_amd.engine import LearnerEngine
    from carla_driving_rl_agent_amd.init import init_engine_parameters
    from carla_driving_rl_agent_amd import synthetic
    B, T, H, W = int(os.environ.get('PC_B', 256)), 4, 90, 120
    eng = LearnerEngine(B, device='cuda:0', T=T, H=H, W=W)
    init_engine_parameters(eng, seed=42)
    r = synthetic.make_rollout(B, T=T, H=H, W=W, seed=7)
    states = {k: torch.as_tensor(v).cuda() for k, v in r['states'].items()}
    adv = torch.as_tensor(np.random.default_rng(1).standard_normal(B).astype(np.float32)).cuda()
    pol = dict(states=states, advantages=adv, old_log_prob=torch.as_tensor(r['old_log_prob']).cuda(),
               speed=(torch.as_tensor(r['speed'][:, 0]) / 100.0).cuda().contiguous(),
               similarity=torch.as_tensor(r['similarity'][:, 0]).cuda().contiguous(), u=torch.as_tensor(r['action']).cuda(), du_da=None, du_db=None)
    eng.policy_forward_backward(pol)
    torch.cuda.synchronize()
    out = {f'{m}/{k}': v.cpu().clone() for m in ('policy', 'trunk') for k, v in eng.grad_views(m).items()}
    out['loss'] = torch.tensor(eng.metrics('policy')['loss'])
    out['dyn'] = eng.buffer(0, (B, eng.cfg.dyn)).cpu().clone()
    mv = {f'mv/{k}': v.cpu().clone() for k, v in eng.param_views('trunk').items() if 'moving' in k}
    out.update(mv)
    torch.save(out, sys.argv[2])
elif sys.argv[1] == 'steps':
    # PC_STEPS whole update-steps (policy pass, apply, value pass, apply) at the benchmark shape; saves the final parameters, optimizer
    # moments and losses: what the stream-synchronisation modes of the engine must reproduce bit for bit (test_gpu_paths.py)
    from carla_driving_rl_agent_amd.engine import LearnerEngine
    from carla_driving_rl_agent_amd.init import init_engine_parameters
    from carla_driving_rl_agent_amd import synthetic
    import contextlib
    B, T, H, W = int(os.environ.get('PC_B', 256)), 4, 90, 120
    eng = LearnerEngine(B, device='cuda:0', T=T, H=H, W=W)
    init_engine_parameters(eng, seed=42)
    r = synthetic.make_rollout(B, T=T, H=H, W=W, seed=7)
    states = {k: torch.as_tensor(v).cuda() for k, v in r['states'].items()}
    adv = torch.as_tensor(np.random.default_rng(1).standard_normal(B).astype(np.float32)).cuda()
    speed = (torch.as_tensor(r['speed'][:, 0]) / 100.0).cuda().contiguous()
    sim = torch.as_tensor(r['similarity'][:, 0]).cuda().contiguous()
    pol = dict(states=states, advantages=adv, old_log_prob=torch.as_tensor(r['old_log_prob']).cuda(), speed=speed, similarity=sim,
               u=torch.as_tensor(r['action']).cuda(), du_da=None, du_db=None)
    val = dict(states=states, returns=torch.as_tensor(r['value']).cuda().contiguous(), speed=speed, similarity=sim)
    seq = os.environ.get('PC_SEQ', '1') == '1'
    losses = []
    for i in range(int(os.environ.get('PC_STEPS', 12))):
        with (eng.sequence() if seq else contextlib.nullcontext()):
            eng.policy_forward_backward_resample(pol, seed=3, offset=i + 1)
            eng.policy_apply()
            eng.value_forward_backward(val)
            eng.value_apply()
        if i % 4 == 3:      # (a read-back in the middle of the run: the hand-over back to the caller's stream is part of what is tested)
            losses.append(eng.metrics('policy')['loss'])
            losses.append(eng.metrics('value')['loss'])
    torch.cuda.synchronize()
    torch.save({'params': eng.params.cpu().clone(), 'grads': eng.grads.cpu().clone(), 'losses': torch.tensor(losses, dtype=torch.float64)},
               sys.argv[2])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    rows = []
    for k in a:
        x, y = a[k].double(), b[k].double()
        d = (x - y).abs().max().item() / (y.abs().max().item() + 1e-30)
        rows.append((d, k))
    rows.sort(reverse=True)
    print('loss', a['loss'].item(), b['loss'].item())
    print('dyn rel', [r for r in rows if r[1] == 'dyn'])
    print('worst moving-stat', max(r for r in rows if r[1].startswith('mv/')))
    tail = [r for r in rows if not r[1].startswith('trunk/img.') and not r[1].startswith('mv/') and r[1] not in ('loss', 'dyn')]
    tower = [r for r in rows if r[1].startswith('trunk/img.')]
    print('worst tail grads', tail[:4])
    print('worst tower grads', tower[:6])
    print('median tower', sorted(r[0] for r in tower)[len(tower) // 2])
