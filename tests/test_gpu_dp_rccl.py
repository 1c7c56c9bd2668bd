"""Two ranks over RCCL on real GPUs (skipped on boxes with fewer than two devices): replica identity after data-parallel
update-steps, and "all-reduced gradient arena == mean of the shards' local gradients" (SURVEY.md section 8(e))."""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, H, W = 8, 48, 64


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dev = f'cuda:{rank}'
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(dev))
    from carla_driving_rl_agent_amd.engine import LearnerEngine
    from carla_driving_rl_agent_amd.init import init_engine_parameters
    from carla_driving_rl_agent_amd.parallel import DataParallelLearner
    from tests.util import make_batches, to_dev
    eng = LearnerEngine(B, device=dev, H=H, W=W)
    init_engine_parameters(eng, seed=100 + rank)              # deliberately different: broadcast_parameters must fix it
    dp = DataParallelLearner(eng)
    dp.broadcast_parameters()
    pol, val = make_batches(B, H, W, seed=60 + rank)
    dpol, dval = to_dev(pol, dev), to_dev(val, dev)
    moving = {k: v.clone() for k, v in eng.param_views('trunk').items() if 'moving' in k}
    eng.policy_forward_backward(dpol, grad_scale=1.0)          # local shard gradient, unscaled
    local = eng.grads.clone()
    for k, v in eng.param_views('trunk').items():
        if 'moving' in k:
            v.copy_(moving[k])
    for k, v in eng.param_views('policy').items():
        if 'moving' in k:
            v.fill_(0.0 if 'mean' in k else 1.0)
    dp.policy_step(dpol)
    reduced = eng.grads.clone()
    dp.value_step(dval)
    dp.sync_moving_statistics()
    torch.cuda.synchronize()
    torch.save(dict(local=local.cpu(), reduced=reduced.cpu(), params=eng.params.cpu(), m=eng.adam_m.cpu()), os.path.join(out, f'r{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs on one node')
def test_two_rank_rccl_data_parallel(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'r0.pt'), torch.load(tmp_path / 'r1.pt')
    assert torch.equal(r0['params'], r1['params']) and torch.equal(r0['m'], r1['m'])         # replicas identical
    assert torch.equal(r0['reduced'], r1['reduced'])
    from carla_driving_rl_agent_amd.engine import LearnerEngine
    lay = LearnerEngine(B, device=None, H=H, W=W)
    p_off, p_n = lay.region('policy', True)
    t_off, t_n = lay.region('trunk', True)
    lo, hi = p_off, t_off + t_n
    mean = 0.5 * (r0['local'][lo:hi] + r1['local'][lo:hi])
    scale = float(mean.abs().max())
    assert float((r0['reduced'][lo:hi] - mean).abs().max()) <= 2e-6 * scale
