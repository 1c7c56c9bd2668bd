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


def _worker(rank, world, port, out, backend='nccl'):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    if backend == 'nccl':
        torch.cuda.set_device(rank)
        dev = f'cuda:{rank}'
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(dev))
    else:       # both ranks on device 0, collectives of device tensors through gloo (RCCL refuses two ranks on one device)
        torch.cuda.set_device(0)
        dev = 'cuda:0'
        dist.init_process_group('gloo', rank=rank, world_size=world)
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


def test_two_ranks_on_one_gpu_data_parallel_over_gloo(tmp_path):
    """The same two-rank check on a ONE-GPU box: both ranks run their HIP engines on device 0 and all-reduce the device tensors
    through gloo (RCCL needs one device per rank).  What it exercises that the world-1 NCCL tests cannot: a SUM over two DIFFERENT
    shards -- replicas identical after broadcast + policy step + value step + moving-statistics average, all-reduced arena = mean of
    the two local gradients -- with the real kernels, the communication stream released in the middle of the backward and the
    early buckets (/root/reference has no multi-GPU code: north_star's data-parallel contract)."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path), 'gloo'), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'r0.pt'), torch.load(tmp_path / 'r1.pt')
    assert torch.equal(r0['params'], r1['params']) and torch.equal(r0['m'], r1['m'])         # replicas identical
    assert torch.equal(r0['reduced'], r1['reduced'])
    assert not torch.equal(r0['local'], r1['local'])                                        # the shards really differ
    from carla_driving_rl_agent_amd.engine import LearnerEngine
    lay = LearnerEngine(B, device=None, H=H, W=W)
    p_off, p_n = lay.region('policy', True)
    t_off, t_n = lay.region('trunk', True)
    lo, hi = p_off, t_off + t_n
    mean = 0.5 * (r0['local'][lo:hi] + r1['local'][lo:hi])
    scale = float(mean.abs().max())
    assert float((r0['reduced'][lo:hi] - mean).abs().max()) <= 2e-6 * scale


def _worker_world1(rank, world, port, out, graph):
    """ONE rank, a real NCCL (= RCCL) process group, collectives forced: the communication stream, the early buckets released in
    the middle of the backward, the tower bucket, the moving-statistics reduce -- everything the N-GPU path enqueues."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    if graph:
        os.environ['CDRL_GRAPH'] = '1'
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    from carla_driving_rl_agent_amd.engine import LearnerEngine
    from carla_driving_rl_agent_amd.init import init_engine_parameters
    from carla_driving_rl_agent_amd.parallel import DataParallelLearner
    from tests.util import make_batches, to_dev
    res = {}
    for tag, force in (('plain', False), ('forced', True)):
        eng = LearnerEngine(B, device='cuda:0', H=H, W=W)
        init_engine_parameters(eng, seed=5)
        dp = DataParallelLearner(eng, force_collectives=force)
        if force:
            t_off, t_n = eng.region('trunk', True)
            if graph:       # hipGraph replay never releases the communication stream mid-pass: no early bucket
                assert eng.tail_offset() == t_n and dp._tower == (t_off, t_off + t_n)
                assert dp._comm is None          # ... and no communication stream at all: one all-reduce behind the replayed pass
            else:
                assert dp._comm is not None and 0 < eng.tail_offset() < t_n
                names = [e['name'] for e in eng.tables['trunk'].entries if e['trainable'] and e['offset'] >= eng.tail_offset()]
                assert names and not any(n.startswith('img.') for n in names)
        dp.broadcast_parameters()
        pol, val = make_batches(B, H, W, seed=61)
        dpol, dval = to_dev(pol), to_dev(val)
        steps = []
        for k in range(3):
            dp.update_step(dpol, dval, resample=(11, k))
            torch.cuda.synchronize()
            steps.append(eng.grads.clone().cpu())
        res[tag] = dict(grads=steps, params=eng.params.clone().cpu(), m=eng.adam_m.clone().cpu(), v=eng.adam_v.clone().cpu())
    torch.save(res, os.path.join(out, 'w1.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('graph', [False, True])
def test_world1_nccl_collective_path_is_bit_identical(tmp_path, graph):
    """The data-parallel code path on ONE GPU (the driver's box has one): a world-1 NCCL group in a child process,
    DataParallelLearner(force_collectives=True) -- communication stream, mid-backward release, 5 collectives per
    update-step -- must reproduce the collective-free update-steps bit for bit (SUM over one rank, scale 1/1), over three
    update-steps (the second and third start from weights the first one wrote: a bucket enqueued too early would show).
    graph=True: the same with hipGraph replay of the passes (CDRL_GRAPH=1), where the engine must NOT pull the communication
    stream into the capture."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_world1, args=(1, port, str(tmp_path), graph), nprocs=1, join=True)
    r = torch.load(tmp_path / 'w1.pt')
    for k in range(3):
        assert torch.equal(r['plain']['grads'][k], r['forced']['grads'][k]), k
    for key in ('params', 'm', 'v'):
        assert torch.equal(r['plain'][key], r['forced'][key]), key
    assert torch.isfinite(r['plain']['params']).all()


def _worker_agent_world1(rank, world, port, out):
    """CARLAgent.learn() (rollout on the fake environment, GAE, update() = 3 + 3 minibatch steps with the re-sampled loss) once
    without a process group and once as the single rank of an NCCL group with the collectives forced (CDRL_FORCE_COLLECTIVES=1):
    broadcast at construction, all-reduce inside get_*_gradients, moving-statistics average after update()."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    torch.cuda.set_device(0)
    from carla_driving_rl_agent_amd.core import CARLAgent, FakeCARLAEnvironment

    def run():
        env = FakeCARLAEnvironment(image_shape=(H, W, 3), time_horizon=4, num_waypoints=5, vehicle_features=4, num_actions=2,
                                   image_range=(0.0, 1.0), seed=3)
        agent = CARLAgent(env, batch_size=B, log_mode=None, seed=3, skip_data=0, drop_batch_remainder=False, shuffle=True,
                          policy_lr=3e-4, value_lr=3e-4, dynamics_lr=3e-4, aug_intensity=0.0)
        agent.learn(episodes=1, timesteps=3 * B + 3, close=False)        # 3 full minibatches + a ragged one of 3 rows
        torch.cuda.synchronize()
        eng = agent.network.engine
        return agent, dict(params=eng.params.clone().cpu(), m=eng.adam_m.clone().cpu(), v=eng.adam_v.clone().cpu(),
                           steps=eng.named_buffer('hparams', torch.int32)[10:13].tolist())

    agent, plain = run()
    assert not agent.data_parallel
    del agent
    os.environ['CDRL_FORCE_COLLECTIVES'] = '1'
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    agent, forced = run()
    assert agent.data_parallel and agent.world == 1 and len(agent._dp) == 2        # main engine + the ragged-minibatch engine
    torch.save(dict(plain=plain, forced=forced), os.path.join(out, 'agent_w1.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_agent_level_world1_nccl_is_bit_identical(tmp_path):
    """Data parallelism through the AGENT API (CARLAgent under torch.distributed) on the one GPU of the driver's box: the forced
    world-1 NCCL run of learn() must reproduce the plain agent bit for bit -- weights, Adam moments and step counters."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_agent_world1, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    r = torch.load(tmp_path / 'agent_w1.pt')
    assert r['plain']['steps'] == r['forced']['steps'] == [4, 4, 8]
    for key in ('params', 'm', 'v'):
        assert torch.equal(r['plain'][key], r['forced'][key]), key
    assert torch.isfinite(r['plain']['params']).all()


def _worker_agent_world2_gloo(rank, world, port, out):
    """Two CARLAgents, one GPU, gloo: rank-own environment shards (seed + rank inside the agent), local GAE, update() with the
    gradient all-reduce inside get_*_gradients, moving statistics averaged once per update()."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from carla_driving_rl_agent_amd.core import CARLAgent, FakeCARLAEnvironment
    env = FakeCARLAEnvironment(image_shape=(H, W, 3), time_horizon=4, num_waypoints=5, vehicle_features=4, num_actions=2,
                               image_range=(0.0, 1.0), seed=3)
    agent = CARLAgent(env, batch_size=B, log_mode=None, seed=3, skip_data=0, drop_batch_remainder=False, shuffle=True,
                      policy_lr=3e-4, value_lr=3e-4, dynamics_lr=3e-4, aug_intensity=0.0)
    assert agent.data_parallel and agent.world == 2 and agent.rank == rank
    first = {}
    orig_update = agent.update

    def update():
        first['image'] = agent.memory.states['state_image'][0].clone().cpu()
        orig_update()

    agent.update = update
    agent.learn(episodes=1, timesteps=2 * B + 3, close=False)        # 2 full minibatches + a ragged one of 3 rows, on both ranks
    torch.cuda.synchronize()
    eng = agent.network.engine
    torch.save(dict(params=eng.params.clone().cpu(), m=eng.adam_m.clone().cpu(), v=eng.adam_v.clone().cpu(), first=first['image'],
                    steps=eng.named_buffer('hparams', torch.int32)[10:13].tolist()), os.path.join(out, f'agent_w2_{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_agent_level_world2_on_one_gpu_over_gloo(tmp_path):
    """Data parallelism through the AGENT API with TWO ranks and the real engines (both on device 0, collectives through gloo):
    after learn() -- rollouts on different environment shards, 3 + 3 minibatch steps incl. the ragged one -- the replicas hold
    bit-identical weights, Adam moments, BatchNorm moving statistics and step counters (reference loop rl/agents/ppo.py:190-226
    inside :464-548; the sharding contract is north_star's)."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_agent_world2_gloo, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'agent_w2_0.pt'), torch.load(tmp_path / 'agent_w2_1.pt')
    assert not torch.equal(r0['first'], r1['first'])                 # rank-own environment shards
    assert r0['steps'] == r1['steps'] == [3, 3, 6]
    for key in ('params', 'm', 'v'):
        assert torch.equal(r0[key], r1[key]), key
    assert torch.isfinite(r0['params']).all()


def test_bench_two_ranks_share_one_gpu(tmp_path):
    """`bench.py --gpus 2` end to end on the one GPU of the box: the plain-process form spawns the two ranks through
    torch.distributed.run, every rank builds its engine, broadcasts, runs the timed update-steps between barriers, the elapsed time is
    MAX-reduced and rank 0 prints ONE line with the whole-job figure (CDRL_BENCH_SHARE_DEVICE=1: both ranks on device 0, gloo instead of
    RCCL -- the line is marked `shared_device`; it checks the N > 1 plumbing, not the scaling)."""
    import json
    import subprocess
    env = dict(os.environ, CDRL_BENCH_SHARE_DEVICE='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2', '--batch', '32',
                        '--height', '48', '--width', '64', '--no-cpu-baseline', '--no-kernel-rooflines'], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['parallelism'] == 'dp2' and d['config']['global_batch'] == 64 and d['scaling'] == 'weak'
    assert d['steps'] == 4 and d['value'] > 0 and abs(d['value'] - 2 * 4 / (d['ms_per_step'] * 4e-3)) < 1e-2 * d['value']
    assert 'shared_device' in d and r.stdout.rstrip().splitlines()[-1] == lines[0]          # the JSON line is the last line of the job
    assert all(map(lambda v: v == v, d['final_losses'].values()))
