"""Data-parallel path on CPU: world_size-2 gloo processes drive the product's DataParallelLearner
(slices, all-reduce, step order) over an oracle-backed stand-in for the GPU engine; the result must
equal ONE oracle learner fed the average of the two shards' gradients."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W, B = 48, 64, 4


class OracleBackedEngine:
    """Implements the engine surface DataParallelLearner touches (region / grads / params /
    *_forward_backward / *_apply) with the CPU oracle, using the real engine's arena layout."""

    def __init__(self, seed):
        from oracle import model as OM
        from oracle.spec import NetConfig, trunk_spec, policy_spec, value_spec
        from carla_driving_rl_agent_amd import synthetic
        from carla_driving_rl_agent_amd.engine import LearnerEngine
        self.layout = LearnerEngine(B, device=None, H=H, W=W)          # host-only: parameter tables + offsets
        cfg = NetConfig(H=H, W=W)
        self.oracle = OM.OracleLearner(cfg, OM.init_params(trunk_spec(cfg), seed + 1), OM.init_params(policy_spec(cfg), seed + 2),
                                       OM.init_params(value_spec(cfg), seed + 3), synthetic.DEFAULT_HP)
        self.params_total = self.layout.params_total
        self.grads = torch.zeros(self.layout.grads_total)
        self.params = torch.zeros(self.layout.params_total)
        self.adam_m = torch.zeros(self.layout.grads_total)
        self.adam_v = torch.zeros(self.layout.grads_total)
        self._pending = None

    def region(self, model, trainable):
        return self.layout.region(model, trainable)

    def _scatter(self, model, grads, scale):
        off, _ = self.region(model, True)
        for e in self.layout.tables[model].entries:
            if e['trainable']:
                self.grads[off + e['offset']: off + e['offset'] + e['numel']] = grads[e['name']].detach().reshape(-1) * scale

    def _gather(self, model):
        off, _ = self.region(model, True)
        return {e['name']: self.grads[off + e['offset']: off + e['offset'] + e['numel']].view(e['shape']).clone()
                for e in self.layout.tables[model].entries if e['trainable']}

    def policy_forward_backward(self, batch, grad_scale=1.0):
        loss, gp, gt, aux = self.oracle.policy_grads(batch)
        self._scatter('policy', gp, grad_scale)
        self._scatter('trunk', gt, grad_scale)
        self._pending = (loss, aux)

    def policy_apply(self):
        loss, aux = self._pending
        self.oracle.policy_step(None, grads=(loss, self._gather('policy'), self._gather('trunk'), aux))

    def value_forward_backward(self, batch, grad_scale=1.0):
        loss, gv, gt, aux = self.oracle.value_grads(batch)
        self._scatter('value', gv, grad_scale)
        self._scatter('trunk', gt, grad_scale)
        self._pending = (loss, aux)

    def value_apply(self):
        loss, aux = self._pending
        self.oracle.value_step(None, grads=(loss, self._gather('value'), self._gather('trunk'), aux))


def _batches(rank):
    from tests.util import make_batches, oracle_batch
    pol, val = make_batches(B, H, W, seed=50 + rank)
    return oracle_batch(pol), oracle_batch(val)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(2)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from carla_driving_rl_agent_amd.parallel import DataParallelLearner
    eng = OracleBackedEngine(seed=7)
    dp = DataParallelLearner(eng, sync_bn_stats=False)
    assert dp.world == world
    pol, val = _batches(rank)
    dp.policy_step(pol)
    torch.save(eng.grads.clone(), os.path.join(out, f'grads_after_policy{rank}.pt'))     # all-reduced arena
    dp.value_step(val)
    torch.save({k: v.detach().clone() for k, v in eng.oracle.trunk.items() if 'moving' not in k}, os.path.join(out, f'trunk{rank}.pt'))
    torch.save({k: v.detach().clone() for k, v in eng.oracle.policy.items() if 'moving' not in k}, os.path.join(out, f'policy{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_two_rank_data_parallel_equals_averaged_gradients(tmp_path, world):
    """world = 3: the 1/world gradient scale is not a power of two (SUM of three 1/3-scaled shard gradients)."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    t0, p0 = torch.load(tmp_path / 'trunk0.pt'), torch.load(tmp_path / 'policy0.pt')
    for r in range(1, world):       # replicas stay identical: same averaged gradients, same Adam update on every rank
        t1, p1 = torch.load(tmp_path / f'trunk{r}.pt'), torch.load(tmp_path / f'policy{r}.pt')
        for k in t0:
            assert torch.equal(t0[k], t1[k]), k
        for k in p0:
            assert torch.equal(p0[k], p1[k]), k
    # reference semantics: one learner, gradients averaged over the two env shards.  Compared on the
    # all-reduced GRADIENT arena (linear in the shard gradients -> exact to rounding); comparing weights
    # after Adam would only measure sign flips of noise-level gradients.
    sys.path.insert(0, ROOT)
    g0 = torch.load(tmp_path / 'grads_after_policy0.pt')
    for r in range(1, world):
        assert torch.equal(g0, torch.load(tmp_path / f'grads_after_policy{r}.pt'))
    lay = OracleBackedEngine(seed=7)
    shards = [OracleBackedEngine(seed=7).oracle for _ in range(world)]
    grads = [sh.policy_grads(_batches(r)[0]) for r, sh in enumerate(shards)]
    for model, idx in (('policy', 1), ('trunk', 2)):
        off, _ = lay.region(model, True)
        gmax = max(float(v.abs().max()) for v in grads[0][idx].values())
        for e in lay.layout.tables[model].entries:
            if not e['trainable']:
                continue
            avg = sum(grads[r][idx][e['name']].detach().double() for r in range(world)).float() / world
            got = g0[off + e['offset']: off + e['offset'] + e['numel']].view(e['shape'])
            assert float((got - avg).abs().max()) <= 1e-5 * gmax, (model, e['name'])   # oracle run-to-run thread noise ~1e-6
    # the value-head region was not part of the policy all-reduce slice and must be untouched (zeros)
    voff, vn = lay.region('value', True)
    assert float(g0[voff:voff + vn].abs().max()) == 0.0


def _worker_sync(rank, world, port, out):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from carla_driving_rl_agent_amd.parallel import DataParallelLearner
    from carla_driving_rl_agent_amd.engine import LearnerEngine

    class Arena:         # the arenas + layout DataParallelLearner's broadcast / statistics sync touch
        def __init__(self):
            self.layout = LearnerEngine(B, device=None, H=H, W=W)
            g = torch.Generator().manual_seed(100 + rank)
            self.params_total = self.layout.params_total
            self.params = torch.randn(self.layout.params_total, generator=g)
            self.grads = torch.zeros(self.layout.grads_total)
            self.adam_m = torch.randn(self.layout.grads_total, generator=g)
            self.adam_v = torch.rand(self.layout.grads_total, generator=g)

        def region(self, model, trainable):
            return self.layout.region(model, trainable)

    eng = Arena()
    dp = DataParallelLearner(eng, sync_bn_stats=True)
    mine = eng.params.clone()
    dp.broadcast_parameters(src=0)
    torch.save(dict(params=eng.params.clone(), m=eng.adam_m.clone(), v=eng.adam_v.clone(), before=mine),
               os.path.join(out, f'bcast{rank}.pt'))
    # rank-local BatchNorm moving statistics (every non-trainable region), identical weights
    for model in ('policy', 'trunk', 'value', 'old_policy'):
        off, n = eng.region(model, False)
        eng.params[off:off + n] = float(rank + 1) + torch.arange(n, dtype=torch.float32) * 1e-3
    dp.sync_moving_statistics()
    torch.save(eng.params.clone(), os.path.join(out, f'sync{rank}.pt'))
    dist.destroy_process_group()


def test_broadcast_parameters_and_moving_statistics_sync(tmp_path):
    """broadcast_parameters: every rank ends on rank 0's weights AND Adam moments; sync_moving_statistics: the BatchNorm
    moving statistics of policy / trunk / value and the old policy's copy become the rank average, while every trainable
    region -- incl. the old-policy WEIGHTS, identical by construction -- is left bit-for-bit alone."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_sync, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    b0, b1 = torch.load(tmp_path / 'bcast0.pt'), torch.load(tmp_path / 'bcast1.pt')
    assert not torch.equal(b0['before'], b1['before'])
    for k in ('params', 'm', 'v'):
        assert torch.equal(b0[k], b1[k]), k
    assert torch.equal(b0['params'], b0['before'])                      # source rank unchanged
    s0, s1 = torch.load(tmp_path / 'sync0.pt'), torch.load(tmp_path / 'sync1.pt')
    assert torch.equal(s0, s1)
    from carla_driving_rl_agent_amd.engine import LearnerEngine
    lay = LearnerEngine(B, device=None, H=H, W=W)
    for model in ('policy', 'trunk', 'value', 'old_policy'):
        off, n = lay.region(model, False)
        expect = 1.5 + torch.arange(n, dtype=torch.float32) * 1e-3          # mean of (1 + x, 2 + x)
        assert torch.allclose(s0[off:off + n], expect, rtol=0, atol=1e-6), model
        toff, tn = lay.region(model, True)
        assert torch.equal(s0[toff:toff + tn], b0['params'][toff:toff + tn]), model
