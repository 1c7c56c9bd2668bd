"""Data parallelism THROUGH THE AGENT API on CPU: two gloo ranks each build a CARLAgent around an
oracle-backed stand-in for the GPU network and run learn() -- rollout on a rank-own environment shard, local GAE, update() with the
gradient all-reduce inside get_*_gradients and the moving-statistics average after the last minibatch.  Checked: the replicas
end bit-identical, the all-reduced arena equals the mean of the two shards' local gradients, the shards really differ, and every
rank ran the same number of minibatch steps.  (The oracle is only the arithmetic stand-in here; what is under test is the
product's host logic: core/carla_agent.py, rl/agents/ppo.py, parallel.py.  Reference loop: rl/agents/ppo.py:190-226, :464-548.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W, B, T, N = 48, 64, 4, 4, 9
STATE_KEYS = ('state_image', 'state_road', 'state_vehicle', 'state_navigation')


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def _make_network_class(log):
    from tests.test_dp_gloo import OracleBackedEngine
    from carla_driving_rl_agent_amd.rl.networks import Network

    class Engine(OracleBackedEngine):
        """OracleBackedEngine + the few extra entry points the agent calls (stage, buffer, set_hparams, cfg)."""

        def __init__(self, seed):
            super().__init__(seed)
            self.loss = dict(policy=torch.zeros(1), value=torch.zeros(1))

        def stage(self, batch, slot):
            return batch

        def set_hparams(self, **kw):
            for k, v in kw.items():
                self.oracle.hp[k] = v

        def buffer(self, which, shape=None):
            return self.loss['policy' if which == 2 else 'value']

        @staticmethod
        def _oracle_batch(b):
            out = dict(states={k: _np(b['states'][k]) for k in STATE_KEYS})
            for k, v in b.items():
                if k != 'states' and v is not None:
                    out[k] = _np(v).reshape(-1, 1) if k in ('speed', 'similarity') else _np(v)
            if 'u' in out:      # stored-action loss: zero pathwise Jacobians (oracle/model.py::policy_objective)
                out.setdefault('du_da', np.zeros_like(out['u']))
                out.setdefault('du_db', np.zeros_like(out['u']))
            return out

        def policy_forward_backward(self, batch, grad_scale=1.0):
            super().policy_forward_backward(self._oracle_batch(batch), grad_scale)
            self.loss['policy'] = self._pending[0].detach().reshape(1).float()
            log.append(('policy_local', self.grads.clone(), grad_scale))

        def value_forward_backward(self, batch, grad_scale=1.0):
            super().value_forward_backward(self._oracle_batch(batch), grad_scale)
            self.loss['value'] = self._pending[0].detach().reshape(1).float()

    class OracleNetwork(Network):
        """The surface of core/networks.py::CARLANetwork that PPOAgent / CARLAgent touch, on the CPU oracle."""

        def __init__(self, agent, **kwargs):
            super().__init__(agent)
            self.engine = Engine(seed=7)
            self._rng = np.random.default_rng(agent.seed)
            self.sample_rank, self.sample_stride = 0, 1
            self.last_value = torch.zeros((1, 2))

        def engine_for(self, rows):
            assert rows == B, 'the test drops the ragged minibatch'
            return self.engine

        def set_hparams(self, **kw):
            self.engine.set_hparams(**kw)

        def predict(self, inputs):
            alpha, beta, value, _ = self.engine.oracle.predict({k: _np(inputs[k]) for k in STATE_KEYS})
            a, b = alpha.double().numpy(), beta.double().numpy()
            u = np.clip(np.random.default_rng([int(self._rng.integers(2 ** 31)), self.sample_rank]).beta(a, b), 1e-4, 1 - 1e-4)
            from scipy import stats
            logp = stats.beta.logpdf(u, a, b)
            mean, std = a / (a + b), np.sqrt(a * b / ((a + b) ** 2 * (a + b + 1)))
            f = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32))
            return f(u), f(mean), f(std), f(logp), value.float()

        def predict_last_value(self, state, is_terminal=False, **kw):
            if is_terminal:
                return self.last_value
            return self.engine.oracle.predict({k: _np(state[k]) for k in STATE_KEYS})[2].float()

    return OracleNetwork


def _patch_device_helpers():
    """cdrl_gae_returns / cdrl_gather_rows have no CPU form (the product has no CPU fallback): test-side stand-ins."""
    from carla_driving_rl_agent_amd.rl import utils
    from oracle import gae as OG

    def returns_and_advantages(rewards, values_be, gamma, lambda_, scale=2.0, device='cpu'):
        r, v = _np(rewards).astype(np.float32), _np(values_be).astype(np.float32)
        ret, ret_be = OG.compute_returns(r, gamma)
        _, adv_raw, adv = OG.compute_advantages(r, v, gamma, lambda_, scale)
        f = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32))
        return dict(returns=f(ret), returns_be=f(ret_be), advantages_raw=f(adv_raw), advantages=f(adv))

    utils.returns_and_advantages = returns_and_advantages
    utils.gather_rows = lambda src, idx: src.index_select(0, idx.long())


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(2)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    _patch_device_helpers()
    from carla_driving_rl_agent_amd.core import CARLAgent, FakeCARLAEnvironment
    log = []
    env = FakeCARLAEnvironment(image_shape=(H, W, 3), time_horizon=T, num_waypoints=5, vehicle_features=4, num_actions=2,
                               image_range=(0.0, 1.0))
    agent = CARLAgent(env, batch_size=B, log_mode=None, seed=21, skip_data=0, drop_batch_remainder=True, shuffle=True, device='cpu',
                      policy_lr=3e-4, value_lr=3e-4, dynamics_lr=3e-4, gamma=0.99, lambda_=0.95, aug_intensity=0.0,
                      resample_actions=False, network=dict(network=_make_network_class(log)))
    assert agent.data_parallel and agent.world == world and agent.rank == rank
    eng = agent.network.engine
    reduced, steps = [], dict(policy=0, value=0)
    orig_pa, orig_va = agent.apply_policy_gradients, agent.apply_value_gradients

    def policy_apply(g):
        reduced.append(eng.grads.clone())       # the arena as the optimizer sees it: after the all-reduce
        steps['policy'] += 1
        return orig_pa(g)

    def value_apply(g):
        steps['value'] += 1
        return orig_va(g)

    agent.apply_policy_gradients, agent.apply_value_gradients = policy_apply, value_apply
    first_obs = {}
    orig_update = agent.update

    def update():
        first_obs['image'] = agent.memory.states['state_image'][0].clone()
        first_obs['n'] = len(agent.memory)
        orig_update()

    agent.update = update
    agent.learn(episodes=1, timesteps=N, close=False)
    o = eng.oracle
    torch.save(dict(trunk={k: v.detach().clone() for k, v in o.trunk.items()}, policy={k: v.detach().clone() for k, v in o.policy.items()},
                    value={k: v.detach().clone() for k, v in o.value.items()}, local=[x[1] for x in log if x[0] == 'policy_local'],
                    scale=[x[2] for x in log if x[0] == 'policy_local'], reduced=reduced, steps=steps, first=first_obs),
               os.path.join(out, f'agent{rank}.pt'))
    dist.destroy_process_group()


def test_agent_level_data_parallel_world2(tmp_path):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'agent0.pt'), torch.load(tmp_path / 'agent1.pt')
    # every rank: N = 9 rollout steps -> 2 full minibatches of 4 per network (the ragged one is dropped), same count on both
    assert r0['steps'] == r1['steps'] == dict(policy=2, value=2)
    assert r0['first']['n'] == r1['first']['n'] == N
    assert not torch.equal(r0['first']['image'], r1['first']['image'])          # rank-own environment shards
    assert r0['scale'] == r1['scale'] == [0.5, 0.5]                               # gradients pre-scaled by 1 / world
    # the arena the optimizer consumed = SUM of the two 1/2-scaled shard gradients, identical on both ranks
    from carla_driving_rl_agent_amd.engine import LearnerEngine
    lay = LearnerEngine(B, device=None, H=H, W=W)
    lo = lay.region('policy', True)[0]
    hi = lay.region('trunk', True)[0] + lay.region('trunk', True)[1]
    for k in range(2):
        assert torch.equal(r0['reduced'][k], r1['reduced'][k]), k
        mean = r0['local'][k][lo:hi] + r1['local'][k][lo:hi]
        assert not torch.equal(r0['local'][k][lo:hi], r1['local'][k][lo:hi])
        assert float((r0['reduced'][k][lo:hi] - mean).abs().max()) <= 1e-6 * float(mean.abs().max()), k
    # replicas identical after the update: weights of every model AND the BatchNorm moving statistics (averaged once per update())
    for model in ('trunk', 'policy', 'value'):
        for name in r0[model]:
            if 'moving' in name:
                continue        # the oracle stand-in keeps its moving statistics outside the arenas the sync touches
            assert torch.equal(r0[model][name], r1[model][name]), (model, name)


def test_comm_stream_is_not_used_when_the_engine_never_releases_it():
    """under hipGraph replay the engine does not release the communication stream in the middle of the
    backward and reports tail_offset() == the whole trunk; DataParallelLearner must then take the single post-pass all-reduce
    (an early bucket on an un-ordered stream would read gradients the replay has not written yet)."""
    sys.path.insert(0, ROOT)
    from carla_driving_rl_agent_amd.parallel import DataParallelLearner as DP
    t_n = 1000
    assert DP.use_comm_stream(True, 2, False, 400, t_n, True)
    assert not DP.use_comm_stream(True, 2, False, t_n, t_n, True)           # graphs: nothing is final before the end of the pass
    assert not DP.use_comm_stream(True, 1, False, 400, t_n, True)           # single rank, collectives not forced
    assert DP.use_comm_stream(True, 1, True, 400, t_n, True)
    assert not DP.use_comm_stream(False, 2, False, 400, t_n, True)
    assert not DP.use_comm_stream(True, 2, False, 400, t_n, False)          # host engine stand-ins


# ---- the number of minibatch steps is a COLLECTIVE decision, also with unequal shards -------------------------
_AGREE_CASES = [
    # (rows of rank 0's policy minibatches, rank 1's) -> the same for value; expected (full steps, ragged rows or 0)
    ([4, 4, 4, 4, 1], [4, 4, 2], 2, 0),         # the advisor's first replay: A ran 3 steps and B 2 before the fix
    ([4, 4], [4, 3], 1, 0),                     # the second: A 2, B 1
    ([4, 4, 3], [4, 1], 1, 0),                  # 2B+3 against B+1 rows: ragged minibatches of different size are dropped
    ([4, 4], [4, 4, 4], 2, 0),                  # exact multiples, different counts
    ([4, 4, 2], [4, 4, 2], 2, 2),               # equal shards keep the ragged step
    ([4, 2, 4], [2, 4, 4], 2, 2),               # shuffle_batches: the ragged minibatch is found by row count and run last
    ([4, 4, 2], [4, 4], 2, 0),                  # one rank without a ragged minibatch
    ([2], [2], 0, 2),
]


def _agree_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from carla_driving_rl_agent_amd.core import CARLAgent

    class Stub:
        data_parallel, batch_size, device = True, 4, 'cpu'
        _all_reduce_ints = CARLAgent._all_reduce_ints
        agree_on_batches = CARLAgent.agree_on_batches

    stub, got = Stub(), []
    for case in _AGREE_CASES:
        mine = case[rank]
        mk = lambda rows, tag: [((tag, i), torch.zeros(r)) for i, r in enumerate(rows)]
        # value minibatches of the case in REVERSE role (rank 0 takes rank 1's list): both lists are agreed on independently
        p, v = stub.agree_on_batches(mk(mine, 'p'), mk(case[1 - rank], 'v'))
        got.append(([int(b[1].shape[0]) for b in p], [int(b[1].shape[0]) for b in v], [b[0][1] for b in p]))
    torch.save(got, os.path.join(out, f'agree{rank}.pt'))
    dist.destroy_process_group()


def test_minibatch_count_is_agreed_collectively_with_unequal_shards(tmp_path):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_agree_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = torch.load(tmp_path / 'agree0.pt'), torch.load(tmp_path / 'agree1.pt')
    for case, a, b in zip(_AGREE_CASES, g0, g1):
        want = [4] * case[2] + ([case[3]] if case[3] else [])
        # the same row sequence on both ranks, for the policy AND the value list: every all-reduce pairs like with like
        assert a[0] == b[0] == want and a[1] == b[1] == want, (case, a, b)
        # full minibatches keep the rank's own order (the first `min` of them), the ragged one comes last
        full_idx = [i for i, r in enumerate(case[0]) if r == 4][:case[2]]
        rag_idx = [i for i, r in enumerate(case[0]) if r != 4] if case[3] else []
        assert a[2] == full_idx + rag_idx, (case, a)


# ---- a writer rank whose save() raises must take every rank down, not leave them in a barrier --------------------------
def _writer_failure_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from carla_driving_rl_agent_amd.core import CARLAgent

    class Stub:
        data_parallel, device = True, 'cpu'
        rank_barrier = CARLAgent.rank_barrier

    stub, got = Stub(), []
    stub.rank_barrier(failed=False)                     # healthy round: nobody raises
    got.append('ok')
    try:                                                # the writer (rank 0) failed: the other rank raises at the meeting point
        stub.rank_barrier(failed=(rank == 0))
        got.append('writer' if rank == 0 else 'missed')
    except RuntimeError as exc:
        got.append(f'raised: {exc}')
    torch.save(got, os.path.join(out, f'wf{rank}.pt'))
    dist.destroy_process_group()


def test_writer_failure_reaches_every_rank(tmp_path):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_writer_failure_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = torch.load(tmp_path / 'wf0.pt'), torch.load(tmp_path / 'wf1.pt')
    assert g0 == ['ok', 'writer']                       # (the writer re-raises its own exception in learn())
    assert g1[0] == 'ok' and g1[1].startswith('raised: data-parallel learn(): the writer rank failed')


def test_checkpoint_writer_leaves_no_temporary_file_behind_on_failure(tmp_path, monkeypatch):
    from carla_driving_rl_agent_amd import tf_checkpoint as TC
    import numpy as np
    real_replace = os.replace

    def failing_replace(src, dst):
        if dst.endswith('.index'):
            raise OSError('disk full (injected)')
        return real_replace(src, dst)

    monkeypatch.setattr(os, 'replace', failing_replace)
    prefix = str(tmp_path / 'net')
    with pytest.raises(OSError):
        TC.save_checkpoint(prefix, {'layer_with_weights-0/kernel': np.zeros((3, 2), np.float32)})
    assert not [f for f in os.listdir(tmp_path) if '.tmp' in f]
