"""CARLAgent / CARLAMemory / FakeCARLAEnvironment on the native learner.

Surface follows the reference core/carla_agent.py: constructor keywords (:70-72), class-level
DEFAULT_* architectures (:61-68), update() guard (:129-145), policy / value batch tensors
(:323-349), get_*_gradients / apply_*_gradients (:351-388, :430-463), CARLAMemory (:586-596) and
the FakeCARLAEnvironment entry point (:26-52).  evaluate() / record() drive a CARLA simulator and
are outside the learner hot path: they raise instead of pretending.
"""
import os
import warnings

import numpy as np
import torch
import torch.distributed as dist

from ..parallel import DataParallelLearner
from ..rl import utils, spaces
from ..rl.agents.ppo import PPOAgent, PPOMemory
from ..rl.parameters import DynamicParameter
from .networks import CARLANetwork, relu6


class FakeCARLAEnvironment(spaces.Env):
    """Environment with the state / action spaces of a CARLA environment and no simulator.

    Defaults reproduce the reference's class (three-camera image (90, 360, 3) in [-1, 1], road 9,
    vehicle 5, past_control 4, command 6, navigation 10, action Box(-1, 1, (3,)), time_horizon 1).
    Superset for running the learner without CARLA: every size is overridable, step() / reset()
    return seeded synthetic observations stacked over `time_horizon`, and `info_buffer` /
    `reset_info()` exist as CARLAgent.update() expects from the real CARLAEnv."""

    def __init__(self, image_shape=(90, 360, 3), time_horizon=1, num_waypoints=10, vehicle_features=5, road_features=9,
                 num_actions=3, image_range=(-1.0, 1.0), episode_length=None, seed=0):
        super().__init__()
        self.num_waypoints = num_waypoints
        self.NAVIGATION_FEATURES = dict(space=spaces.Box(low=0.0, high=25.0, shape=(num_waypoints,)),
                                        default=np.zeros(shape=num_waypoints, dtype=np.float32))
        self.time_horizon = time_horizon
        self.action_space = spaces.Box(low=-1.0, high=1.0, shape=(num_actions,))
        self.observation_space = spaces.Dict(
            road=spaces.Box(low=0.0, high=15.0, shape=(road_features,)),
            vehicle=spaces.Box(low=-np.inf, high=np.inf, shape=(vehicle_features,)),
            past_control=spaces.Box(low=-1.0, high=1.0, shape=(4,)), command=spaces.Box(low=0.0, high=1.0, shape=(6,)),
            image=spaces.Box(low=image_range[0], high=image_range[1], shape=tuple(image_shape)),
            navigation=self.NAVIGATION_FEATURES['space'])
        self.image_range = image_range
        self.episode_length = episode_length
        self.info_buffer = dict(speed=[], similarity=[])
        self._rng = np.random.default_rng(seed)
        self._t = 0

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def reset_info(self):
        self.info_buffer = dict(speed=[], similarity=[])

    def _observation(self):
        T, sp = self.time_horizon, self.observation_space.spaces
        lo, hi = self.image_range
        obs = dict(image=self._rng.uniform(lo, hi, size=(T,) + sp['image'].shape).astype(np.float32),
                   road=self._rng.integers(0, 2, size=(T,) + sp['road'].shape).astype(np.float32),
                   vehicle=self._rng.uniform(0.0, 1.0, size=(T,) + sp['vehicle'].shape).astype(np.float32),
                   navigation=np.sort(self._rng.uniform(0.0, 25.0, size=(T,) + sp['navigation'].shape), axis=-1).astype(np.float32),
                   past_control=np.zeros((T, 4), np.float32), command=np.zeros((T, 6), np.float32))
        return obs

    def reset(self):
        self._t = 0
        return self._observation()

    def step(self, action):
        self._t += 1
        speed = float(self._rng.uniform(0.0, 30.0))
        similarity = float(self._rng.uniform(-1.0, 1.0))
        self.info_buffer['speed'].append(speed)
        self.info_buffer['similarity'].append(similarity)
        done = self.episode_length is not None and self._t >= self.episode_length
        return self._observation(), speed / 3.0 * abs(similarity), done, {}

    def render(self, mode='human'):
        pass


class CARLAgent(PPOAgent):
    DEFAULT_CONTROL = dict(units=320, num_layers=2, activation=utils.swish6)
    DEFAULT_CONTROL_VALUE = dict(units=320, num_layers=2, activation=utils.swish6)
    DEFAULT_DYNAMICS = dict(road=dict(units=16, num_layers=2, activation=relu6),
                            vehicle=dict(units=16, num_layers=2, activation=relu6),
                            navigation=dict(units=16, num_layers=2, activation=relu6),
                            shufflenet=dict(g=1.0, last_channels=768),
                            rnn=dict(image=256, road=32, vehicle=32, navigation=32),
                            dynamics=dict(units=512))

    def __init__(self, *args, aug_intensity=1.0, clip_norm=(1.0, 1.0, 1.0), name='carla', load_full=True, eta=0.0,
                 dynamics_lr=1e-3, update_dynamics=True, delta=0.0, aux=1.0, resample_actions=True, compute='f32', **kwargs):
        """`resample_actions=True` (default, the reference's behaviour): the policy loss is evaluated on a fresh Beta sample of
        the NEW policy with pathwise gradients, as PolicyNetwork.call does (reference core/networks.py:96-110);
        the sample is drawn on the device.  `False` is the textbook-PPO variant on the stored rollout actions
        (rl/agents/ppo.py:322-325 semantics; deterministic, same cost).
        `compute='bf16'`: the engine's bf16-operand mode (not in the reference; CARLANetwork docstring)."""
        assert aug_intensity >= 0.0
        if not update_dynamics:
            raise NotImplementedError('update_dynamics=False (frozen trunk) is not implemented natively')
        network_spec = dict(kwargs.pop('network', {}))
        network_spec.setdefault('network', CARLANetwork)
        network_spec.setdefault('control_policy', self.DEFAULT_CONTROL)
        network_spec.setdefault('control_value', self.DEFAULT_CONTROL_VALUE)
        network_spec.setdefault('dynamics', self.DEFAULT_DYNAMICS)
        network_spec.setdefault('compute', compute)
        self.should_update_dynamics = update_dynamics
        self.dynamics_path = os.path.join(kwargs.get('weights_dir', 'weights'), name, 'dynamics_model')
        self.load_full = load_full
        head_clip = clip_norm if isinstance(clip_norm, float) or clip_norm is None else tuple(clip_norm[:2])
        super().__init__(*args, name=name, network=network_spec, clip_norm=head_clip, **kwargs)
        self.network: CARLANetwork = self.network
        self.aug_intensity = aug_intensity
        self.delta, self.eta, self.aux = delta, eta, aux                  # stored, unused (as in the reference)
        self.resample_actions = resample_actions
        self._sample_offset = 0
        # the reference computes should_clip_dynamics_grads but never reads it: trunk grads are unclipped (F9)
        self.should_clip_dynamics_grads = isinstance(clip_norm, float) or (clip_norm is not None and len(clip_norm) > 2
                                                                          and isinstance(clip_norm[2], float))
        self.dynamics_lr = DynamicParameter.create(value=dynamics_lr)
        self.dynamics_lr.load(config=self.config.get('dynamics_lr', {}))
        self._augmenter = None
        self._aug_rng = np.random.default_rng(self.seed)
        self._aug_calls = 0
        self._shard, self._info_segments = [self.env], []
        self._init_data_parallel()

    # -- data parallelism (reference loop rl/agents/ppo.py:190-226 inside :464-548) ------------------------------
    def _init_data_parallel(self):
        """One CARLAgent per GPU under torch.distributed: every rank rolls out ITS OWN environment shard (environment and
        rollout-sampler seeds offset by the rank), computes returns / GAE locally (per-episode quantities), runs the same
        number of minibatch steps on its shard with gradients pre-scaled by 1 / world, and the gradient arenas are SUM-
        all-reduced before the identical clip + Adam update (DataParallelLearner).  Parameters are broadcast from rank 0 here
        and after load(); BatchNorm moving statistics are averaged once per update().  Without an initialised process group
        (or at world size 1) nothing changes; CDRL_FORCE_COLLECTIVES=1 runs the collectives at world size 1 as well."""
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if on else 1
        self.rank = dist.get_rank() if on else 0
        force = on and os.environ.get('CDRL_FORCE_COLLECTIVES') == '1'
        self.data_parallel = self.world > 1 or force
        self._dp = {}
        self._dp_force = force
        if not self.data_parallel:
            return
        self._dp_for(self.network.engine).broadcast_parameters()
        if self.rank:
            # rank-disjoint environment shard and action-sampling stream; weights stay identical (same init seed + broadcast)
            if self.seed is not None:
                self.env.seed(self.seed + self.rank)
            self._aug_rng = np.random.default_rng((self.seed or 0) + 7919 * self.rank)
        self.network.sample_rank, self.network.sample_stride = self.rank, self.world

    def _dp_for(self, eng) -> DataParallelLearner:
        """The DataParallelLearner of an engine (the main one, or a ragged-minibatch engine over the same arenas)."""
        dp = self._dp.get(id(eng))
        if dp is None:
            dp = self._dp[id(eng)] = DataParallelLearner(eng, force_collectives=self._dp_force)
        return dp

    def _all_reduce_ints(self, values, op):
        dev = self.device if dist.get_backend() == 'nccl' else 'cpu'
        t = torch.tensor(list(values), dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=op)
        return [int(x) for x in t.tolist()]

    def agree_on_batches(self, policy_batches, value_batches):
        """Every minibatch step is a collective (the gradient all-reduce), so the NUMBER of steps is decided collectively and
        never from a rank's own list: all ranks run the minimum count of FULL minibatches (episodes that ended early on one rank
        drop the surplus minibatches of the others for this update), followed by ONE ragged step only if every rank holds exactly
        one ragged minibatch with the same row count.  The ragged minibatch is found by its row count, not by its position
        (with shuffle_batches it is not the last one), and is moved to the end of the list."""
        if not self.data_parallel:
            return policy_batches, value_batches

        def split(batches):
            full = [b for b in batches if int(b[1].shape[0]) == self.batch_size]
            ragged = [b for b in batches if int(b[1].shape[0]) != self.batch_size]
            # rows of THE ragged minibatch; 0 = none (or several: a pipeline this agent does not produce, never agreed on)
            return full, ragged, (int(ragged[0][1].shape[0]) if len(ragged) == 1 else 0)

        pf, pr, prow = split(policy_batches)
        vf, vr, vrow = split(value_batches)
        sig = [len(pf), len(vf), prow, vrow]
        lo = self._all_reduce_ints(sig, dist.ReduceOp.MIN)
        hi = self._all_reduce_ints(sig, dist.ReduceOp.MAX)
        policy_batches = pf[:lo[0]] + (pr if lo[2] > 0 and lo[2] == hi[2] else [])
        value_batches = vf[:lo[1]] + (vr if lo[3] > 0 and lo[3] == hi[3] else [])
        return policy_batches, value_batches

    def after_update(self):
        if self.data_parallel:
            self._dp_for(self.network.engine).sync_moving_statistics()

    def is_writer(self) -> bool:
        # every rank runs learn() to its end: checkpoints, summaries and traces are written by rank 0 alone
        return not self.data_parallel or self.rank == 0

    def rank_barrier(self, failed: bool = False):
        if not self.data_parallel:
            return
        # the meeting point doubles as a status exchange: MAX over the ranks' failure flags, so that a writer whose save() raised
        # takes every rank down with an error instead of leaving them in a barrier until the process-group timeout
        flag = torch.tensor([1.0 if failed else 0.0], device=self.device if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if float(flag.item()) > 0.0 and not failed:
            raise RuntimeError('data-parallel learn(): the writer rank failed to save the checkpoint (see its traceback)')

    def load(self):
        super().load()
        if getattr(self, 'data_parallel', False):       # (load=True in the constructor runs before _init_data_parallel, which broadcasts itself)
            self._dp_for(self.network.engine).broadcast_parameters()

    def hyper_parameters(self) -> dict:
        hp = super().hyper_parameters()
        hp['dynamics_lr'] = self.dynamics_lr()
        return hp

    # -- update -----------------------------------------------------------------------------------
    def update(self):
        small = len(self.memory) < self.batch_size
        if self.data_parallel:      # a collective decision: one rank skipping alone would leave the others in an all-reduce
            small = bool(self._all_reduce_ints([int(small)], dist.ReduceOp.MAX)[0])
        if small:
            print('[Not updated] memory too small!')
            self._reset_info()
            return
        super().update()
        self._reset_info()

    def _reset_info(self):
        for env in self._shard:
            env.reset_info()
        self._info_segments = []

    def collect(self, shard, *args, **kwargs):
        self._shard = shard
        return super().collect(shard, *args, **kwargs)

    def trajectory_stored(self, env_index, rollout):
        # several environments: remember which slice of WHICH environment's info buffer belongs to the rows just appended
        if len(self._shard) > 1:
            have = len(self._shard[env_index].info_buffer['speed'])
            self._info_segments.append((env_index, have - rollout.env_steps[env_index], rollout.length[env_index]))

    def _info(self, n):
        """speed / similarity targets of the auxiliary heads, one per memory row, from the environments' info buffers
        (reference core/carla_agent.py:328-329,341-347: padded / truncated to the number of rows).  With an environment shard
        the buffers are cut per trajectory, in the order the trajectories were appended."""
        if self._info_segments:
            sp, si = [], []
            for e, start, rows in self._info_segments:
                buf = self._shard[e].info_buffer
                for dst, key in ((sp, 'speed'), (si, 'similarity')):
                    piece = np.asarray(buf[key][start:start + rows], dtype=np.float32)
                    dst.append(np.pad(piece, (0, rows - piece.shape[0])))
            speed_h, sim_h = np.concatenate(sp), np.concatenate(si)
        else:
            speed_h = np.asarray(self.env.info_buffer['speed'], dtype=np.float32)
            sim_h = np.asarray(self.env.info_buffer['similarity'], dtype=np.float32)
        speed = torch.as_tensor(speed_h, device=self.device) / 100.0
        sim = torch.as_tensor(sim_h, device=self.device)
        if speed.shape[0] >= n:
            return speed[:n].contiguous(), sim[:n].contiguous()
        pad = torch.zeros(n - speed.shape[0], device=self.device)
        return torch.cat([speed, pad]), torch.cat([sim, pad])

    def policy_batch_tensors(self):
        states, advantages, actions, log_probabilities = super().policy_batch_tensors()
        speed, similarity = self._info(advantages.shape[0])
        return states, advantages, actions, log_probabilities, speed, similarity

    def value_batch_tensors(self):
        states, returns = super().value_batch_tensors()
        speed, similarity = self._info(returns.shape[0])
        return states, returns, speed, similarity

    def get_policy_gradients(self, batch):
        states, advantages, actions, log_probabilities, speed, similarity = batch
        eng = self._step_engine = self.network.engine_for(advantages.shape[0])
        b = dict(states={k: states[k] for k in ('state_image', 'state_road', 'state_vehicle', 'state_navigation')},
                 advantages=advantages, old_log_prob=log_probabilities, speed=speed, similarity=similarity, u=actions)
        b = eng.stage(b, 'policy')           # fixed addresses -> the captured hipGraph of the step is replayed
        b.update(du_da=None, du_db=None)
        scale = 1.0 / self.world
        if self.resample_actions:
            # Beta(alpha, beta) of the NEW policy is sampled on the device with pathwise Jacobians (rank-disjoint Philox offsets)
            self._sample_offset += 1
            eng.policy_forward_backward_resample(b, seed=self.seed if self.seed is not None else 0,
                                                 offset=self._sample_offset * self.world + self.rank, grad_scale=scale)
        else:
            eng.policy_forward_backward(b, grad_scale=scale)
        if self.data_parallel:
            self._dp_for(eng).reduce_policy_gradients()
        return eng.buffer(2)[0].clone(), 'policy'  # device scalar (copy of CDRL_BUF_METRICS_P[0]); gradients stay in the arena

    def apply_policy_gradients(self, gradients):
        self._step_engine.policy_apply()          # trunk Adam -> clip -> old_policy <- policy -> policy Adam
        return gradients

    def get_value_gradients(self, batch):
        states, returns, speed, similarity = batch
        eng = self._step_engine = self.network.engine_for(returns.shape[0])
        b = dict(states={k: states[k] for k in ('state_image', 'state_road', 'state_vehicle', 'state_navigation')},
                 returns=returns, speed=speed, similarity=similarity)
        eng.value_forward_backward(eng.stage(b, 'value'), grad_scale=1.0 / self.world)
        if self.data_parallel:
            self._dp_for(eng).reduce_value_gradients()
        return eng.buffer(3)[0].clone(), 'value'

    def apply_value_gradients(self, gradients):
        self._step_engine.value_apply()
        return gradients

    # -- rollout helpers ------------------------------------------------------------------------------
    def get_memory(self):
        return CARLAMemory(state_spec=self.state_spec, num_actions=self.num_actions, time_horizon=self.env.time_horizon,
                           device=self.device)

    def preprocess(self):
        """Stacks a list of T per-step observation dicts (real CARLAEnv) or passes a dict of (T, ...)
        arrays through, prefixing keys with 'state_' when needed."""
        def prepare(state):
            if isinstance(state, (list, tuple)):
                keys = state[0].keys()
                state = {k: np.stack([np.asarray(s[k], dtype=np.float32) for s in state], axis=0) for k in keys}
            return {(k if k.startswith('state_') else f'state_{k}'): v for k, v in state.items()}

        alpha = self.aug_intensity
        if alpha <= 0.0:
            return prepare

        from ..rl.augmentations import Augmenter, draw_plan

        def augment_fn(state):
            """CARLAgent.augment (core/carla_agent.py:527-579): the image stack of every observation is augmented with
            probability-gated ops of intensity `aug_intensity`, on the device."""
            state = prepare(state)
            if self._augmenter is None:
                self._augmenter = Augmenter(self.device)
            self._aug_calls += 1
            plan = draw_plan(alpha, self._aug_rng, offset=self._aug_calls)
            state['state_image'] = self._augmenter(state['state_image'], plan)
            return state
        return augment_fn

    def evaluate(self, *args, **kwargs):
        raise NotImplementedError('CARLAgent.evaluate drives a CARLA simulator (collision / waypoint metrics); '
                                  'it is outside the learner hot path this package implements')

    def record(self, *args, **kwargs):
        raise NotImplementedError('CARLAgent.record drives a CARLA simulator; outside the learner hot path')

    def load_weights(self):
        self.network.load_weights(full=self.load_full)

    def save_config(self):
        self.update_config(dynamics_lr=self.dynamics_lr.serialize())
        super().save_config()


class CARLAMemory(PPOMemory):
    """PPOMemory whose state rows carry the `time_horizon` axis: (N, T, ...)."""

    def __init__(self, state_spec: dict, num_actions: int, time_horizon: int, device='cuda:0'):
        super().__init__(state_spec, num_actions, device=device)
        self.time_horizon = time_horizon
