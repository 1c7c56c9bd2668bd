"""CARLANetwork on top of the native learner engine.

Keeps the reference's surface (core/networks.py): module-level `dynamics_layers`, `control_branch`,
`PolicyNetwork`, and `CARLANetwork` with predict / predict_last_value / dynamics_predict(_train) /
update_old_policy / save_weights / load_weights / reset / summary, while the four Keras models
(dynamics, policy, old_policy, value) live in flat device arenas owned by a LearnerEngine.
"""
import os
from typing import Dict

import numpy as np
import torch

from .. import _lib
from ..engine import LearnerEngine
from ..init import init_engine_parameters
from ..rl.networks import Network
from ..rl import utils
from . import architectures as nn


def relu6(x):
    return torch.clamp(x, 0.0, 6.0)


def dynamics_layers(inputs: dict, time_horizon: int, **kwargs) -> dict:
    """Shared-network description: ShuffleNet-v2 tower + road / vehicle / navigation feature nets, one
    GRU per modality, concat -> BatchNorm -> Dense(units).  inputs: {name: per-slice shape}."""
    rnn = kwargs.get('rnn')
    if rnn is None:
        raise ValueError("dynamics_layers needs rnn=dict(image=, road=, vehicle=, navigation=)")
    small = {rnn['road'], rnn['vehicle'], rnn['navigation']}
    if len(small) != 1:
        raise NotImplementedError('road / vehicle / navigation GRUs must share one width')
    feats = {k: nn.feature_net(inputs[f'state_{k}'], time_horizon, **kwargs.get(k, {})) for k in ('road', 'vehicle', 'navigation')}
    if len({f['units'] for f in feats.values()}) != 1:
        raise NotImplementedError('feature nets must share one width')
    return dict(kind='dynamics', time_horizon=time_horizon,
                image=nn.shufflenet_v2(inputs['state_image'], time_horizon, **kwargs.get('shufflenet', {})),
                features=feats, rnn=dict(rnn), units=int(kwargs.get('dynamics', {}).get('units', 32)))


def control_branch(inputs: dict, units: int, num_layers: int, activation=utils.swish6) -> dict:
    """[BatchNorm -> Dense(units, swish6)] x num_layers on the dynamics output."""
    if num_layers != 2 or getattr(activation, '__name__', activation) != 'swish6':
        raise NotImplementedError('control branches are built as 2 x [BN -> Dense(units, swish6)] (reference default)')
    return dict(kind='control_branch', units=int(units))


class PolicyNetwork:
    """Policy head description: control branch + Beta(alpha, beta) + auxiliary speed / similarity."""

    def __init__(self, agent, inputs: dict = None, name='PolicyNetwork', **kwargs):
        self.agent = agent
        self.name = name
        self.spec = dict(branch=control_branch(inputs or {}, **kwargs), num_actions=agent.num_actions)


class CARLANetwork(Network):
    def __init__(self, agent, control_policy: dict, control_value: dict, dynamics: dict, update_dynamics=False, compute='f32'):
        """compute: 'f32' (the reference's arithmetic) or 'bf16' -- bf16 MFMA operands in the tower's 1x1 convolutions
        (include/cdrl.h CDRL_COMPUTE_BF16_OPERANDS, BASELINE.json configs[2]; see DESIGN.md section 7 for what it costs numerically)."""
        super().__init__(agent)
        env = agent.env
        T = env.time_horizon
        spec = agent.state_spec
        for key in ('state_image', 'state_road', 'state_vehicle', 'state_navigation'):
            if key not in spec:
                raise ValueError(f'observation space lacks {key[6:]!r}')
        self.dynamics_spec = dynamics_layers({k: spec[k] for k in spec}, time_horizon=T, **dynamics)
        control_value = dict(control_value)
        self.exp_scale = float(control_value.pop('exponent_scale', 6.0))
        if control_value.pop('components', 1) != 1:
            raise NotImplementedError('value head with components != 1')
        p_branch = control_branch({}, **control_policy)
        v_branch = control_branch({}, **control_value)
        if p_branch['units'] != v_branch['units']:
            raise NotImplementedError('policy and value control branches must share one width')
        img = self.dynamics_spec['image']
        self.cfg = dict(T=T, H=img['image'][0], W=img['image'][1], road=spec['state_road'][0],
                        vehicle=spec['state_vehicle'][0], navigation=spec['state_navigation'][0], A=agent.num_actions,
                        stem=img['stem'], stage_c=img['stage_c'], stage_n=img['stage_n'], last=img['last'],
                        feat=self.dynamics_spec['features']['road']['units'], rnn_image=self.dynamics_spec['rnn']['image'],
                        rnn_small=self.dynamics_spec['rnn']['road'], dyn=self.dynamics_spec['units'],
                        head=p_branch['units'], exp_scale=self.exp_scale, compute=compute)
        self.device = agent.device
        self.engine = LearnerEngine(agent.batch_size, device=self.device, **self.cfg)          # learner minibatches
        self._rollouts = {}                # number of environments E -> inference engine over the same arenas
        self.rollout = self.rollout_for(1)                                                       # B = 1 inference
        self._ragged = {}                  # minibatch rows -> engine for a ragged last minibatch (shares arenas + optimizer)
        init_engine_parameters(self.engine, seed=agent.seed if agent.seed is not None else 0)
        self.last_value = torch.zeros((1, 2), dtype=torch.float32, device=self.device)   # (base, exp) at terminal states
        self.action_index = 0              # Philox offset of the rollout sampler: one stream per predict() call
        self.sample_seed = (agent.seed if agent.seed is not None else 0) + 0x5eed
        self.sample_rank, self.sample_stride = 0, 1      # data-parallel agents: rank-disjoint Philox offsets (CARLAgent._init_data_parallel)
        self.update_dynamics = update_dynamics

    def rollout_for(self, envs: int) -> LearnerEngine:
        """Inference engine for `envs` environments stepped together (one batched forward + one sampling launch per step)."""
        if envs not in self._rollouts:
            self._rollouts[envs] = LearnerEngine(envs, device=self.device, share_with=self.engine, **self.cfg)
        return self._rollouts[envs]

    def engine_for(self, rows: int) -> LearnerEngine:
        """The learner engine for a minibatch of `rows` samples: the main engine, or -- for the ragged last minibatch the
        reference's `batch(drop_remainder=False)` pipeline yields (rl/utils.py:388) -- an engine planned for that size over the
        SAME parameter / gradient / Adam arenas and optimizer counters (built on first use, kept)."""
        if rows == self.engine.cfg.B:
            return self.engine
        if rows not in self._ragged:
            self._ragged[rows] = LearnerEngine(rows, device=self.device, share_with=self.engine, **self.cfg)
        return self._ragged[rows]

    # -- hyper-parameters -----------------------------------------------------------------------
    def set_hparams(self, **kw):
        self.engine.set_hparams(**kw)

    # -- inference (rollout) ------------------------------------------------------------------------
    def _pick(self, inputs: dict) -> Dict[str, torch.Tensor]:
        return {k: inputs[k].to(self.device, torch.float32).contiguous()
                for k in ('state_image', 'state_road', 'state_vehicle', 'state_navigation')}

    def predict(self, inputs: dict):
        """-> (action sample, mean, std, log_prob of the clipped sample, value (base, exp)); uses old_policy and BatchNorm
        moving statistics, like the reference's rollout forward (core/networks.py:181-193).  The leading axis of the inputs
        is the number of environments E stepped together (1 in the reference's loop); the Beta sample and its log-density
        come from one cdrl_beta_sample_logp launch on the (alpha, beta) the forward left on the device -- no host round trip.
        Returned tensors are fresh (not views of the engine's persistent output buffers)."""
        st = self._pick(inputs)
        E = st['state_image'].shape[0]
        out = self.rollout_for(E).predict(st)
        A = out['alpha'].shape[1]
        action = torch.empty((E, A), dtype=torch.float32, device=self.device)
        log_prob = torch.empty((E, A), dtype=torch.float32, device=self.device)
        self.action_index += 1
        alpha, beta = out['alpha'], out['beta']              # views of the persistent (E, 4, A) block: row stride 4A
        _lib.check(self.engine.lib.cdrl_beta_sample_logp(_lib.ptr(alpha), _lib.ptr(beta), E, A, 4 * A, int(self.sample_seed),
                                                         int(self.action_index * self.sample_stride + self.sample_rank), _lib.ptr(action), _lib.ptr(log_prob),
                                                         self.engine._stream()), 'cdrl_beta_sample_logp')
        return action, out['mean'].clone(), out['std'].clone(), log_prob, out['value'].clone()

    def dynamics_predict(self, inputs: dict):
        st = self._pick(inputs)
        return self.rollout_for(st['state_image'].shape[0]).predict(st)['dynamics'].clone()

    def dynamics_predict_train(self, inputs: dict):
        return self.engine.trunk_forward_train(self._pick(inputs))

    def data_for_dynamics(self, inputs):
        return inputs        # the 'action' input is a pass-through that no layer consumes

    def predict_last_value(self, state, is_terminal: bool, **kwargs):
        if is_terminal:
            return self.last_value
        st = self._pick(state)
        return self.rollout_for(st['state_image'].shape[0]).predict(st)['value'].clone()

    def value_predict(self, inputs):
        st = self._pick(inputs)
        return self.rollout_for(st['state_image'].shape[0]).predict(st)['value'].clone()

    # -- weights ----------------------------------------------------------------------------------
    def reset(self):
        super().reset()

    def update_old_policy(self, weights=None):
        if weights:
            self.engine.load_params('old_policy', weights)
        else:
            self.engine.update_old_policy()

    def get_weights(self) -> dict:
        return {m: self.engine.export_params(m) for m in ('policy', 'value', 'trunk')}

    def set_weights(self, weights: dict):
        for m, w in weights.items():
            self.engine.load_params(m, w)

    def trainable_variables(self):
        return {m: {k: v for k, v in self.engine.param_views(m).items() if self.engine.tables[m].by_name[k]['trainable']}
                for m in ('policy', 'value', 'trunk')}

    def _paths(self):
        return dict(policy=self.agent.weights_path['policy'] + '.npz', value=self.agent.weights_path['value'] + '.npz',
                    trunk=self.agent.dynamics_path + '.npz')

    def save_weights(self):
        """policy_net / value_net / dynamics_model as TensorFlow checkpoint-V2 files (`<name>.index` + two data shards), the
        layout Keras' `save_weights` leaves in the reference (core/networks.py:297-300), keyed `layer_with_weights-N/<var>` in
        Keras' layer order; optimizer state is not saved, as in the reference."""
        from .. import tf_checkpoint
        for model, path in self._paths().items():
            tf_checkpoint.save_from_engine(self.engine, model, path[:-4])

    def load_weights(self, full=True):
        """Loads TensorFlow checkpoint-V2 files (`<name>.index`: this package's own or the reference's shipped ones), or the
        `.npz` containers earlier versions of this package wrote (tf_checkpoint.py; tensors whose data shard is absent are skipped
        with a warning, e.g. the trunk shards the reference repository does not ship)."""
        from .. import tf_checkpoint
        paths = self._paths()
        for model in (('policy', 'value', 'trunk') if full else ('trunk',)):
            prefix = paths[model][:-4]
            if os.path.exists(prefix + '.index'):
                loaded = tf_checkpoint.load_into_engine(self.engine, model, prefix, strict=False)
                n = len(self.engine.tables[model].entries)
                if len(loaded) < n:
                    print(f'[load_weights] {prefix}: {len(loaded)}/{n} tensors present in the TF checkpoint')
                continue
            with np.load(paths[model]) as f:
                self.engine.load_params(model, {k: f[k] for k in f.files})
        if full:
            self.engine.update_old_policy()

    def summary(self):
        for model, title in (('policy', 'Policy Network'), ('value', 'Value Network'), ('trunk', 'Dynamics Model')):
            table = self.engine.tables[model]
            total = sum(e['numel'] for e in table.entries)
            train = sum(e['numel'] for e in table.entries if e['trainable'])
            print(f'==== {title} ====')
            for e in table.entries:
                print(f"  {e['name']:<34} {str(e['shape']):<20} {e['numel']:>9}{'' if e['trainable'] else '  (non-trainable)'}")
            print(f'  Total params: {total:,}  trainable: {train:,}  non-trainable: {total - train:,}\n')
