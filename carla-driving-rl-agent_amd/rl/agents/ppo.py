"""PPOAgent / PPOMemory: host-side control flow of the learner (what to run, in which order);
all arithmetic runs in libcdrl_hip.so through the network's LearnerEngine.

Surface and step order follow the reference rl/agents/ppo.py: constructor keywords (:26-32),
update() = all policy minibatches then all value minibatches (:190-226), learn() rollout loop
(:464-568), end_episode (:574-585), PPOMemory (:629-733)."""
import os
import time
from typing import Union

import numpy as np
import torch

from .. import utils
from ..parameters import DynamicParameter
from .agents import Agent


class PPOAgent(Agent):
    def __init__(self, *args, policy_lr=1e-3, gamma=0.99, lambda_=0.95, value_lr=3e-4, load=False,
                 optimization_steps=(1, 1), name='ppo-agent', optimizer='adam', clip_norm=(1.0, 1.0), clip_ratio=0.2,
                 seed_regularization=False, entropy_regularization=0.0, network: dict = None, update_frequency=1,
                 polyak=1.0, repeat_action=1, advantage_scale=2.0, **kwargs):
        assert 0.0 < polyak <= 1.0
        assert repeat_action >= 1
        if str(optimizer).lower() != 'adam':
            raise ValueError("only optimizer='adam' is implemented natively (the reference stages all use it)")
        if polyak < 1.0:
            raise NotImplementedError('polyak averaging < 1.0 is off in every reference stage and not implemented')
        super().__init__(*args, name=name, **kwargs)
        self.memory: PPOMemory = None
        self.gamma = gamma
        self.lambda_ = lambda_
        self.repeat_action = repeat_action
        self.adv_scale = DynamicParameter.create(value=advantage_scale)
        if seed_regularization:
            def _seed_regularization():
                self.set_random_seed(int(np.random.randint(0, 2 ** 31 - 1)))
            self.seed_regularization = _seed_regularization
            self.seed_regularization()
        else:
            self.seed_regularization = lambda: None
        self.entropy_strength = DynamicParameter.create(value=entropy_regularization)
        self.clip_ratio = DynamicParameter.create(value=clip_ratio)
        self._init_action_space()
        self._init_gradient_clipping(clip_norm)
        self.update_frequency = update_frequency
        self.policy_lr = DynamicParameter.create(value=policy_lr)
        self.value_lr = DynamicParameter.create(value=value_lr)
        self.optimization_steps = dict(policy=optimization_steps[0], value=optimization_steps[1])
        self.should_polyak_average = False
        self.polyak_coeff = polyak
        if not isinstance(network, dict) or 'network' not in network:
            raise ValueError("PPOAgent needs network=dict(network=<Network class>, ...)")
        network = dict(network)
        network_class = network.pop('network')
        self.network = network_class(agent=self, **network)
        if load:
            self.load()

    # -- setup ------------------------------------------------------------------------------------
    def _init_gradient_clipping(self, clip_norm):
        def one(c):
            return (False, None) if c is None else (True, float(c))
        if clip_norm is None:
            clip_norm = (None, None)
        elif isinstance(clip_norm, float):
            assert clip_norm > 0.0
            clip_norm = (clip_norm, clip_norm)
        self.should_clip_policy_grads, self.grad_norm_policy = one(clip_norm[0])
        self.should_clip_value_grads, self.grad_norm_value = one(clip_norm[1])

    def _init_action_space(self):
        space = self.env.action_space
        if not (isinstance(space, utils.spaces.Box) and space.is_bounded()):
            raise NotImplementedError('only bounded Box action spaces (Beta policy) are on the native path')
        self.num_actions = space.shape[0]
        self.distribution_type = 'beta'
        self.action_low = np.asarray(space.low, dtype=np.float32)
        self.action_high = np.asarray(space.high, dtype=np.float32)
        self.action_range = self.action_high - self.action_low
        self.convert_action = lambda a: (a[0].detach().cpu().numpy() * self.action_range + self.action_low)

    # -- acting -----------------------------------------------------------------------------------
    def predict(self, state, *args, **kwargs):
        return self.network.predict(inputs=state)

    def act(self, state, *args, **kwargs):
        return self.convert_action(self.network.predict(inputs=state)[0])

    # -- update -----------------------------------------------------------------------------------
    def hyper_parameters(self) -> dict:
        return dict(policy_lr=self.policy_lr(), value_lr=self.value_lr(), clip_ratio=self.clip_ratio(),
                    entropy_coef=self.entropy_strength(),
                    clip_norm_policy=self.grad_norm_policy if self.should_clip_policy_grads else 0.0,
                    clip_norm_value=self.grad_norm_value if self.should_clip_value_grads else 0.0)

    def update(self):
        t0 = time.time()
        self.seed_regularization()
        self.network.set_hparams(**self.hyper_parameters())
        value_batches = list(self.get_value_batches())
        policy_batches = list(self.get_policy_batches())
        policy_batches, value_batches = self.agree_on_batches(policy_batches, value_batches)
        for _ in range(self.optimization_steps['policy']):
            for batch in policy_batches:
                self.seed_regularization()
                total_loss, grads = self.get_policy_gradients(batch)
                self.update_policy(grads)
                self.log(loss_total=total_loss, lr_policy=self.policy_lr.value)
        for _ in range(self.optimization_steps['value']):
            for batch in value_batches:
                self.seed_regularization()
                value_loss, grads = self.get_value_gradients(batch)
                self.update_value(grads)
                self.log(loss_value=value_loss, lr_value=self.value_lr.value)
        self.after_update()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        print(f'Update took {round(time.time() - t0, 3)}s')

    def agree_on_batches(self, policy_batches, value_batches):
        """Hook for data-parallel agents: every rank must run the same number of minibatch steps (each one is a collective)."""
        return policy_batches, value_batches

    def after_update(self):
        """Hook for data-parallel agents (BatchNorm moving statistics are averaged once per update())."""

    def update_policy(self, gradients):
        return self.apply_policy_gradients(gradients), True

    def update_value(self, gradients):
        return self.apply_value_gradients(gradients), True

    def get_policy_gradients(self, batch):
        raise NotImplementedError

    def get_value_gradients(self, batch):
        raise NotImplementedError

    def apply_policy_gradients(self, gradients):
        raise NotImplementedError

    def apply_value_gradients(self, gradients):
        raise NotImplementedError

    def value_batch_tensors(self):
        return self.memory.states, self.memory.returns

    def policy_batch_tensors(self):
        return self.memory.states, self.memory.advantages, self.memory.actions, self.memory.log_probabilities

    def _batches(self, tensors, shuffle, shuffle_batches):
        return utils.data_to_batches(tensors=tensors, batch_size=self.batch_size, drop_remainder=self.drop_batch_remainder,
                                     skip=self.skip_count, num_shards=self.obs_skipping, shuffle=shuffle,
                                     shuffle_batches=shuffle_batches, seed=int(self.rng.integers(2 ** 31)))

    def get_value_batches(self):
        return self._batches(self.value_batch_tensors(), True, False)

    def get_policy_batches(self):
        return self._batches(self.policy_batch_tensors(), self.shuffle, self.shuffle_batches)

    # -- learning loop ------------------------------------------------------------------------------
    def learn(self, episodes: int, timesteps: int, save_every: Union[bool, str, int] = False,
              render_every: Union[bool, str, int] = False, close=True):
        assert episodes % self.update_frequency == 0
        if save_every in (False, None):
            save_every = episodes + 1
        elif save_every is True:
            save_every = 1
        elif save_every == 'end':
            save_every = episodes
        else:
            assert episodes % save_every == 0
        if render_every is False:
            render_every = episodes + 1
        elif render_every is True:
            render_every = 1
        try:
            self.memory = self.get_memory()
            for episode in range(1, episodes + 1):
                self.seed_regularization()
                self.on_episode_start()
                preprocess_fn = self.preprocess()
                self.reset()
                state = self.env.reset()
                episode_reward = 0.0
                t0 = time.time()
                render = episode % render_every == 0
                for t in range(1, timesteps + 1):
                    if render:
                        self.env.render()
                    if isinstance(state, dict):
                        state = {f'state_{k}': v for k, v in state.items()}
                    state = utils.to_tensor(preprocess_fn(state), device=self.device)
                    action, mean, std, log_prob, value = self.predict(state)
                    action_env = self.convert_action(action)
                    reward, done = 0.0, False
                    for _ in range(self.repeat_action):
                        next_state, reward, done, _ = self.env.step(action_env)
                        episode_reward += reward
                        if done:
                            break
                    self.log(actions=action, rewards=reward, distribution_mean=mean, distribution_std=std)
                    self.memory.append(state, action, reward, value, log_prob)
                    state = next_state
                    if done or (t == timesteps):
                        print(f'Episode {episode} terminated after {t} timesteps in {round(time.time() - t0, 3)}s '
                              f'with reward {round(episode_reward, 3)}.')
                        self.log(timestep=t)
                        if isinstance(state, dict):
                            state = {f'state_{k}': v for k, v in state.items()}
                        state = utils.to_tensor(preprocess_fn(state), device=self.device)
                        last_value = self.network.predict_last_value(state, timestep=(t + 1) / timesteps, is_terminal=done)
                        self.end_episode(last_value, append=self.update_frequency > 1)
                        break
                if episode % self.update_frequency == 0:
                    self.update()
                    self.memory.delete()
                    self.memory = self.get_memory()
                elif self.update_frequency > 1:
                    self.memory.drop_bootstrap()
                self.log(episode_rewards=episode_reward)
                self.write_summaries()
                if self.should_record:
                    self.record(episode)
                self.on_episode_end()
                if episode % save_every == 0:
                    self.save()
        finally:
            if close:
                self.env.close()

    def get_memory(self):
        return PPOMemory(state_spec=self.state_spec, num_actions=self.num_actions, device=self.device)

    def end_episode(self, last_value, append=False):
        self.memory.end_trajectory(last_value)
        returns = self.memory.compute_returns(discount=self.gamma, append=append)
        values, advantages = self.memory.compute_advantages(self.gamma, self.lambda_, scale=self.adv_scale(), append=append)
        self.memory.update_index(append=append)
        self.log(returns=returns, advantages=advantages, values=values, advantage_scale=self.adv_scale.value)

    def summary(self):
        self.network.summary()

    def save_weights(self):
        self.network.save_weights()

    def load_weights(self):
        self.network.load_weights()

    def save_config(self):
        self.update_config(policy_lr=self.policy_lr.serialize(), value_lr=self.value_lr.serialize(),
                           adv_scale=self.adv_scale.serialize(), entropy_strength=self.entropy_strength.serialize(),
                           clip_ratio=self.clip_ratio.serialize())
        super().save_config()

    def load_config(self):
        super().load_config()
        for key, p in (('policy_lr', self.policy_lr), ('value_lr', self.value_lr), ('adv_scale', self.adv_scale),
                       ('entropy_strength', self.entropy_strength), ('clip_ratio', self.clip_ratio)):
            p.load(config=self.config.get(key, {}))

    def reset(self):
        super().reset()
        self.network.reset()

    def on_episode_end(self):
        super().on_episode_end()
        self.policy_lr.on_episode()
        self.value_lr.on_episode()
        self.adv_scale.on_episode()


class PPOMemory:
    """Rollout buffer resident in HBM.  Rows are appended as (1, ...) device tensors and stacked on
    demand; returns / advantages come from the cdrl_gae_returns kernel."""

    def __init__(self, state_spec: dict, num_actions: int, device='cuda:0'):
        self.device = device
        self.index = 0
        self.simple_state = list(state_spec.keys()) == ['state']
        self.state_spec = state_spec
        self.num_actions = num_actions
        self._states = [] if self.simple_state else {name: [] for name in state_spec}
        self._rewards, self._values, self._actions, self._log_probs = [], [], [], []
        self.returns = None
        self.advantages = None
        self._cache = {}

    def __len__(self):
        return len(self._actions)

    def delete(self):
        self._states = None
        self._rewards = self._values = self._actions = self._log_probs = None
        self.returns = self.advantages = None
        self._cache = {}

    @staticmethod
    def _row(x, device, width=None):
        t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x, dtype=np.float32))
        t = t.to(device=device, dtype=torch.float32)
        return t.reshape(1, -1) if width else t

    def append(self, state, action, reward, value, log_prob):
        self._cache = {}
        if self.simple_state:
            self._states.append(self._row(state, self.device))
        else:
            assert isinstance(state, dict)
            for k, v in state.items():
                if k in self._states:
                    self._states[k].append(self._row(v, self.device))
        self._actions.append(self._row(action, self.device, True))
        self._rewards.append(float(reward))
        self._values.append(self._row(value, self.device, True))
        self._log_probs.append(self._row(log_prob, self.device, True))

    def _cat(self, key, rows):
        if key not in self._cache:
            self._cache[key] = torch.cat(rows, dim=0).contiguous()
        return self._cache[key]

    @property
    def states(self):
        if self.simple_state:
            return self._cat('states', self._states)
        return {k: self._cat(f'states/{k}', v) for k, v in self._states.items()}

    @property
    def actions(self):
        return self._cat('actions', self._actions) if self._actions else torch.zeros((0, self.num_actions), device=self.device)

    @property
    def log_probabilities(self):
        return self._cat('log_probs', self._log_probs)

    @property
    def values(self):
        return self._cat('values', self._values)

    @property
    def rewards(self):
        if 'rewards' not in self._cache:
            self._cache['rewards'] = torch.as_tensor(np.asarray(self._rewards, dtype=np.float32), device=self.device)
        return self._cache['rewards']

    def end_trajectory(self, last_value):
        """bootstrap: reward <- base * 10^exp of the last value, value <- last_value."""
        lv = self._row(last_value, self.device, True)
        self._cache = {}
        self._rewards.append(float((lv[0, 0] * torch.pow(torch.tensor(10.0, device=lv.device), lv[0, 1])).item()))
        self._values.append(lv)

    def drop_bootstrap(self):
        self._cache = {}
        self._rewards = self._rewards[:-1]
        self._values = self._values[:-1]

    def _kernel(self, gamma, lambda_, scale):
        return utils.returns_and_advantages(self.rewards[self.index:], self.values[self.index:], gamma, lambda_, scale,
                                            device=self.device)

    def compute_returns(self, discount: float, append=False):
        out = self._kernel(discount, 0.0, 1.0)
        new = out['returns_be']
        self.returns = new if (self.returns is None or not append) else torch.cat([self.returns, new], dim=0)
        return out['returns']

    def compute_advantages(self, gamma: float, lambda_: float, scale=2.0, append=False):
        out = self._kernel(gamma, lambda_, scale)
        new = out['advantages']
        self.advantages = new if (self.advantages is None or not append) else torch.cat([self.advantages, new], dim=0)
        v = self.values[self.index:]
        return v[:, 0] * torch.pow(torch.tensor(10.0, device=v.device), v[:, 1]), out['advantages_raw']

    def update_index(self, append=False):
        self.index = len(self._rewards) - 1 if append else len(self._rewards)

    def serialize(self, episode: int, save_path: str):
        path = os.path.join(save_path, f'trace-{episode}-{time.strftime("%Y%m%d-%H%M%S")}.npz')
        buf = dict(reward=self.rewards.cpu().numpy(), action=self.actions.cpu().numpy(), value=self.values.cpu().numpy(),
                   log_prob=self.log_probabilities.cpu().numpy())
        st = self.states
        if self.simple_state:
            buf['state'] = st.cpu().numpy()
        else:
            buf.update({k: v.cpu().numpy() for k, v in st.items()})
        np.savez_compressed(file=path, **buf)
