"""PPOAgent / PPOMemory: host-side control flow of the learner (what to run, in which order);
all arithmetic runs in libcdrl_hip.so through the network's LearnerEngine.

Surface and step order follow the reference rl/agents/ppo.py: constructor keywords (:26-32),
update() = all policy minibatches then all value minibatches (:190-226), learn() rollout loop
(:464-568), end_episode (:574-585), PPOMemory (:629-733)."""
import os
import time
from typing import Optional, Sequence, Union

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
        # (E, A) device sample in [0, 1] -> environment actions; one row (the reference's single environment) comes back as (A,)
        self.convert_action = lambda a: np.squeeze(a.detach().cpu().numpy() * self.action_range + self.action_low, axis=0) \
            if a.shape[0] == 1 else a.detach().cpu().numpy() * self.action_range + self.action_low

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
    @staticmethod
    def _period(every, episodes: int, end_keyword=False) -> int:
        """`save_every` / `render_every` of the reference's learn() as a period in episodes: False / None -> never,
        True -> every episode, 'end' (saving only) -> the last episode, an int -> that period (must divide `episodes`)."""
        if every is False or every is None:
            return episodes + 1
        if every is True:
            return 1
        if end_keyword and every == 'end':
            return episodes
        assert not isinstance(every, str) and episodes % every == 0
        return int(every)

    def learn(self, episodes: int, timesteps: int, save_every: Union[bool, str, int] = False,
              render_every: Union[bool, str, int] = False, close=True, envs: Optional[Sequence] = None):
        """The reference's training loop (rl/agents/ppo.py:464-568: roll out up to `timesteps` steps per episode, close the
        trajectory with the bootstrap value, update() every `update_frequency` episodes, then the host side effects) built around
        an ENVIRONMENT SHARD: `envs` (default `[self.env]`, the reference's case) are stepped in lockstep, one batched
        `predict` per step for all of them, their rows stay on the device, and at the end of the episode every environment's
        trajectory is appended to the memory with its own bootstrap value and its own returns / GAE(lambda).  With one environment
        the sequence of environment, sampler and memory operations is the reference's."""
        assert episodes % self.update_frequency == 0
        save_period = self._period(save_every, episodes, end_keyword=True)
        render_period = self._period(render_every, episodes)
        shard = list(envs) if envs is not None else [self.env]
        try:
            self.memory = self.get_memory()
            for episode in range(1, episodes + 1):
                self.seed_regularization()
                self.on_episode_start()
                self.reset()
                rollout = self.collect(shard, timesteps, render=episode % render_period == 0, episode=episode)
                updating = episode % self.update_frequency == 0
                self.store(rollout, timesteps, keep_open=not updating)
                if updating:
                    self.update()
                    self.memory.delete()
                    self.memory = self.get_memory()
                self.log(episode_rewards=rollout.episode_reward if len(shard) > 1 else rollout.episode_reward[0])
                # host side effects: ONE writer under data parallelism (N ranks writing the same checkpoint / summary files
                # concurrently leave torn files), everyone waits for it
                if self.is_writer():
                    self.write_summaries()
                    if self.should_record:
                        self.record(episode)
                else:
                    self.statistics.stats = {}
                self.on_episode_end()
                if episode % save_period == 0:
                    # the writer's failure (disk full, permissions) must not leave the other ranks waiting in a bare barrier until the
                    # process-group timeout: the meeting point carries a status flag and every rank raises
                    failure = None
                    if self.is_writer():
                        try:
                            self.save()
                        except Exception as exc:       # noqa: BLE001 -- re-raised below, on every rank
                            failure = exc
                    self.rank_barrier(failed=failure is not None)
                    if failure is not None:
                        raise failure
        finally:
            if close:
                for env in shard:
                    env.close()

    def is_writer(self) -> bool:
        """Hook for data-parallel agents: the rank that writes checkpoints, summaries and traces."""
        return True

    def rank_barrier(self, failed: bool = False):
        """Hook for data-parallel agents: all ranks meet here after the writer has written; `failed` = this rank's write raised --
        the hook must make EVERY rank raise then."""

    def observe(self, observations: list, preprocess_fn) -> dict:
        """Per-environment observations -> one dict of (E, ...) device tensors: keys get the reference's `state_` prefix, the
        agent's preprocess function runs per environment (it may return device tensors: the augmentation kernels), host arrays
        cross PCIe as ONE copy per key."""
        prepared = []
        for obs in observations:
            if isinstance(obs, dict):
                obs = {f'state_{k}': v for k, v in obs.items()}
            prepared.append(preprocess_fn(obs))
        if not isinstance(prepared[0], dict):
            prepared = [dict(state=p) for p in prepared]
        out = {}
        for key in prepared[0]:
            vals = [p[key] for p in prepared]
            if any(isinstance(v, torch.Tensor) for v in vals):
                vals = [v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v, dtype=np.float32)) for v in vals]
                out[key] = torch.stack([v.to(device=self.device, dtype=torch.float32) for v in vals], dim=0)
            else:
                out[key] = torch.as_tensor(np.stack([np.asarray(v, dtype=np.float32) for v in vals], axis=0)).to(self.device)
        return out

    def collect(self, shard: list, timesteps: int, render=False, episode=0) -> 'Rollout':
        """One episode of every environment of the shard, in lockstep.  An environment that terminates early keeps its last
        observation in the batch (its rows are not recorded any more), so the inference engine sees one shape throughout."""
        E = len(shard)
        preprocess_fn = self.preprocess()
        observations = [env.reset() for env in shard]
        rollout = Rollout(E, timesteps, self.device)
        running = list(range(E))
        t0 = time.time()
        batch = self.observe(observations, preprocess_fn)
        for t in range(1, timesteps + 1):
            if render:
                for e in running:
                    shard[e].render()
            action, mean, std, log_prob, value = self.predict(batch)
            rollout.record(batch, action, log_prob, value, running)
            env_actions = np.atleast_2d(self.convert_action(action))
            rewards = []
            for e in list(running):
                reward, done = 0.0, False
                for _ in range(self.repeat_action):
                    observations[e], reward, done, _ = shard[e].step(env_actions[e])
                    rollout.episode_reward[e] += reward
                    rollout.env_steps[e] += 1
                    if done:
                        break
                rollout.rewards[e].append(float(reward))
                rewards.append(reward)
                if done or t == timesteps:
                    rollout.terminal[e] = done
                    running.remove(e)
                    print(f'Episode {episode}{f" [env {e}]" if E > 1 else ""} terminated after {t} timesteps in '
                          f'{round(time.time() - t0, 3)}s with reward {round(rollout.episode_reward[e], 3)}.')
                    self.log(timestep=t)
            self.log(actions=action, rewards=rewards if E > 1 else rewards[0], distribution_mean=mean, distribution_std=std)
            batch = self.observe(observations, preprocess_fn)
            if not running:
                break
        rollout.final = batch               # the observation behind every environment's last recorded step
        return rollout

    def store(self, rollout: 'Rollout', timesteps: int, keep_open: bool):
        """Appends every environment's trajectory to the memory: rows, bootstrap value (zero at a terminal state, the value
        head's estimate of the final observation otherwise), returns and GAE(lambda) of THAT trajectory (end_episode).
        `keep_open`: more rows follow before the next update() (update_frequency > 1) -- the bootstrap entry is removed again."""
        E = rollout.envs
        estimate = None
        if not all(rollout.terminal):
            t_last = max(rollout.length)
            estimate = self.network.predict_last_value(rollout.final, timestep=(t_last + 1) / timesteps, is_terminal=False)
        # `append` (reference: update_frequency > 1) = the memory's row index advances PAST rows only, because the bootstrap entry of
        # this trajectory is removed again before the next one arrives -- also the case between the trajectories of a shard
        append = self.update_frequency > 1 or E > 1
        for e in range(E):
            self.memory.extend(*rollout.trajectory(e))
            last_value = (self.network.predict_last_value(None, is_terminal=True) if rollout.terminal[e] else estimate[e:e + 1])
            self.end_episode(last_value, append=append)
            self.trajectory_stored(e, rollout)
            if keep_open or e < E - 1:
                self.memory.drop_bootstrap()
        rollout.blocks = None               # the trajectories were copied into the memory: release the timesteps-long staging blocks

    def trajectory_stored(self, env_index: int, rollout: 'Rollout'):
        """Hook: environment `env_index`'s trajectory of this rollout now sits at the end of the memory."""

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


class Rollout:
    """What an environment shard stepped in lockstep leaves behind: (steps, E, ...) device blocks written in place, one per
    state component / action / log-probability / value, plus the host-side bookkeeping per environment (rewards, number of
    recorded steps, terminal flag).  `trajectory(e)` hands environment e's rows to the memory as contiguous blocks."""

    def __init__(self, envs: int, timesteps: int, device):
        self.envs, self.timesteps, self.device = envs, timesteps, device
        self.blocks = None                                  # name -> (timesteps, E, ...) tensor, allocated by the first record()
        self.rewards = [[] for _ in range(envs)]
        self.length = [0] * envs
        self.env_steps = [0] * envs                         # environment steps incl. repeated actions (info-buffer entries)
        self.terminal = [False] * envs
        self.episode_reward = [0.0] * envs
        self.final = None
        self.step = 0

    def record(self, states: dict, action, log_prob, value, running):
        rows = dict(states)
        rows.update({'/action': action, '/log_prob': log_prob, '/value': value})
        if self.blocks is None:
            self.blocks = {k: torch.empty((self.timesteps,) + tuple(v.shape), dtype=torch.float32, device=self.device)
                           for k, v in rows.items()}
        for k, v in rows.items():
            self.blocks[k][self.step].copy_(v)
        self.step += 1
        for e in running:
            self.length[e] += 1

    def trajectory(self, e: int):
        """-> (states, actions, rewards, values, log_probs) of environment e: its first `length[e]` steps."""
        n = self.length[e]
        # a COPY of the environment's rows: for one environment `[:n, 0].contiguous()` would be a view that keeps the whole
        # timesteps-long block alive inside the memory after an early episode end
        take = lambda k: self.blocks[k][:n, e].clone(memory_format=torch.contiguous_format)
        states = {k: take(k) for k in self.blocks if not k.startswith('/')}
        return states, take('/action'), self.rewards[e][:n], take('/value'), take('/log_prob')


class PPOMemory:
    """Rollout buffer resident in HBM.  Rows arrive one at a time (`append`, (1, ...) device tensors) or as whole
    trajectories (`extend`, (n, ...) blocks) and are concatenated on demand; returns / advantages come from the
    cdrl_gae_returns kernel."""

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
        self._n = 0

    def __len__(self):
        return self._n

    def delete(self):
        self._states = None
        self._rewards = self._values = self._actions = self._log_probs = None
        self.returns = self.advantages = None
        self._cache = {}
        self._n = 0

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
        self._n += 1

    def extend(self, states, actions, rewards, values, log_probs):
        """A whole trajectory of n rows: (n, ...) device blocks (state dict or tensor, actions (n, A), values (n, 2),
        log-probabilities (n, A)) and n host rewards."""
        n = int(actions.shape[0])
        if n == 0:
            return
        self._cache = {}
        if self.simple_state:
            self._states.append(states['state'] if isinstance(states, dict) else states)
        else:
            for k, v in states.items():
                if k in self._states:
                    self._states[k].append(v)
        self._actions.append(actions.reshape(n, -1))
        self._rewards.extend(float(r) for r in rewards)
        self._values.append(values.reshape(n, -1))
        self._log_probs.append(log_probs.reshape(n, -1))
        self._n += n

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
