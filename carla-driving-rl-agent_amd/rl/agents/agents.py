"""`Agent` base class: constructor keywords, bookkeeping paths, config json, logging hooks
(same surface as reference rl/agents/agents.py:15-216; no TensorFlow / gym)."""
import json
import os
import random
from typing import List

import numpy as np
import torch

from .. import utils


class Agent:
    def __init__(self, env, batch_size: int, seed=None, weights_dir='weights', name='agent', log_mode='summary',
                 drop_batch_remainder=False, skip_data=0, consider_obs_every=1, evaluation_dir='evaluation',
                 shuffle_batches=False, shuffle=True, traces_dir: str = None, summary_keys: List[str] = None,
                 device='cuda:0'):
        if isinstance(env, str):
            raise ValueError('gym.make(env_id) is not supported: pass an environment object')
        self.env = env
        self.device = device
        self.seed = None
        self.set_random_seed(seed)
        self.batch_size = batch_size
        self.state_spec = utils.space_to_flat_spec(space=self.env.observation_space, name='state')
        self.action_spec = utils.space_to_flat_spec(space=self.env.action_space, name='action')
        if isinstance(traces_dir, str):
            self.should_record = True
            self.traces_dir = utils.makedir(traces_dir, name)
        else:
            self.should_record = False
        self.drop_batch_remainder = drop_batch_remainder
        self.skip_count = skip_data
        self.obs_skipping = consider_obs_every
        self.shuffle_batches = shuffle_batches
        self.shuffle = shuffle
        self.base_path = os.path.join(weights_dir, name)
        self.evaluation_path = os.path.join(evaluation_dir, name)
        self.weights_path = dict(policy=os.path.join(self.base_path, 'policy_net'),
                                 value=os.path.join(self.base_path, 'value_net'))
        self.config_path = os.path.join(self.base_path, 'config.json')
        self.config = dict()
        self.statistics = utils.Summary(mode=log_mode, name=name, keys=summary_keys)

    def set_random_seed(self, seed):
        if seed is not None:
            assert 0 <= seed < 2 ** 32
            torch.manual_seed(seed)
            np.random.seed(seed)
            random.seed(seed)
            self.env.seed(seed)
            self.seed = seed
        self.rng = np.random.default_rng(seed)

    # -- interface ------------------------------------------------------------------------------
    def act(self, state, *args, **kwargs):
        raise NotImplementedError

    def predict(self, state, *args, **kwargs):
        raise NotImplementedError

    def record(self, *args, **kwargs):
        pass

    def update(self):
        raise NotImplementedError

    def learn(self, *args, **kwargs):
        raise NotImplementedError

    def get_memory(self, *args, **kwargs):
        raise NotImplementedError

    def preprocess(self):
        return lambda state: state

    def summary(self):
        raise NotImplementedError

    # -- logging / config ---------------------------------------------------------------------------
    def log(self, **kwargs):
        self.statistics.log(**kwargs)

    def write_summaries(self):
        try:
            self.statistics.write_summaries()
        except Exception as e:           # logging must never stop training (reference :166-170)
            print(f'[write_summaries] error: {e}')

    def update_config(self, **kwargs):
        self.config.update(kwargs)

    def load_config(self):
        with open(self.config_path, 'r') as f:
            self.config = json.load(f)

    def save_config(self):
        os.makedirs(self.base_path, exist_ok=True)
        with open(self.config_path, 'w') as f:
            json.dump(self.config, fp=f)

    def reset(self):
        pass

    def load(self):
        self.load_weights()
        self.load_config()

    def save(self):
        self.save_weights()
        self.save_config()

    def load_weights(self):
        raise NotImplementedError

    def save_weights(self):
        raise NotImplementedError

    def on_episode_start(self):
        pass

    def on_episode_end(self):
        pass
