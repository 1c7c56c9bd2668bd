"""Minimal observation/action space descriptions (the reference uses gym.spaces; gym is not a
dependency of the learner path).  Only what Agent / PPOAgent read: shape, low, high, is_bounded,
Dict.spaces (reference rl/utils.py:212-247, rl/agents/ppo.py:148-181)."""
import numpy as np


class Space:
    shape = None


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            low = np.asarray(low, dtype=dtype)
            high = np.asarray(high, dtype=dtype)
            shape = low.shape
        else:
            low = np.full(shape, low, dtype=dtype)
            high = np.full(shape, high, dtype=dtype)
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    def is_bounded(self):
        return bool(np.all(np.isfinite(self.low)) and np.all(np.isfinite(self.high)))

    def sample(self, rng=None):
        rng = rng or np.random.default_rng()
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return rng.uniform(lo, hi).astype(self.dtype)

    def __repr__(self):
        return f'Box{self.shape}'


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()


class Dict(Space):
    def __init__(self, spaces=None, **kwargs):
        self.spaces = dict(spaces or {})
        self.spaces.update(kwargs)

    def __getitem__(self, k):
        return self.spaces[k]

    def items(self):
        return self.spaces.items()


class Env:
    """gym.Env-shaped base: step / reset / render / close / seed."""
    observation_space = None
    action_space = None

    def step(self, action):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def render(self, mode='human'):
        pass

    def close(self):
        pass

    def seed(self, seed=None):
        pass
