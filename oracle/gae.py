"""numpy restatement of returns / GAE / sp-norm / decompose (TEST INFRASTRUCTURE ONLY).

Follows rl/utils.py:57-84,140-151,344-349 and rl/agents/ppo.py:692-727.
PINNED: `discount_cumsum` is checked against scipy.signal.lfilter itself (the very
routine the reference calls, rl/utils.py:59) in tests/test_oracle_gae.py.
float32 element-wise ops mirror TF's float32 eager ops; the recurrence runs in
float64 because lfilter promotes float32 input to float64.
"""
import numpy as np

f32 = np.float32


def discount_cumsum(x: np.ndarray, discount: float) -> np.ndarray:
    """y[n] = x[n] + discount * y[n+1] in float64 (direct-form-II-transposed lfilter with
    b=[1], a=[1,-discount] on the reversed sequence; rl/utils.py:57-59)."""
    x64 = np.asarray(x, dtype=np.float64)
    y = np.empty_like(x64)
    acc = 0.0
    d = float(discount)
    for i in range(len(x64) - 1, -1, -1):
        acc = x64[i] + d * acc
        y[i] = acc
    return y


def decompose_number(num: f32):
    """rl/utils.py:140-151 on a float32 scalar: while |x| > 1: x /= 10 (float32)."""
    x = f32(num)
    e = 0
    ten = f32(10.0)
    while abs(x) > f32(1.0):
        x = f32(x / ten)
        e += 1
    return x, f32(e)


def compute_returns(rewards: np.ndarray, gamma: float):
    """PPOMemory.compute_returns (rl/agents/ppo.py:699-712). `rewards` already has the
    bootstrap value appended (end_trajectory :692-697).  Returns (returns f32 (N,),
    decomposed (N,2) f32)."""
    ret = discount_cumsum(rewards, gamma)[:-1].astype(f32)          # rewards_to_go + to_float
    dec = np.array([decompose_number(r) for r in ret], dtype=f32).reshape(-1, 2)
    return ret, dec


def sp_norm(x: np.ndarray, eps=1e-3) -> np.ndarray:
    """tf_sp_norm (rl/utils.py:344-349), float32."""
    x = x.astype(f32)
    pos = x * (x > 0).astype(f32)
    neg = x * (x < 0).astype(f32)
    return (pos / f32(x.max() + f32(eps))) + (neg / f32(-(x.min() - f32(eps))))


def compute_advantages(rewards: np.ndarray, values_be: np.ndarray, gamma: float, lambda_: float, scale=2.0):
    """PPOMemory.compute_advantages (rl/agents/ppo.py:714-727) + utils.gae (rl/utils.py:62-72).
    rewards (N+1,) f32 incl. bootstrap; values_be (N+1,2) f32 (base, exp).
    Returns (values f32 (N+1,), raw advantages f32 (N,), normalised*scale f32 (N,))."""
    rewards = rewards.astype(f32)
    values = (values_be[:, 0].astype(f32) * np.power(f32(10.0), values_be[:, 1].astype(f32))).astype(f32)
    g = f32(gamma)
    deltas = (rewards[:-1] + g * values[1:] - values[:-1]).astype(f32)
    if lambda_ == 0.0:
        adv = deltas
    else:
        adv = discount_cumsum(deltas, gamma * lambda_).astype(f32)
    return values, adv, (sp_norm(adv) * f32(scale)).astype(f32)


def end_trajectory(rewards: np.ndarray, values_be: np.ndarray, last_value: np.ndarray):
    """PPOMemory.end_trajectory (rl/agents/ppo.py:692-697): append bootstrap reward and value."""
    last_value = np.asarray(last_value, dtype=f32).reshape(1, 2)
    boot = (last_value[:, 0] * np.power(f32(10.0), last_value[:, 1])).astype(f32)
    return np.concatenate([rewards.astype(f32), boot]), np.concatenate([values_be.astype(f32), last_value])
