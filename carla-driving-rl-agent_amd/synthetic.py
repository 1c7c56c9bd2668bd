"""Deterministic synthetic rollout buffers.

Shapes and value ranges follow the real CARLAEnv observation contract
(reference core/carla_env.py:335-382 feature builders, :182-202 reward) so that the
learner hot path sees realistic magnitudes.  Generator = numpy default_rng(seed).
"""
import math

import numpy as np

f32 = np.float32


def _beta_logpdf(x, a, b):
    lg = math.lgamma
    return ((a - 1.0) * np.log(x) + (b - 1.0) * np.log1p(-x) - (lg(a) + lg(b) - lg(a + b))).astype(f32)


def make_rollout(n: int, T=4, H=90, W=120, road=9, vehicle=4, navigation=5, A=2, seed=42, terminal_spike=False):
    """A rollout of `n` timesteps (one env shard / episode)."""
    rng = np.random.default_rng(seed)
    image = rng.random((n, T, H, W, 3), dtype=f32)                                  # Box(0,1)
    rd = np.zeros((n, T, road), dtype=f32)
    rd[..., :3] = rng.integers(0, 2, size=(n, T, 3))
    if road > 3:
        rd[..., 3] = rng.uniform(0.0, 0.9, size=(n, T))
    if road > 4:
        rd[..., 4:] = rng.integers(0, 2, size=(n, T, road - 4))
    vh = np.zeros((n, T, vehicle), dtype=f32)
    vh[..., 0] = rng.uniform(-1.0, 1.0, size=(n, T))
    if vehicle > 1:
        vh[..., 1] = rng.uniform(0.0, 0.3, size=(n, T))
    if vehicle > 2:
        vh[..., 2:] = rng.uniform(0.0, 1.0, size=(n, T, vehicle - 2))
    nav = np.sort(rng.uniform(0.0, 25.0, size=(n, T, navigation)), axis=-1).astype(f32)
    action = rng.beta(2.0, 2.0, size=(n, A)).astype(f32)
    action = np.clip(action, 1e-4, 1.0 - 1e-4)
    old_log_prob = _beta_logpdf(action.astype(np.float64), 2.0, 2.0)
    value = np.stack([rng.uniform(-1.0, 1.0, size=n), rng.uniform(0.0, 6.0, size=n)], axis=1).astype(f32)
    speed = rng.uniform(0.0, 30.0, size=(n, 1)).astype(f32)
    similarity = rng.uniform(-1.0, 1.0, size=(n, 1)).astype(f32)
    reward = (speed[:, 0] / 3.0 * np.abs(similarity[:, 0])).astype(f32)             # in [0, 10]
    if terminal_spike:
        reward[-1] = f32(-1000.0)
    return dict(states=dict(state_image=image, state_road=rd, state_vehicle=vh, state_navigation=nav),
                action=action, old_log_prob=old_log_prob, value=value, reward=reward,
                speed=speed, similarity=similarity)


def beta_sample_with_jacobian(alpha, beta, rng):
    """Host-side Beta sample u ~ Beta(alpha, beta) and a finite-difference-free pathwise
    Jacobian (du/dalpha, du/dbeta) via the implicit function theorem on the CDF:
    du/dtheta = -dF/dtheta / pdf(u).  Used only by tests / harness to produce explicit
    inputs for the faithful (re-sampling) policy loss (F8)."""
    from scipy import special, stats
    a = np.asarray(alpha, dtype=np.float64)
    b = np.asarray(beta, dtype=np.float64)
    u = rng.beta(a, b)
    u = np.clip(u, 1e-6, 1 - 1e-6)
    pdf = stats.beta.pdf(u, a, b)
    h = 1e-5
    dFa = (special.betainc(a + h, b, u) - special.betainc(a - h, b, u)) / (2 * h)
    dFb = (special.betainc(a, b + h, u) - special.betainc(a, b - h, u)) / (2 * h)
    return u.astype(f32), (-dFa / pdf).astype(f32), (-dFb / pdf).astype(f32)


DEFAULT_HP = dict(gamma=0.9999, lambda_=0.999, clip_ratio=0.2, entropy_coef=1.0, policy_lr=3e-4, value_lr=3e-4,
                  dynamics_lr=3e-4, clip_norm_policy=1.0, clip_norm_value=1.0, advantage_scale=2.0)
