"""Device augmentation (cdrl_augment_images) against the numpy oracle, element by element: the plan fixes the scalar
decisions and both sides draw the per-pixel random fields from the same Philox streams."""
import numpy as np
import pytest
import torch

from oracle import augment as A

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _plan(**kw):
    from carla_driving_rl_agent_amd.rl.augmentations import empty_plan
    p = empty_plan(seed=0x1234567890abcdef, offset=17)
    p.update(kw)
    return p


def _kernel(k, seed=3):
    w = np.random.default_rng(seed).normal(1.0, 0.25, (k, k, 3)).astype(np.float32).reshape(-1)
    return list(w) + [0.0] * (75 - w.size)


CASES = [
    ('identity', {}),
    ('jitter', dict(jitter=1, brightness=0.13, contrast=1.4, saturation=0.6, hue=-0.11)),
    ('jitter2', dict(jitter=1, brightness=-0.2, contrast=0.3, saturation=1.7, hue=0.2)),
    ('blur3', dict(blur_size=3, blur_kernel=_kernel(3))),
    ('blur5', dict(blur_size=5, blur_kernel=_kernel(5), normalize=1)),
    ('salt_pepper', dict(salt_pepper=1, sp_amount=0.1, sp_prob=0.5)),
    ('gauss', dict(gauss_noise=1, gn_amount=0.1, gn_std=0.075)),
    ('normalize', dict(normalize=1)),
    ('cutout', dict(cutout_size=6, cutout_cell=21)),
    ('dropout', dict(dropout_size=81, dropout_amount=0.04)),
    ('all', dict(jitter=1, brightness=0.05, contrast=1.2, saturation=1.3, hue=0.07, blur_size=3, blur_kernel=_kernel(3, 9),
                 salt_pepper=1, gauss_noise=1, normalize=1, cutout_size=6, cutout_cell=3, dropout_size=81)),
]


@pytest.mark.parametrize('shape', [(4, 90, 120), (2, 23, 31)])
@pytest.mark.parametrize('name,kw', CASES, ids=[c[0] for c in CASES])
def test_augment_matches_oracle(lib, shape, name, kw):
    from carla_driving_rl_agent_amd.rl.augmentations import Augmenter
    T, H, W = shape
    x = np.random.default_rng(T * H + W).uniform(0.0, 1.0, (T, H, W, 3)).astype(np.float32)
    plan = _plan(**kw)
    got = Augmenter(DEV)(x, plan).cpu().numpy()
    ref = A.augment(x, plan)
    scale = max(1.0, float(np.abs(ref).max()))
    err = np.abs(got - ref) / scale
    # HSV hue is discontinuous where two channels tie for the maximum: allow a handful of pixels to take the other branch
    assert np.quantile(err, 0.9999) < 2e-5, (name, float(err.max()))
    assert (err > 1e-3).mean() < 1e-4, (name, float(err.max()))
    # the random masks must be identical, not just close
    if name in ('salt_pepper', 'cutout', 'dropout'):
        assert np.array_equal(got == 0.0, ref == 0.0)


def test_agent_preprocess_augments_on_device(lib):
    from carla_driving_rl_agent_amd.core import CARLAgent, FakeCARLAEnvironment
    env = FakeCARLAEnvironment(image_shape=(48, 64, 3), time_horizon=4, num_actions=2, vehicle_features=4, num_waypoints=5,
                               image_range=(0.0, 1.0))
    agent = CARLAgent(env, batch_size=4, aug_intensity=1.0, log_mode=None, seed=7)
    fn = agent.preprocess()
    s = fn(env.reset())
    img = s['state_image']
    assert isinstance(img, torch.Tensor) and img.is_cuda and tuple(img.shape) == (4, 48, 64, 3)
    assert float(img.min()) >= 0.0 and float(img.max()) <= 1.0 + 1e-6         # normalised stack
    s2 = fn(env.reset())
    assert not torch.equal(s2['state_image'], img)
