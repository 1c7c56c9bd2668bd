"""CPU checks of the augmentation oracle (oracle/augment.py) and of the host-side plan logic."""
import colorsys

import numpy as np

from oracle import augment as A


def test_philox_known_answers():
    """Random123 known-answer vectors for philox4x32-10 (kat_vectors: zeros, all-ones, pi digits)."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, out in kat:
        got = A.philox4x32_10(np.array([ctr], np.uint32), np.array([key], np.uint32))[0]
        assert tuple(int(v) for v in got) == out


def test_hsv_matches_colorsys_and_roundtrips():
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 1, (3, 5, 7, 3))
    h = A.rgb_to_hsv(x)
    for idx in [(0, 0, 0), (1, 2, 3), (2, 4, 6)]:
        assert np.allclose(h[idx], colorsys.rgb_to_hsv(*x[idx]), atol=1e-12)
    assert np.abs(A.hsv_to_rgb(h) - x).max() < 1e-12


def test_identity_plan_and_op_properties():
    rng = np.random.default_rng(1)
    x = rng.uniform(0, 1, (4, 9, 11, 3))
    plan = dict(seed=5, offset=1)
    assert np.array_equal(A.augment(x, plan), x)
    j = A.color_jitter(x, 0.0, 1.0, 1.0, 0.0)
    assert np.abs(j - x).max() < 1e-12                        # neutral jitter
    n = A.normalize(x * 3.0 - 1.0)
    assert abs(n.min()) < 1e-12 and abs(n.max() - 1.0) < 1e-6 and np.all(n.reshape(4, -1).min(axis=1) == 0.0)
    c = A.cutout(x, 6, 14)
    assert (c == 0).any() and np.array_equal(c[c != 0], x[c != 0])
    assert np.array_equal(c[0] == 0, c[3] == 0)               # one mask for the whole stack (reference quirk)
    d = A.coarse_dropout(x, 81, 0.5, 7, 3)
    frac = (d[..., 0] == 0).mean()
    assert 0.3 < frac < 0.7
    sp = A.salt_and_pepper(x, 5.0, 0.5, 7, 3)                 # amount/10 = 0.5 of the pixels replaced by 0 or 1
    changed = np.any(sp != x, axis=-1)
    assert 0.4 < changed.mean() < 0.6 and set(np.unique(sp[changed])) <= {0.0, 1.0}
    g = A.gaussian_noise(x, 1.0, 0.5, 7, 3)
    assert np.all(g >= x) and (g > x).mean() > 0.3            # only positive noise is added (clip after masking)


def test_plan_drawing_is_deterministic_and_gated():
    from carla_driving_rl_agent_amd.rl.augmentations import draw_plan, to_struct, empty_plan
    p0 = draw_plan(0.0, np.random.default_rng(3))
    assert not any(p0[k] for k in ('jitter', 'blur_size', 'salt_pepper', 'gauss_noise', 'normalize', 'cutout_size', 'dropout_size'))
    a = draw_plan(1.0, np.random.default_rng(3), offset=9)
    b = draw_plan(1.0, np.random.default_rng(3), offset=9)
    assert a == b and a['normalize'] == 1 and a['offset'] == 9
    n = 400
    plans = [draw_plan(1.0, np.random.default_rng(s)) for s in range(n)]
    rate = lambda k: sum(1 for p in plans if p[k]) / n
    assert rate('jitter') > 0.95 and 0.15 < rate('blur_size') < 0.35 and 0.1 < rate('salt_pepper') < 0.3
    assert 0.22 < rate('gauss_noise') < 0.44 and 0.07 < rate('cutout_size') < 0.24 and 0.07 < rate('dropout_size') < 0.24
    s = to_struct(a)
    assert s.seed == a['seed'] and s.normalize == 1 and abs(s.contrast - a['contrast']) < 1e-6
    assert len(empty_plan()['blur_kernel']) == 75
