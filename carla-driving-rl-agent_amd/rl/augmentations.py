"""Rollout-time image augmentation (reference rl/augmentations/* as used by CARLAgent.augment,
core/carla_agent.py:527-579) on the native kernels.

`draw_plan(alpha, rng)` makes the same sequence of random decisions as the reference's `augment_fn` (one
`tf_chance` per op compared with the op's probability times the intensity `alpha`, then the op's own random scalars);
`augment_images(images, plan)` applies the plan on the device through `cdrl_augment_images`.  The random *fields*
(salt & pepper masks, gaussian noise, dropout grid) are generated on the device from (plan seed, plan offset)."""
import ctypes as C

import numpy as np
import torch

from .. import _lib


class AugPlan(C.Structure):
    _fields_ = [('jitter', C.c_int), ('brightness', C.c_float), ('contrast', C.c_float), ('saturation', C.c_float),
                ('hue', C.c_float), ('blur_size', C.c_int), ('blur_kernel', C.c_float * 75), ('salt_pepper', C.c_int),
                ('sp_amount', C.c_float), ('sp_prob', C.c_float), ('gauss_noise', C.c_int), ('gn_amount', C.c_float),
                ('gn_std', C.c_float), ('normalize', C.c_int), ('cutout_size', C.c_int), ('cutout_cell', C.c_int),
                ('dropout_size', C.c_int), ('dropout_amount', C.c_float), ('seed', C.c_uint64), ('offset', C.c_uint64)]


def empty_plan(seed=0, offset=0) -> dict:
    return dict(jitter=0, brightness=0.0, contrast=1.0, saturation=1.0, hue=0.0, blur_size=0, blur_kernel=[0.0] * 75,
                salt_pepper=0, sp_amount=0.1, sp_prob=0.5, gauss_noise=0, gn_amount=0.1, gn_std=0.075, normalize=0,
                cutout_size=0, cutout_cell=0, dropout_size=0, dropout_amount=0.04, seed=int(seed), offset=int(offset))


def draw_plan(alpha: float, rng: np.random.Generator, offset=0) -> dict:
    """Random decisions of CARLAgent.augment_fn (core/carla_agent.py:549-576) for intensity `alpha`."""
    plan = empty_plan(seed=int(rng.integers(0, 2 ** 63 - 1)), offset=offset)
    if alpha <= 0.0:
        return plan
    chance = lambda: float(rng.uniform(0.0, 1.0))
    if chance() < alpha:                                     # simclr.color_jitter(strength=alpha)
        plan.update(jitter=1, brightness=float(rng.uniform(-0.2 * alpha, 0.2 * alpha)),
                    contrast=float(rng.uniform(1.0 - 0.8 * alpha, 1.0 + 0.8 * alpha)),
                    saturation=float(rng.uniform(1.0 - 0.8 * alpha, 1.0 + 0.8 * alpha)),
                    hue=float(rng.uniform(-0.2 * alpha, 0.2 * alpha)))
    if chance() < 0.25 * alpha:                              # tf_gaussian_blur(size=3|5): random N(1, 0.25) kernel
        k = 3 if chance() >= 0.5 else 5
        kern = rng.normal(1.0, 0.25, size=(k, k, 3)).astype(np.float32).reshape(-1)
        plan.update(blur_size=k, blur_kernel=list(kern) + [0.0] * (75 - kern.size))
    if chance() < 0.2 * alpha:
        plan.update(salt_pepper=1, sp_amount=0.1, sp_prob=0.5)
    if chance() < 0.33 * alpha:
        plan.update(gauss_noise=1, gn_amount=0.10, gn_std=0.075)
    plan.update(normalize=1)
    if chance() < 0.15 * alpha:                              # tf_cutout_batch(size=6): the argmax cell of a random grid
        plan.update(cutout_size=6, cutout_cell=int(rng.integers(0, 36)))
    if chance() < 0.15 * alpha:
        plan.update(dropout_size=81, dropout_amount=0.04)
    return plan


def to_struct(plan: dict) -> AugPlan:
    p = AugPlan()
    for k, v in plan.items():
        if k == 'blur_kernel':
            for i, x in enumerate(v):
                p.blur_kernel[i] = float(x)
        else:
            setattr(p, k, v)
    return p


class Augmenter:
    """Holds the workspace / output buffers for one observation-stack shape."""

    def __init__(self, device='cuda:0'):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self._ws = None
        self._shape = None

    def __call__(self, images, plan: dict) -> torch.Tensor:
        x = torch.as_tensor(images, dtype=torch.float32).to(self.device).contiguous()
        if x.dim() != 4 or x.shape[-1] != 3:
            raise ValueError(f'expected an image stack (T, H, W, 3), got {tuple(x.shape)}')
        T, H, W, _ = x.shape
        if self._shape != (T, H, W):
            self._ws = torch.empty(int(self.lib.cdrl_augment_workspace_floats(T, H, W)), device=self.device)
            self._shape = (T, H, W)
        out = torch.empty_like(x)
        st = to_struct(plan)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self.lib.cdrl_augment_images(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), T, H, W, C.byref(st),
                                                C.c_void_p(self._ws.data_ptr()), stream), 'cdrl_augment_images')
        return out
