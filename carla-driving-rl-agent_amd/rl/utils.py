"""Numeric + bookkeeping helpers of the learner path, mirroring the symbols of the reference's
rl/utils.py that the hot path uses (names and argument meaning kept; TensorFlow replaced by
device kernels of libcdrl_hip.so, torch tensors as plain device memory).

  discount_cumsum / gae / rewards_to_go / decompose_number / tf_sp_norm  -> cdrl_gae_returns
  data_to_batches                                                         -> index pipeline +
                                                                             cdrl_gather_rows
  space_to_flat_spec, to_tensor, Summary, makedir                         -> host bookkeeping
"""
import ctypes as C
import json
import os
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from .. import _lib
from . import spaces

NP_EPS = np.finfo(np.float32).eps
EPSILON = float(NP_EPS)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def makedir(*args) -> str:
    path = os.path.join(*args)
    os.makedirs(path, exist_ok=True)
    return path


# ------------------------------------------------------------------------------------------------
# returns / advantages (device)
# ------------------------------------------------------------------------------------------------

def _f32(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.float32).contiguous()
    return torch.as_tensor(np.asarray(x, dtype=np.float32), device=device)


def returns_and_advantages(rewards, values_be, gamma: float, lambda_: float, scale: float = 2.0, device='cuda:0'):
    """One device launch for PPOMemory.compute_returns + compute_advantages.
    rewards (N+1,), values_be (N+1, 2) incl. the bootstrap entry.
    -> dict(returns (N,), returns_be (N,2), advantages_raw (N,), advantages (N,))"""
    from ..engine import gae_returns
    r, v = _f32(rewards, device), _f32(values_be, device)
    ret, ret_be, adv_raw, adv = gae_returns(r, v, gamma, lambda_, scale)
    return dict(returns=ret, returns_be=ret_be, advantages_raw=adv_raw, advantages=adv)


def discount_cumsum(x, discount: float, device='cuda:0'):
    """y[i] = x[i] + discount * y[i+1] (float64 recurrence, like scipy.signal.lfilter in the reference)."""
    x = _f32(x, device)
    n = x.numel()
    padded = torch.cat([x, torch.zeros(1, device=x.device)])
    dummy = torch.zeros((n + 1, 2), device=x.device)
    return returns_and_advantages(padded, dummy, discount, 0.0, 1.0, x.device)['returns']


def rewards_to_go(rewards, discount: float, decompose=False, device='cuda:0'):
    r = _f32(rewards, device)
    out = returns_and_advantages(r, torch.zeros((r.numel(), 2), device=r.device), discount, 0.0, 1.0, r.device)
    if decompose:
        return out['returns_be'], out['returns']
    return out['returns']


def gae(rewards, values, gamma: float, lambda_: float, normalize=False, device='cuda:0'):
    """`values` are scalar state values (N+1,); returns the raw GAE advantages (N,)."""
    v = _f32(values, device)
    vbe = torch.stack([v, torch.zeros_like(v)], dim=1)         # value = base * 10^0
    out = returns_and_advantages(rewards, vbe, gamma, lambda_, 1.0, v.device)
    adv = out['advantages_raw']
    if normalize:
        adv = (adv - adv.mean()) / (adv.std(unbiased=False) + EPSILON)
    return adv


def decompose_number(num: float) -> Tuple[float, float]:
    """n = base * 10^exponent with |base| <= 1 (float32 repeated division by ten)."""
    x = np.float32(num)
    e = 0
    while abs(x) > np.float32(1.0):
        x = np.float32(x / np.float32(10.0))
        e += 1
    return float(x), float(e)


def tf_sp_norm(x: torch.Tensor, eps=1e-3):
    pos = x * (x > 0)
    neg = x * (x < 0)
    return pos / (x.max() + eps) + neg / -(x.min() - eps)


# ------------------------------------------------------------------------------------------------
# spaces / tensors
# ------------------------------------------------------------------------------------------------

def space_to_flat_spec(space, name: str) -> Dict[str, tuple]:
    """Box/Discrete -> {name: shape}; Dict -> {name_key: shape, ...} (nested keys joined by '_')."""
    if isinstance(space, spaces.Discrete):
        return {name: (space.n,)}
    if isinstance(space, spaces.Box):
        return {name: tuple(space.shape)}
    if isinstance(space, spaces.Dict):
        out = {}
        for key, sub in space.spaces.items():
            out.update(space_to_flat_spec(sub, f'{name}_{key}'))
        return out
    raise ValueError('space must be one of Box, Discrete or Dict')


def to_tensor(x, expand_axis=0, device='cuda:0'):
    if isinstance(x, dict):
        return {k: to_tensor(v, expand_axis, device) for k, v in x.items()}
    t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x, dtype=np.float32))
    return t.to(device=device, dtype=torch.float32).unsqueeze(expand_axis)


# ------------------------------------------------------------------------------------------------
# minibatching
# ------------------------------------------------------------------------------------------------

def batch_indices(n: int, batch_size: int, skip=0, shuffle=False, num_shards=1, drop_remainder=False,
                  rng: Optional[np.random.Generator] = None) -> List[np.ndarray]:
    """Row order produced by the reference's tf.data pipeline (rl/utils.py:365-393):
    skip(skip) -> shuffle(buffer_size=batch_size) -> shard/concatenate -> batch(batch_size, drop_remainder).
    The shuffle is tf.data's streaming buffer shuffle: keep a buffer of `batch_size` rows, emit a
    random one, refill from the stream."""
    order = list(range(skip, n))
    if shuffle:
        rng = rng or np.random.default_rng()
        buf, out, pos = order[:batch_size], [], min(batch_size, len(order))
        while buf:
            j = int(rng.integers(len(buf)))
            out.append(buf[j])
            if pos < len(order):
                buf[j] = order[pos]
                pos += 1
            else:
                buf.pop(j)
        order = out
    if num_shards > 1:
        order = [x for s in range(num_shards) for x in order[s::num_shards]]
    batches = [np.asarray(order[i:i + batch_size], dtype=np.int32) for i in range(0, len(order), batch_size)]
    if drop_remainder:
        batches = [b for b in batches if len(b) == batch_size]
    return [b for b in batches if len(b)]


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """dst[i] = src[idx[i]] through cdrl_gather_rows (device index-gather)."""
    lib = _lib.load()
    src = src.contiguous()
    n = idx.numel()
    row = int(np.prod(src.shape[1:])) if src.dim() > 1 else 1
    dst = torch.empty((n,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    if n == 0 or row == 0:
        return dst
    _lib.check(lib.cdrl_gather_rows(_lib.ptr(src), _lib.ptr(idx), _lib.ptr(dst), n, row, _stream()), 'cdrl_gather_rows')
    return dst


def _gather_struct(t, idx):
    if isinstance(t, dict):
        return {k: _gather_struct(v, idx) for k, v in t.items()}
    return gather_rows(t, idx)


def data_to_batches(tensors: Union[List, Tuple], batch_size: int, shuffle_batches=False, seed=None,
                    drop_remainder=False, map_fn=None, prefetch_size=2, num_shards=1, skip=0, shuffle=False):
    """Generator of minibatches with the structure of `tensors` (tuple of tensors / dicts of tensors)."""
    first = tensors[0]
    while isinstance(first, dict):
        first = next(iter(first.values()))
    n = first.shape[0]
    rng = np.random.default_rng(seed)
    batches = batch_indices(n, batch_size, skip, shuffle, num_shards, drop_remainder, rng)
    if shuffle_batches:
        rng.shuffle(batches)
    for b in batches:
        idx = torch.as_tensor(b, device=first.device)
        out = tuple(_gather_struct(t, idx) for t in tensors)
        yield map_fn(*out) if map_fn is not None else out


# ------------------------------------------------------------------------------------------------
# logging
# ------------------------------------------------------------------------------------------------

class Summary:
    """log(**kw) buffers values; write_summaries() flushes them as one JSON line per key
    (the reference writes TensorBoard scalars; same log()/write_summaries() surface)."""

    def __init__(self, mode='summary', name=None, keys: List[str] = None, summary_dir='logs'):
        self.mode = mode
        self.stats: Dict[str, list] = {}
        self.allowed = set(keys) if keys is not None else None
        self.step = 0
        self.path = None
        if mode is not None:
            self.path = os.path.join(makedir(summary_dir, name or 'agent'), 'summary.jsonl')

    def log(self, **kwargs):
        if self.mode is None:
            return
        for k, v in kwargs.items():
            if self.allowed is not None and k not in self.allowed:
                continue
            self.stats.setdefault(k, []).append(v)

    @staticmethod
    def _scalar(v):
        if isinstance(v, torch.Tensor):
            return v.detach().float().mean().item()
        if isinstance(v, (list, tuple)):
            return float(np.mean([Summary._scalar(x) for x in v])) if len(v) else 0.0
        return float(np.mean(v))

    def write_summaries(self):
        if self.mode is None or not self.stats:
            self.stats = {}
            return
        with open(self.path, 'a') as f:
            for k, vals in self.stats.items():
                f.write(json.dumps(dict(step=self.step, key=k, mean=float(np.mean([self._scalar(v) for v in vals])),
                                        n=len(vals))) + '\n')
        self.step += 1
        self.stats = {}


def swish6(x):
    return torch.clamp_max(x * torch.sigmoid(x), 6.0)


def softplus(value=1.0):
    def activation(x):
        return torch.nn.functional.softplus(x) + value
    return activation
