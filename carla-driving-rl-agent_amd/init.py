"""Keras-default initial weights for the engine's parameter arenas (host side, numpy).

Mirrors the initialisers the reference's layer constructors ask for: glorot-uniform kernels,
`bias_initializer='glorot_uniform'` on the Dense / GRU layers that say so (reference
core/architectures.py:20, core/networks.py:30,47-50,64,120-124,260-273), zeros elsewhere,
orthogonal GRU recurrent kernels, BatchNorm gamma=1 / beta=0 / moving_mean=0 / moving_var=1.
"""
import math

import numpy as np

_GLOROT_BIAS_PREFIXES = ('road.', 'vehicle.', 'navigation.', 'dyn.fc', 'pi.fc', 'v.fc', 'pi.similarity', 'pi.speed',
                         'v.base', 'v.exp', 'v.speed', 'v.similarity')


def _glorot(rng, shape):
    if len(shape) == 4:
        rf = shape[0] * shape[1]
        fan_in, fan_out = shape[2] * rf, shape[3] * rf
    elif len(shape) == 2:
        fan_in, fan_out = shape
    else:
        fan_in = fan_out = shape[0]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape)


def _orthogonal(rng, shape):
    n = max(shape)
    q, r = np.linalg.qr(rng.standard_normal((n, n)))
    q = q * np.sign(np.diag(r))
    return q[:shape[0], :shape[1]]


def initial_value(name, shape, rng):
    if name.endswith('.gamma') or name.endswith('.moving_var'):
        return np.ones(shape)
    if name.endswith('.beta') or name.endswith('.moving_mean'):
        return np.zeros(shape)
    if name.endswith('.recurrent'):
        return _orthogonal(rng, shape)
    if name.endswith('.kernel') or name.endswith('.w'):
        return _glorot(rng, shape)
    if name.endswith('.bias'):           # GRU bias (2, 3u), glorot_uniform
        return _glorot(rng, shape)
    if name.endswith('.b'):
        if name.startswith(_GLOROT_BIAS_PREFIXES):
            return _glorot(rng, shape)
        return np.zeros(shape)           # Conv2D / DepthwiseConv2D / alpha / beta heads: zeros
    raise ValueError(f'no initialiser rule for {name}')


def init_engine_parameters(engine, seed=42):
    rng = np.random.default_rng(seed)
    for model in ('trunk', 'policy', 'value'):
        values = {e['name']: initial_value(e['name'], e['shape'], rng).astype(np.float32)
                  for e in engine.tables[model].entries}
        engine.load_params(model, values)
    engine.update_old_policy()
    engine.reset_optimizer()
