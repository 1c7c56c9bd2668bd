#!/usr/bin/env python3
"""Generates tests/golden/ref_gae_vectors.npz by RUNNING the reference's own functions.

Run in the build container only (`python tests/golden/make_gae_vectors.py`): it reads
/root/reference/rl/utils.py, which does not exist on the GPU box and is never copied.

`rl/utils.py` cannot be imported (it imports gym / tensorflow / matplotlib at module level), but five
of its functions are plain numpy / scipy / Python:

    np_normalize      rl/utils.py:53-54
    discount_cumsum   rl/utils.py:57-59
    gae               rl/utils.py:62-72      (normalize=False branch; the other calls TF)
    clip              rl/utils.py:101-102
    decompose_number  rl/utils.py:140-151

Their `ast.FunctionDef` nodes are located by name in the parsed module, compiled as they stand (the
source text is not reproduced here or anywhere in this repository) and executed with a namespace that
provides only `np`, `scipy` and `tf_normalize = None`.  The output file holds DATA only: seeded inputs
and what the reference functions returned for them.

How the vectors mirror the reference's call chain (rl/agents/ppo.py:692-727):
  end_trajectory     rewards <- concat(rewards, last_value_as_number); values <- concat(values, last_value)
  compute_returns    rewards_to_go(rewards, gamma) = discount_cumsum(rewards, gamma)[:-1]; the decomposition
                     runs decompose_number per element on the float32 returns (tf.map_fn over to_float(returns))
  compute_advantages gae(rewards, values = base * 10**exp, gamma, lambda)
The float32 casts around the calls (TF's to_float / float32 eager tensors) are applied here exactly where the
reference applies them; decompose_number is fed numpy float32 scalars because tf.map_fn hands it float32
tensors element by element (its `num /= 10.0` is a float32 division there).
"""
import ast
import os
import sys

import numpy as np
import scipy.signal  # noqa: F401  (used by the extracted discount_cumsum through the namespace)
import scipy

REF = '/root/reference/rl/utils.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ref_gae_vectors.npz')
WANTED = ('np_normalize', 'discount_cumsum', 'gae', 'clip', 'decompose_number')


def load_reference_functions(path=REF):
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    nodes = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert sorted(n.name for n in nodes) == sorted(WANTED), [n.name for n in nodes]
    for n in nodes:                      # annotations may name TF types: drop them, the bodies stay untouched
        n.returns = None
        for a in n.args.args + n.args.kwonlyargs:
            a.annotation = None
    mod = ast.Module(body=nodes, type_ignores=[])
    ns = {'np': np, 'scipy': scipy, 'tf_normalize': None}
    exec(compile(ast.fix_missing_locations(mod), path, 'exec'), ns)
    lines = {n.name: (n.lineno, n.end_lineno) for n in nodes}
    return {k: ns[k] for k in WANTED}, lines


def episode(n, seed, spike):
    """Seeded reward / value sequences of one episode (float32, as PPOMemory stores them)."""
    rng = np.random.default_rng(1000 + seed)
    rewards = rng.uniform(-2.0, 10.0, n).astype(np.float32)
    if spike and n > 0:
        rewards[-1] = np.float32(-1000.0)          # the collision penalty of the CARLA environment
    base = rng.uniform(-1.0, 1.0, n).astype(np.float32)
    exp = rng.uniform(0.0, 4.0, n).astype(np.float32)
    last = np.array([0.0, 0.0] if spike else [rng.uniform(-1, 1), rng.uniform(0, 3)], np.float32)
    return rewards, np.stack([base, exp], 1), last


def main():
    fn, lines = load_reference_functions()
    out = {}
    cases = []
    for n, spike, gamma, lam in [(1, False, 0.99, 0.95), (2, True, 0.9999, 0.999), (33, False, 0.9999, 0.999),
                                 (256, True, 0.9999, 0.999), (512, False, 0.99, 0.95), (2049, True, 0.9999, 0.999),
                                 (64, False, 0.99, 0.0)]:
        tag = f'n{n}_s{int(spike)}_l{lam}'
        cases.append(tag)
        rewards, values_be, last = episode(n, n, spike)
        # end_trajectory (rl/agents/ppo.py:692-697)
        boot = np.float32(last[0] * np.power(np.float32(10.0), last[1]))
        r = np.concatenate([rewards, np.array([boot], np.float32)])
        vbe = np.concatenate([values_be, last.reshape(1, 2)])
        # compute_returns (rl/agents/ppo.py:699-712) -> rewards_to_go -> discount_cumsum
        ret64 = fn['discount_cumsum'](r, discount=gamma)[:-1]
        ret32 = ret64.astype(np.float32)
        dec = np.array([fn['decompose_number'](x) for x in ret32], dtype=np.float32).reshape(-1, 2)
        # compute_advantages (rl/agents/ppo.py:714-727) -> gae
        values = (vbe[:, 0] * np.power(np.float32(10.0), vbe[:, 1])).astype(np.float32)
        adv = fn['gae'](r, values, gamma=gamma, lambda_=lam, normalize=False)
        out[f'{tag}.rewards'] = r
        out[f'{tag}.values_be'] = vbe
        out[f'{tag}.values'] = values
        out[f'{tag}.gamma_lambda'] = np.array([gamma, lam], np.float64)
        out[f'{tag}.returns64'] = np.asarray(ret64)
        out[f'{tag}.returns_dec'] = dec
        out[f'{tag}.adv'] = np.asarray(adv)
        out[f'{tag}.adv_dtype'] = np.array(str(np.asarray(adv).dtype))
        out[f'{tag}.np_normalize'] = fn['np_normalize'](np.asarray(adv, np.float32))
    # decompose_number on hand-picked float32 scalars and on Python floats (the two ways the reference can call it)
    probes = np.array([0.0, 1.0, -1.0, 1.0000001, 2.34, -1234.5, 9.999999, 10.0, 1e-8, 99999.99, -1e10, 3.4e38], np.float32)
    out['decompose.in'] = probes
    out['decompose.f32'] = np.array([fn['decompose_number'](x) for x in probes], np.float32)
    out['decompose.pyfloat'] = np.array([fn['decompose_number'](float(x)) for x in probes], np.float64)
    out['clip.in'] = np.array([[-3.0, -1.0, 1.0], [0.5, 0.0, 1.0], [7.0, 0.0, 6.0]], np.float64)
    out['clip.out'] = np.array([fn['clip'](*row) for row in out['clip.in']], np.float64)
    out['cases'] = np.array(cases)
    out['reference_lines'] = np.array([f'{k}:{a}-{b}' for k, (a, b) in sorted(lines.items())])
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes;', ', '.join(out['reference_lines']))


if __name__ == '__main__':
    sys.exit(main())
