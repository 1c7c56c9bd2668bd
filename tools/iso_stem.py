#!/usr/bin/env python3
"""Runs the stem block backward alone at the benchmark shape (B=256, T=4, 90x120) for rocprofv3 --kernel-trace."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carla_driving_rl_agent_amd import _lib  # noqa: E402

lib = _lib.load()
DEV = 'cuda:0'
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
B, T, H, W, Cc = 256, 4, 90, 120, 24
N = B * T
Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
Hp, Wp = -(-Ho // 2), -(-Wo // 2)
x = torch.rand(B, T, H, W, 3, device=DEV)
w = torch.randn(3, 3, 3, Cc, device=DEV) * 0.3
b = torch.randn(Cc, device=DEV)
y = torch.empty(N, Ho, Wo, Cc, device=DEV)
lib.cdrl_stem_fwd(P(x), P(w), P(b), P(y), B, T, H, W, Cc, S())
g, be = torch.ones(Cc, device=DEV), torch.zeros(Cc, device=DEV)
mm, mv = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
stats = torch.zeros(4 * T * Cc, device=DEV)
ws0 = torch.zeros(T * 256 * 2 * Cc, dtype=torch.float64, device=DEV)
scratch = torch.empty(N * Ho * Wo, Cc, device=DEV)
lib.cdrl_bn_train_fwd(P(y), T, B * Ho * Wo, Cc, P(g), P(be), P(mm), P(mv), 1, 1, P(scratch), Cc, 0, 0, P(stats), P(ws0), S())
pool = torch.empty(N, Hp, Wp, Cc, device=DEV)
am = torch.zeros(N, Hp, Wp, Cc, dtype=torch.uint8, device=DEV)
lib.cdrl_maxpool_bn_fwd(P(y), P(stats), T, B, P(pool), P(am), N, Ho, Wo, Cc, S())
dp = torch.randn(N, Hp, Wp, Cc, device=DEV)
ws = torch.zeros(int(lib.cdrl_stem_block_bwd_workspace_doubles(B, T, H, W, Cc)), dtype=torch.float64, device=DEV)
dg, dbt, coef = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV), torch.zeros(3 * T * Cc, device=DEV)
dw, db = torch.zeros(3, 3, 3, Cc, device=DEV), torch.zeros(Cc, device=DEV)
for _ in range(5):
    _lib.check(lib.cdrl_stem_block_bwd(P(x), P(y), P(stats), P(am), P(dp), B, T, H, W, Cc, P(dg), P(dbt), P(coef), P(dw), P(db), P(ws), S()))
torch.cuda.synchronize()
print('done')
