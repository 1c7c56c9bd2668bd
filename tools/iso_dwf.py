#!/usr/bin/env python3
"""Runs the fused depthwise-block backward (and forward) alone at the learner's shapes, for rocprofv3 --kernel-trace:
isolated kernel durations (no side-stream contention).  Usage: rocprofv3 --kernel-trace ... -- python3 tools/iso_dwf.py"""
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
T, B = 4, 256
SHAPES = [(11, 15, 58, 1, 1), (6, 8, 116, 1, 1), (3, 4, 232, 1, 1), (22, 30, 58, 2, 1), (22, 30, 24, 2, 0), (11, 15, 116, 2, 1), (6, 8, 232, 2, 1)]
if len(sys.argv) > 1:
    SHAPES = [SHAPES[int(a)] for a in sys.argv[1:]]
for (H, W, Cc, stride, pre) in SHAPES:
    N = T * B
    Ho, Wo = -(-H // stride), -(-W // stride)
    R = 5                                                    # rotating tensor sets: every launch reads cold data (> 256 MB apart)
    xs = [torch.randn(N, H, W, Cc, device=DEV) for _ in range(R)]
    w = torch.randn(3, 3, Cc, 1, device=DEV)
    b = torch.randn(Cc, device=DEV)
    ys = [torch.empty(N, Ho, Wo, Cc, device=DEV) for _ in range(R)]
    douts = [torch.randn(N, Ho, Wo, Cc, device=DEV) for _ in range(R)]
    pre_stats = torch.rand(4 * T * Cc, device=DEV) + 0.5 if pre else None
    post_stats = torch.zeros(4 * T * Cc, device=DEV)
    g = torch.ones(Cc, device=DEV); be = torch.zeros(Cc, device=DEV); mm = torch.zeros(Cc, device=DEV); mv = torch.ones(Cc, device=DEV)
    ws = torch.zeros(int(lib.cdrl_dwconv_bn_workspace_doubles(T, B, H, W, Cc, stride)), dtype=torch.float64, device=DEV)
    dxs = [torch.empty_like(xs[0]) for _ in range(R)]; dw = torch.empty_like(w); db = torch.empty_like(b)
    vecs = [torch.zeros(Cc, device=DEV) for _ in range(4)]
    coefs = [torch.zeros(3 * T * Cc, device=DEV) for _ in range(2)]
    for it in range(2 * R):
        x, y, dout, dx = xs[it % R], ys[it % R], douts[it % R], dxs[it % R]
        _lib.check(lib.cdrl_dwconv_bn_fwd(P(x), P(pre_stats), P(w), P(b), P(y), T, B, H, W, Cc, stride, P(g), P(be), P(mm), P(mv), 1, P(post_stats), P(ws), S()))
        _lib.check(lib.cdrl_dwconv_bn_bwd(P(x), P(pre_stats), P(dout), P(y), P(post_stats), P(w), T, B, H, W, Cc, stride, P(dx), P(dw), P(db),
                                          P(vecs[0]), P(vecs[1]), P(coefs[0]), P(vecs[2]), P(vecs[3]), P(coefs[1]), P(ws), S()))
    torch.cuda.synchronize()
print('done')
