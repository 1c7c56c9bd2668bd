#!/usr/bin/env python3
"""Runs the backward of [1x1 conv -> BatchNorm (+ReLU6, shuffled store)] (cdrl_pwconv_bn_bwd: BN-backward reduce + finalize,
backward-data GEMM with the BN-backward prologue, filter-gradient GEMM with both prologues) alone at the learner's shapes, for
rocprofv3 --kernel-trace: isolated durations of pw_nn<PRO=2>, tn_direct_tr and tn_reduce.
Usage: rocprofv3 --kernel-trace --stats ... -- python3 tools/iso_pwbwd.py"""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [(165, 58, 58), (48, 116, 116), (12, 232, 232)]
if len(sys.argv) > 2:
    SHAPES = [SHAPES[int(a)] for a in sys.argv[2:]]
for (px, K, N) in SHAPES:
    G, Mg = 4, B * px
    M = G * Mg
    ctot = 2 * N
    R = 5                                            # rotating operand sets: cold reads
    xs = [torch.randn(M, K, device=DEV) for _ in range(R)]
    w = torch.randn(K, N, device=DEV) / K ** 0.5
    ys = [torch.randn(M, N, device=DEV) for _ in range(R)]
    douts = [torch.randn(M, ctot, device=DEV) for _ in range(R)]
    y = ys[0]
    xst = torch.rand(4, G, K, device=DEV) + 0.5
    stats = torch.zeros(4 * G * N, device=DEV)
    tmp = torch.zeros(M, N, device=DEV)
    ws0 = torch.zeros(G * 256 * 2 * N, dtype=torch.float64, device=DEV)
    gam, bet = torch.rand(N, device=DEV) + 0.5, torch.rand(N, device=DEV)
    mm, mv = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
    _lib.check(lib.cdrl_bn_train_fwd(P(y), G, Mg, N, P(gam), P(bet), P(mm), P(mv), 1, 1, P(tmp), N, 0, 0, P(stats), P(ws0), S()))
    ws = torch.zeros(int(lib.cdrl_pwconv_bn_bwd_workspace_bytes(G, Mg, N, K)), dtype=torch.uint8, device=DEV)
    dg, dbt, coef = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV), torch.zeros(3 * G * N, device=DEV)
    dxs = [torch.zeros(M, K, device=DEV) for _ in range(R)]
    dw, db = torch.zeros(K, N, device=DEV), torch.zeros(N, device=DEV)
    for it in range(2 * R):
        _lib.check(lib.cdrl_pwconv_bn_bwd(P(douts[it % R]), ctot, N, ctot, 1, P(ys[it % R]), P(stats), P(xs[it % R]), K, 0, P(xst), P(w), G, Mg,
                                          N, K, P(dg), P(dbt), P(coef), P(dxs[it % R]), K, 0, 0, P(dw), P(db), P(ws), S()))
    torch.cuda.synchronize()
print('done')
