#!/usr/bin/env python3
"""Runs the pointwise-conv backward with BN-backward operand prologues (cdrl_pwconv_bn_bwd: BN sums, finalize, backward-data
GEMM with prologue, bias reduce, filter-gradient GEMM with prologue) alone at the stage-1 shape, for rocprofv3 --kernel-trace."""
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
G = 4
for (Mg, K, N, shuffle, dxhalf) in [(12288, 116, 116, 0, 0), (12288, 116, 116, 0, 1), (12288, 116, 116, 1, 0), (42240, 58, 58, 0, 1)]:
    M = G * Mg
    ctot = 2 * N if shuffle else N
    coff = N if shuffle else 0
    dout = torch.randn(M, ctot, device=DEV)
    y = torch.randn(M, N, device=DEV)
    x = torch.randn(M, K, device=DEV)
    w = torch.randn(K, N, device=DEV)
    stats = torch.cat([torch.zeros(G * N), torch.ones(G * N), torch.ones(G * N), torch.full((G * N,), 3.0)]).to(DEV)
    ws = torch.zeros(int(lib.cdrl_pwconv_bn_bwd_workspace_bytes(G, Mg, N, K)), dtype=torch.uint8, device=DEV)
    dg, dbt, coef = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV), torch.zeros(3 * G * N, device=DEV)
    ldx = 2 * K if dxhalf else K
    dx = torch.zeros(M, ldx, device=DEV)
    dw, db = torch.zeros(K, N, device=DEV), torch.zeros(N, device=DEV)
    for _ in range(5):
        _lib.check(lib.cdrl_pwconv_bn_bwd(P(dout), ctot, coff, ctot if shuffle else 0, 1, P(y), P(stats), P(x), K, 0, None,
                                          P(w), G, Mg, N, K, P(dg), P(dbt), P(coef), P(dx), ldx, K if dxhalf else 0, 0, P(dw), P(db), P(ws), S()))
    torch.cuda.synchronize()
print('done')
