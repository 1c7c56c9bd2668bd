#!/usr/bin/env python3
"""Runs the filter-gradient GEMM (tn_direct + tn_reduce) alone at the learner's shapes, for rocprofv3 --kernel-trace:
isolated kernel durations.  Usage: rocprofv3 --kernel-trace ... -- python3 tools/iso_tn.py"""
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
for (M, K, N) in [(168960, 58, 58), (49152, 116, 116), (12288, 232, 232), (12288, 464, 768), (168960, 116, 116)]:
    a = torch.randn(M, K, device=DEV)
    y = torch.randn(M, N, device=DEV)
    dw = torch.empty(K, N, device=DEV)
    ws = torch.empty(int(lib.cdrl_gemm_tn_workspace_elems(M, N, K)), device=DEV)
    for _ in range(5):
        _lib.check(lib.cdrl_gemm_tn(P(a), K, 0, P(y), N, 0, P(dw), M, N, K, P(ws), 0, S()))
    torch.cuda.synchronize()
print('done')
