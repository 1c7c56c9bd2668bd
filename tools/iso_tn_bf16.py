#!/usr/bin/env python3
"""Isolated timing (HIP events, rotating buffer sets) of the filter-gradient GEMM of a stage-1 unit with the BatchNorm-backward
operand prologue (cdrl_pwconv_bn_bwd_packed would add the backward-data conv; here only the TN product through cdrl_gemm_tn) at
B = 1024 shapes: float32 direct, bf16-operand (float32 tensors) and bf16-storage forms.  CDRL_TN_LDS=0 selects the direct form."""
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


def timeit(fn, nsets, iters=24):
    for k in range(nsets):
        fn(k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(iters):
        fn(k % nsets)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (M, K, N) in [(196608, 116, 116), (675840, 58, 58), (49152, 232, 232), (2703360, 24, 58)]:
    nsets = max(2, int(600e6 / (M * (K + N) * 4)))
    a = [torch.randn(M, K, device=DEV) for _ in range(nsets)]
    d = [torch.randn(M, N, device=DEV) for _ in range(nsets)]
    ab, db = [x.bfloat16() for x in a], [x.bfloat16() for x in d]
    dw = torch.empty(K, N, device=DEV)
    ws = torch.empty(int(lib.cdrl_gemm_tn_workspace_elems(M, N, K)), device=DEV)
    t32 = timeit(lambda k: lib.cdrl_gemm_tn(P(a[k]), K, 0, P(d[k]), N, 0, P(dw), M, N, K, P(ws), 0, S()), nsets)
    lib.cdrl_set_op_activation_type(1)
    t16 = timeit(lambda k: lib.cdrl_gemm_tn(P(ab[k]), K, 0, P(db[k]), N, 0, P(dw), M, N, K, P(ws), 0, S()), nsets)
    lib.cdrl_set_op_activation_type(0)
    by = M * (K + N)
    print(f'M={M} K={K} N={N}: float32 direct {t32:.1f} us ({by * 4 / t32 / 1e3:.0f} GB/s) | bf16 storage {t16:.1f} us ({by * 2 / t16 / 1e3:.0f} GB/s)'
          f'  [CDRL_TN_LDS={os.environ.get("CDRL_TN_LDS", "1")}]')
