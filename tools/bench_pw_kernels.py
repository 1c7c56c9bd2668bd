#!/usr/bin/env python3
"""Isolated 1x1-conv kernels at the stage-1 shape (K = N = 116, 48 px/frame): float32 MFMA (pw_nn), float32 on the bf16
matrix pipe (pw_x3, exact three-way split) and bf16 activations (pw_bf16, configuration 3), at B = 256 and B = 1024, cold
(launches rotate over 8 buffer sets).  Usage: tools/bench_pw_kernels.py [iters]   (also the target of tools/pmc_pw_kernels.sh)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carla_driving_rl_agent_amd import _lib

lib = _lib.load()
dev = 'cuda:0'
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def timeit(fn, nsets):
    for k in range(nsets):
        fn(k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(iters):
        fn(k % nsets)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for B in (256, 1024):
    G, px, Cc, nsets = 4, 48, 116, 8
    Mg = B * px
    M = G * Mg
    a = [torch.randn(M, Cc, device=dev) for _ in range(nsets)]
    y = [torch.empty(M, Cc, device=dev) for _ in range(nsets)]
    w = torch.randn(Cc, Cc, device=dev)
    bias = torch.randn(Cc, device=dev)
    by = 4.0 * M * 2 * Cc
    fl = 2.0 * M * Cc * Cc
    nb = int(lib.cdrl_pwconv_fused_partial_rows(G, Mg, Cc, Cc))
    part = torch.zeros(G * nb * 2 * Cc, dtype=torch.float64, device=dev)
    t = timeit(lambda k: lib.cdrl_pwconv_fused(P(a[k]), Cc, 0, None, P(w), Cc, 1, P(bias), P(y[k]), Cc, 0, 0, G, Mg, Cc, Cc, 1, None, None,
                                               P(part), S()), nsets)
    print(f'B={B:5d} pw_nn   float32 MFMA            {t * 1e6:7.1f} us  {by / t / 1e9:7.1f} GB/s  {fl / t / 1e12:6.1f} TFLOP/s')
    wp3 = torch.zeros(int(lib.cdrl_pwconv_x3_packed_bytes(Cc)), dtype=torch.uint8, device=dev)
    lib.cdrl_pwconv_x3_pack(P(w), Cc, Cc, Cc, 1, P(wp3), S())
    nb3 = int(lib.cdrl_pwconv_x3_partial_rows(G, Mg, Cc, Cc))
    part3 = torch.zeros(G * nb3 * 2 * Cc, dtype=torch.float64, device=dev)
    t = timeit(lambda k: lib.cdrl_pwconv_x3(P(a[k]), Cc, 0, None, P(wp3), P(bias), P(y[k]), Cc, 0, G, Mg, Cc, Cc, P(part3), S()), nsets)
    print(f'B={B:5d} pw_x3   float32 via 6 bf16 MFMAs {t * 1e6:7.1f} us  {by / t / 1e9:7.1f} GB/s  {fl / t / 1e12:6.1f} TFLOP/s')
    ab = [x.to(torch.bfloat16) for x in a]
    yb = [torch.empty(M, Cc, dtype=torch.bfloat16, device=dev) for _ in range(nsets)]
    nbb = int(lib.cdrl_pwconv_bf16_partial_rows(G, Mg, Cc, Cc))
    partb = torch.zeros(G * nbb * 2 * Cc, dtype=torch.float64, device=dev)
    wpb = torch.zeros(int(lib.cdrl_pwconv_bf16_packed_elems(Cc)), dtype=torch.bfloat16, device=dev)
    lib.cdrl_pwconv_bf16_pack(P(w), Cc, Cc, P(wpb), S())
    t = timeit(lambda k: lib.cdrl_pwconv_bf16(P(ab[k]), Cc, 0, None, None, P(wpb), P(bias), P(yb[k]), Cc, 0, G, Mg, Cc, Cc, P(partb), S()),
               nsets)
    print(f'B={B:5d} pw_bf16 bf16 activations + MFMA  {t * 1e6:7.1f} us  {by / 2 / t / 1e9:7.1f} GB/s  {fl / t / 1e12:6.1f} TFLOP/s')
    del a, y, ab, yb
