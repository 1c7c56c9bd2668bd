#!/usr/bin/env python3
"""Isolated timing of the wide pointwise conv (gemm_pw_wide.hip) against the register-resident-W form (gemm_pw.hip) at the stage-2
shapes of the benchmark (K = N = 232; M = 12288 at 3x4 pixels, 49152 at 6x8), rotating buffer sets.  Usage: python tools/iso_pww.py [B]"""
import ctypes as C
import sys
sys.path.insert(0, '.')
import torch
from carla_driving_rl_agent_amd import _lib
lib = _lib.load()
dev = torch.device('cuda', 0)
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
G, K, N = 4, 232, 232
nsets = 8


def timeit(fn, it=40):
    for k in range(nsets):
        fn(k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(it):
        fn(k % nsets)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3


for px in (12, 48):
    Mg = B * px
    M = G * Mg
    a = [torch.randn(M, K, device=dev) for _ in range(nsets)]
    c = [torch.empty(M, N, device=dev) for _ in range(nsets)]
    w = torch.randn(K, N, device=dev) / K ** 0.5
    bias = torch.randn(N, device=dev)
    st = torch.rand(4 * G * K, device=dev) + 0.5
    wp = torch.zeros(int(lib.cdrl_gemm_x3_packed_bytes(N, K)), dtype=torch.uint8, device=dev)
    _lib.check(lib.cdrl_gemm_x3_pack(P(w), K, N, N, 1, P(wp), S()))
    nbw = int(lib.cdrl_pwconv_wide_partial_rows(G, Mg, N, K))
    partw = torch.zeros(G * nbw * 2 * N, dtype=torch.float64, device=dev)
    nbo = int(lib.cdrl_pwconv_fused_partial_rows(G, Mg, N, K))
    parto = torch.zeros(G * nbo * 2 * N, dtype=torch.float64, device=dev)
    by = 4.0 * M * (K + N)
    for pro in (None, st):
        tw = timeit(lambda k: lib.cdrl_pwconv_wide(P(a[k]), K, 0, P(pro), P(wp), P(bias), P(c[k]), N, 0, G, Mg, N, K, P(partw), S()))
        to = timeit(lambda k: lib.cdrl_pwconv_fused(P(a[k]), K, 0, P(pro), P(w), N, 1, P(bias), P(c[k]), N, 0, 0, G, Mg, N, K, 1, None, None, P(parto), S()))
        print(f'M={M} K=N={K} pro={pro is not None}: wide {tw * 1e6:.1f} us ({by / tw / 1e9:.0f} GB/s, {nbw} partial rows)   '
              f'register-resident W {to * 1e6:.1f} us ({by / to / 1e9:.0f} GB/s, {nbo} partial rows)')
