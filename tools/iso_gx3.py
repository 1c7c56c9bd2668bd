#!/usr/bin/env python3
"""Isolated timing of gemm_x3 (the 464 -> 768 head conv and its backward-data product) at the benchmark's shapes, rotating buffer sets.
Usage (GPU box): python tools/iso_gx3.py [B]"""
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
M = B * 4 * 12
nsets = 6


def run(K, N, check=False):
    a = [torch.randn(M, K, device=dev) for _ in range(nsets)]
    c = [torch.empty(M, N, device=dev) for _ in range(nsets)]
    w = torch.randn(K, N, device=dev) / K ** 0.5
    bias = torch.randn(N, device=dev)
    wp = torch.zeros(int(lib.cdrl_gemm_x3_packed_bytes(N, K)), dtype=torch.uint8, device=dev)
    _lib.check(lib.cdrl_gemm_x3_pack(P(w), K, N, N, 1, P(wp), S()))
    fn = lambda k: lib.cdrl_gemm_x3(P(a[k]), K, 0, P(wp), P(bias), P(c[k]), N, 0, M, N, K, 0, S())
    for k in range(nsets):
        fn(k)
    if check:
        ref = (a[0].double() @ w.double() + bias.double())
        err = float((c[0].double() - ref).abs().max() / ref.abs().max())
        print(f'  max rel err {err:.2e}')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 30
    e0.record()
    for k in range(it):
        fn(k % nsets)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / it * 1e-3
    fl = 2.0 * M * K * N
    print(f'M={M} K={K} N={N}: {t * 1e6:.1f} us  {fl / t / 1e12:.1f} TFLOP/s (float32-equivalent; x6 on the bf16 pipe: {6 * fl / t / 1e12:.0f})  '
          f'bytes {(M * K + M * N) * 4 / 1e6:.0f} MB -> {(M * K + M * N) * 4 / t / 1e9:.0f} GB/s')


for K, N in ((464, 768), (768, 464), (232, 232)):
    run(K, N, check=True)
