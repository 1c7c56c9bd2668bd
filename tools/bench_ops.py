#!/usr/bin/env python3
"""Per-op micro-benchmarks through the C ABI at the learner's real shapes (B=256, T=4, 90x120).
Prints achieved GB/s (algorithmic bytes) and TFLOP/s per op; used to tune kernels against the
gfx950 rooflines.  Usage (GPU box): python tools/bench_ops.py [gemm|bn|dw|all]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carla_driving_rl_agent_amd import _lib  # noqa: E402

lib = _lib.load()
DEV = 'cuda:0'


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


FRAMES = 1024
PW = [  # (pixels/frame, K, N, count per pass)
    (660, 24, 58, 1), (165, 58, 92, 1), (165, 24, 24, 1), (165, 58, 58, 6), (165, 116, 116, 1), (48, 116, 116, 16),
    (48, 232, 232, 1), (12, 232, 232, 8), (12, 464, 768, 1)]


def bench_gemm():
    tot = dict(fwd=0.0, bwd_data=0.0, bwd_filter=0.0)
    print(f'{"shape":<26}{"fwd us":>9}{"TF/s":>7}{"GB/s":>7} |{"bwdD us":>9}{"TF/s":>7} |{"bwdF us":>9}{"TF/s":>7}')
    for px, K, N, cnt in PW:
        M = FRAMES * px
        a = torch.randn(M, K, device=DEV)
        w = torch.randn(K, N, device=DEV)
        b = torch.randn(N, device=DEV)
        y = torch.empty(M, N, device=DEV)
        da = torch.empty(M, K, device=DEV)
        dw = torch.empty(K, N, device=DEV)
        ws = torch.empty(int(lib.cdrl_gemm_tn_workspace_elems(M, N, K)), device=DEV)
        f = timeit(lambda: lib.cdrl_gemm_nn(P(a), K, 0, P(w), N, 1, P(b), P(y), N, 0, M, N, K, 0, S()))
        d = timeit(lambda: lib.cdrl_gemm_nn(P(y), N, 0, P(w), 1, N, None, P(da), K, 0, M, K, N, 0, S()))
        t = timeit(lambda: lib.cdrl_gemm_tn(P(a), K, 0, P(y), N, 0, P(dw), M, N, K, P(ws), 0, S()))
        fl = 2.0 * M * K * N
        by = 4.0 * M * (K + N)
        print(f'M={M:<7} K={K:<4} N={N:<4} x{cnt:<2} {f:9.1f}{fl / f / 1e6:7.1f}{by / f / 1e3:7.0f} |{d:9.1f}{fl / d / 1e6:7.1f} |{t:9.1f}{fl / t / 1e6:7.1f}')
        tot['fwd'] += f * cnt
        tot['bwd_data'] += d * cnt
        tot['bwd_filter'] += t * cnt
    print('per pass (us):', {k: round(v) for k, v in tot.items()}, 'sum', round(sum(tot.values())))


def bench_bn():
    print(f'{"BN shape":<30}{"fwd us":>9}{"GB/s":>7} |{"bwd us":>9}{"GB/s":>7}')
    for px, Cc in [(2596, 24), (660, 58), (165, 58), (165, 92), (48, 116), (12, 232), (12, 768)]:
        G, Mg = 4, 256 * px
        y = torch.randn(G * Mg, Cc, device=DEV)
        gam, bet = torch.ones(Cc, device=DEV), torch.zeros(Cc, device=DEV)
        mm, mv = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
        out = torch.empty_like(y)
        stats = torch.empty(4 * G * Cc, device=DEV)
        coef = torch.empty(3 * G * Cc, device=DEV)
        ws = torch.empty(G * 256 * 2 * Cc, dtype=torch.float64, device=DEV)
        dg, db = torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV)
        dy = torch.empty_like(y)
        f = timeit(lambda: lib.cdrl_bn_train_fwd(P(y), G, Mg, Cc, P(gam), P(bet), P(mm), P(mv), 1, 1, P(out), Cc, 0, 0,
                                                  P(stats), P(ws), S()))
        b = timeit(lambda: lib.cdrl_bn_train_bwd(P(out), Cc, 0, 0, P(y), G, Mg, Cc, P(stats), 1, P(dg), P(db), P(dy), P(coef),
                                                  P(ws), S()))
        n = y.numel() * 4
        print(f'rows={G * Mg:<9} C={Cc:<5} {f:9.1f}{3 * n / f / 1e3:7.0f} |{b:9.1f}{5 * n / b / 1e3:7.0f}')


def bench_dw():
    print(f'{"dw shape":<34}{"fwd us":>9}{"GB/s":>7} |{"bwdD us":>9} |{"bwdF us":>9}')
    for H, W, Cc, s in [(22, 30, 58, 2), (22, 30, 24, 2), (11, 15, 58, 1), (11, 15, 116, 2), (6, 8, 116, 1), (6, 8, 232, 2), (3, 4, 232, 1)]:
        N = FRAMES
        Ho, Wo = -(-H // s), -(-W // s)
        a = torch.randn(N, H, W, Cc, device=DEV)
        w = torch.randn(3, 3, Cc, 1, device=DEV)
        b = torch.randn(Cc, device=DEV)
        y = torch.empty(N, Ho, Wo, Cc, device=DEV)
        da = torch.empty_like(a)
        dw = torch.empty_like(w)
        db = torch.empty_like(b)
        ws = torch.empty(int(lib.cdrl_dwconv_bwd_workspace_doubles(N, H, W, Cc, s)), dtype=torch.float64, device=DEV)
        f = timeit(lambda: lib.cdrl_dwconv_fwd(P(a), P(w), P(b), P(y), N, H, W, Cc, s, S()))
        d = timeit(lambda: lib.cdrl_dwconv_bwd_data(P(y), P(w), P(da), N, H, W, Cc, s, S()))
        t = timeit(lambda: lib.cdrl_dwconv_bwd_filter(P(a), P(y), P(dw), P(db), N, H, W, Cc, s, P(ws), S()))
        n = (a.numel() + y.numel()) * 4
        print(f'{H}x{W}x{Cc} s{s:<22} {f:9.1f}{n / f / 1e3:7.0f} |{d:9.1f} |{t:9.1f}')


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if which in ('gemm', 'all'):
        bench_gemm()
    if which in ('bn', 'all'):
        bench_bn()
    if which in ('dw', 'all'):
        bench_dw()


def bench_pw():
    """fused persistent pointwise GEMM vs the generic tiled gemm_nn at the tower's K, N <= 128 shapes"""
    print(f'{"shape":<28}{"gemm_nn us":>11}{"pw us":>9}{"pw+stats":>10}{"pw+pro+st":>10}{"ideal us":>9}')
    for px, K, N in [(660, 24, 58), (165, 58, 92), (165, 24, 24), (165, 58, 58), (48, 116, 116), (165, 92, 58), (660, 58, 24), (12, 232, 232), (12, 116, 232)]:
        G, Mg = 4, 256 * px
        M = G * Mg
        a = torch.randn(M, K, device=DEV)
        w = torch.randn(K, N, device=DEV)
        b = torch.randn(N, device=DEV)
        y = torch.empty(M, N, device=DEV)
        st = torch.rand(4 * G * K, device=DEV)
        nb = int(lib.cdrl_pwconv_fused_partial_rows(G, Mg, N, K))
        part = torch.zeros(G * nb * 2 * N, dtype=torch.float64, device=DEV)
        t0 = timeit(lambda: lib.cdrl_gemm_nn(P(a), K, 0, P(w), N, 1, P(b), P(y), N, 0, M, N, K, 0, S()))
        t1 = timeit(lambda: lib.cdrl_pwconv_fused(P(a), K, 0, None, P(w), N, 1, P(b), P(y), N, 0, 0, G, Mg, N, K, 0, None, None, None, S()))
        t2 = timeit(lambda: lib.cdrl_pwconv_fused(P(a), K, 0, None, P(w), N, 1, P(b), P(y), N, 0, 0, G, Mg, N, K, 1, None, None, P(part), S()))
        t3 = timeit(lambda: lib.cdrl_pwconv_fused(P(a), K, 0, P(st), P(w), N, 1, P(b), P(y), N, 0, 0, G, Mg, N, K, 1, None, None, P(part), S()))
        print(f'M={M:<8} K={K:<4} N={N:<4} {t0:11.1f}{t1:9.1f}{t2:10.1f}{t3:10.1f}{4.0 * M * (K + N) / 5e6:9.1f}')


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'pw':
    bench_pw()


def bench_tn():
    print(f'{"TN shape":<30}{"us":>9}{"ideal us":>10}')
    for px, K, N in [(660, 24, 58), (165, 58, 92), (165, 24, 24), (165, 58, 58), (48, 116, 116), (12, 232, 232), (12, 464, 768)]:
        M = FRAMES * px
        a = torch.randn(M, K, device=DEV)
        d = torch.randn(M, N, device=DEV)
        dw = torch.empty(K, N, device=DEV)
        ws = torch.empty(int(lib.cdrl_gemm_tn_workspace_elems(M, N, K)), device=DEV)
        t = timeit(lambda: lib.cdrl_gemm_tn(P(a), K, 0, P(d), N, 0, P(dw), M, N, K, P(ws), 0, S()))
        print(f'M={M:<8} K={K:<4} N={N:<4} {t:9.1f}{4.0 * M * (K + N) / 5e6:10.1f}')


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'tn':
    bench_tn()
