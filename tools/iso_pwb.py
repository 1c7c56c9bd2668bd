#!/usr/bin/env python3
"""Isolated timing of the fused pointwise-conv backward (cdrl_pwconv_bwd_fused: kernel + reduce) against the kernels it replaces
(cdrl_pwconv_bn_bwd minus its BN reduce: backward-data GEMM with prologue / epilogue + filter-gradient GEMM), rotating buffer sets."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from carla_driving_rl_agent_amd import _lib

lib = _lib.load()
DEV = 'cuda:0'
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(G, Mg, K, N, shuffle, anorm, nsets=4, iters=24):
    M = G * Mg
    torch.manual_seed(G * 1000003 + Mg + K * 7 + N + shuffle * 2 + anorm)
    ctot, coff = (2 * N, N) if shuffle else (N, 0)
    sets = []
    for _ in range(nsets):
        sets.append(dict(do=torch.randn(M, ctot, device=DEV), y=torch.randn(M, N, device=DEV), x=torch.randn(M, K, device=DEV),
                         da=torch.zeros(M, K, device=DEV)))
    w = torch.randn(K, N, device=DEV) / K ** 0.5
    st = torch.rand(4 * G * N, device=DEV) + 0.5
    cf = torch.rand(3 * G * N, device=DEV) * 0.1
    ast = torch.rand(4 * G * K, device=DEV) + 0.5
    ga, ba = torch.rand(K, device=DEV), torch.rand(K, device=DEV)
    wp = torch.zeros(int(lib.cdrl_pwconv_x3_packed_bytes(N)), dtype=torch.uint8, device=DEV)
    _lib.check(lib.cdrl_pwconv_x3_pack(P(w), N, K, 1, N, P(wp), S()))
    qpart = torch.zeros(int(lib.cdrl_pwconv_bwd_fused_workspace(G, Mg, N, K, 0)), device=DEV)
    dbpart = torch.zeros(int(lib.cdrl_pwconv_bwd_fused_workspace(G, Mg, N, K, 1)), dtype=torch.float64, device=DEV)
    dW, dB = torch.zeros(K, N, device=DEV), torch.zeros(N, device=DEV)
    adg, adb, acf = torch.zeros(K, device=DEV), torch.zeros(K, device=DEV), torch.zeros(3 * G * K, device=DEV)

    def new(k):
        s = sets[k]
        _lib.check(lib.cdrl_pwconv_bwd_fused(P(s['do']), ctot, coff, ctot if shuffle else 0, 1, P(s['y']), P(st), P(cf), P(s['x']), K, 0,
                                             P(ast) if anorm else None, P(ga) if anorm else None, P(ba) if anorm else None,
                                             P(adg) if anorm else None, P(adb) if anorm else None, P(acf) if anorm else None, P(w), P(wp),
                                             P(s['da']), K, 0, 0, P(dW), P(dB), P(qpart), P(dbpart), G, Mg, N, K, S()))

    ws = torch.zeros(int(lib.cdrl_pwconv_bn_bwd_workspace_bytes(G, Mg, N, K)), dtype=torch.uint8, device=DEV)
    dg, dbt, coef = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV), torch.zeros(3 * G * N, device=DEV)

    def old(k):
        s = sets[k]
        _lib.check(lib.cdrl_pwconv_bn_bwd(P(s['do']), ctot, coff, ctot if shuffle else 0, 1, P(s['y']), P(st), P(s['x']), K, 0,
                                          P(ast) if anorm else None, P(w), G, Mg, N, K, P(dg), P(dbt), P(coef), P(s['da']), K, 0, 0,
                                          P(dW), P(dB), P(ws), S()))

    if os.environ.get('PWB_HASH'):      # bit pattern of every output of the fused form (compare two builds / switches)
        import hashlib
        new(0)
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for t in (sets[0]['da'], dW, dB, adg, adb, acf):
            h.update(t.cpu().numpy().tobytes())
        print(f'G={G} Mg={Mg} K={K} N={N} shuffle={shuffle} anorm={anorm}: outputs sha256 {h.hexdigest()[:16]}')
        return
    out = {}
    for name, fn in ((('fused', new),) if os.environ.get('PWB_NEW_ONLY') else (('fused', new), ('old composite (reduce + finalize + bwd-data + filter gradient)', old))):
        for k in range(nsets):
            fn(k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(i % nsets)
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / iters * 1e3
    print(f'G={G} Mg={Mg} K={K} N={N} shuffle={shuffle} anorm={anorm}: ' + ', '.join(f'{k}: {v:.1f} us' for k, v in out.items()))


if __name__ == '__main__' and os.environ.get('PWB_FIXED'):        # fixed cost of a launch against its per-tile cost: 0 / 1 / 2 / 4 / 6 tiles per workgroup
    for mg in (32, 2048, 4096, 8192, 12288):
        run(4, mg, 116, 116, 0, 0)
    for mg in (64, 4096, 8192, 16384, 42240):
        run(4, mg, 58, 58, 0, 0)
    sys.exit(0)
if __name__ == '__main__':
    run(4, 12288, 116, 116, 1, 1)
    run(4, 12288, 116, 116, 0, 0)
    run(4, 42240, 58, 58, 1, 1)
    run(4, 42240, 58, 58, 0, 0)
    if os.environ.get('PWB_HASH'):
        run(4, 12288 - 37, 116, 116, 1, 1)      # ragged last tiles
        run(3, 42240 - 5, 58, 58, 1, 1)
        run(4, 42240, 24, 58, 0, 0)
        run(4, 12288, 58, 116, 0, 0)            # 64 -> 128
