"""Op-level parity: every HIP entry point of include/cdrl.h against a plain PyTorch fp32 / fp64
CPU reference of the same Keras op (tolerance: 1e-4 relative to the tensor scale, the bar
BASELINE.json's north_star states for fp32; most ops land at ~1e-6)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from carla_driving_rl_agent_amd import _lib
from oracle import model as OM
from tests.util import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = 'cuda:0'


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(x):
    return torch.as_tensor(x).to(DEV).contiguous()


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize('M,K,N', [(1000, 58, 58), (777, 116, 116), (300, 24, 58), (513, 232, 232), (130, 464, 768),
                                   (5001, 58, 58), (9000, 116, 116), (4100, 24, 58), (70000, 116, 116), (4097, 57, 116),
                                   (256, 320, 2), (64, 9, 16), (5, 3, 1),
                                   (2500, 232, 232), (2111, 464, 768), (4099, 232, 464), (12288, 232, 232)])   # row-stacked tiles
def test_gemm_nn(lib, M, K, N):
    rng = np.random.default_rng(M + K + N)
    a = rng.standard_normal((M, K)).astype(np.float32)
    b = rng.standard_normal((K, N)).astype(np.float32)       # asymmetric operands catch transposes
    bias = rng.standard_normal(N).astype(np.float32)
    A, B_, bi = dev(a), dev(b), dev(bias)
    out = torch.zeros((M, N), device=DEV)
    _lib.check(lib.cdrl_gemm_nn(P(A), K, 0, P(B_), N, 1, P(bi), P(out), N, 0, M, N, K, 0, S()))
    ref = a.astype(np.float64) @ b.astype(np.float64) + bias
    assert rel_err(out.cpu().numpy(), ref) < 1e-5
    # transposed-B form (backward-data) with accumulate into a channel-slice view
    big = torch.ones((M, K + 7), device=DEV)
    _lib.check(lib.cdrl_gemm_nn(P(out), N, 0, P(B_), 1, N, None, P(big), K + 7, 3, M, K, N, 1, S()))
    ref2 = ref @ b.astype(np.float64).T + 1.0
    got = big.cpu().numpy()
    assert rel_err(got[:, 3:3 + K], ref2) < 1e-5
    assert np.all(got[:, :3] == 1.0) and np.all(got[:, 3 + K:] == 1.0)


@pytest.mark.parametrize('M,K,N', [(5000, 58, 58), (3000, 116, 116), (2049, 24, 58), (1500, 232, 232), (700, 464, 768),
                                   (256, 512, 320), (33, 16, 96), (40000, 16, 96), (30000, 32, 96), (50000, 24, 92), (20000, 96, 24),
                                   (12288, 232, 232), (3000, 464, 768)])
def test_gemm_tn(lib, M, K, N):
    rng = np.random.default_rng(M + K + N)
    a = rng.standard_normal((M, K + 5)).astype(np.float32)
    d = rng.standard_normal((M, N)).astype(np.float32)
    A, D = dev(a), dev(d)
    ws = torch.zeros(int(lib.cdrl_gemm_tn_workspace_elems(M, N, K)), device=DEV)
    out = torch.zeros((K, N), device=DEV)
    _lib.check(lib.cdrl_gemm_tn(P(A), K + 5, 2, P(D), N, 0, P(out), M, N, K, P(ws), 0, S()))
    ref = a[:, 2:2 + K].astype(np.float64).T @ d.astype(np.float64)
    assert rel_err(out.cpu().numpy(), ref) < 1e-5


@pytest.mark.parametrize('mode', ['1', '2'])
def test_gemm_tn_lds_float32_forms(mode):
    """The opt-in LDS-staged filter gradients for float32 tensors (CDRL_TN_LDS_F32 = 1: float32 MFMA from LDS; 2: three bf16 planes
    per operand, six plane products on the bf16 pipe) keep float32 accuracy -- plain product and the fused BN-backward op.  The
    switch is read once per process: child process."""
    import subprocess
    import sys
    env = dict(os.environ, CDRL_TN_LDS_F32=mode)
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-k',
                        'test_gemm_tn and (116 or 232 or 464) or test_pwconv_bn_bwd'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout, r.stdout[-500:]


@pytest.mark.parametrize('G,Mg,K,N,pro,epi,bt', [(4, 700, 58, 58, 0, 1, 0), (4, 333, 116, 116, 1, 1, 0), (2, 1000, 24, 58, 0, 1, 0),
                                                (4, 130, 58, 92, 1, 1, 0), (4, 257, 24, 24, 1, 1, 0), (1, 5000, 116, 116, 0, 0, 0),
                                                (4, 513, 116, 116, 0, 2, 1), (4, 300, 92, 58, 0, 2, 1), (4, 200, 58, 24, 0, 0, 1),
                                                (4, 20000, 58, 58, 1, 1, 0), (4, 64, 116, 58, 0, 2, 1), (4, 300, 232, 232, 1, 1, 0),
                                                (4, 200, 232, 232, 0, 2, 1), (2, 150, 116, 232, 0, 1, 0)])
def test_pwconv_fused(lib, G, Mg, K, N, pro, epi, bt):
    """Persistent skinny GEMM with BN-apply prologue and statistics / BN-backward-sum epilogues."""
    rng = np.random.default_rng(G * Mg + K + N)
    M = G * Mg
    lda, coff = K + 6, 2
    a = rng.standard_normal((M, lda)).astype(np.float32)
    w = rng.standard_normal((N, K) if bt else (K, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    pst = rng.uniform(0.5, 1.5, (4, G, K)).astype(np.float32)
    est = rng.uniform(0.5, 1.5, (4, G, N)).astype(np.float32)
    ey = rng.standard_normal((M, N)).astype(np.float32)
    a64 = a[:, coff:coff + K].astype(np.float64).reshape(G, Mg, K)
    if pro:
        a64 = a64 * pst[2][:, None, :] + pst[3][:, None, :]
    w64 = w.astype(np.float64).T if bt else w.astype(np.float64)
    ref = (a64 @ w64 + (0.0 if bt else bias)).reshape(M, N)
    nb = int(lib.cdrl_pwconv_fused_partial_rows(G, Mg, N, K))
    A, Wd, Bd, PS, ES, EY = dev(a), dev(w), dev(bias), dev(pst), dev(est), dev(ey)
    out = torch.full((M, N + 3), 1.0, device=DEV)
    part = torch.zeros((G, nb, 2, N), dtype=torch.float64, device=DEV)
    sbk, sbn = (1, K) if bt else (N, 1)
    _lib.check(lib.cdrl_pwconv_fused(P(A), lda, coff, P(PS) if pro else None, P(Wd), sbk, sbn, None if bt else P(Bd), P(out), N + 3, 1,
                                     1 if bt else 0, G, Mg, N, K, epi, P(EY), P(ES), P(part), S()))
    got = out.cpu().numpy()
    assert rel_err(got[:, 1:1 + N], ref + (1.0 if bt else 0.0)) < 1e-5
    assert np.all(got[:, 0] == 1.0) and np.all(got[:, N + 1:] == 1.0)
    if epi:
        ps = part.sum(dim=1).cpu().numpy()              # (G, 2, N)
        r3 = ref.reshape(G, Mg, N)
        assert rel_err(ps[:, 0], r3.sum(axis=1)) < 1e-5
        if epi == 1:
            assert rel_err(ps[:, 1], (r3 * r3).sum(axis=1)) < 1e-5
        else:
            xh = (ey.astype(np.float64).reshape(G, Mg, N) - est[0][:, None, :]) * est[1][:, None, :]
            assert rel_err(ps[:, 1], (r3 * xh).sum(axis=1)) < 1e-5


@pytest.mark.parametrize('G,Mg,K,N,relu,shuffle,xpro', [(4, 330, 58, 58, 1, 1, 0), (4, 96, 116, 116, 1, 1, 1), (2, 500, 24, 58, 1, 0, 0),
                                                        (4, 257, 58, 24, 0, 0, 1), (4, 1500, 116, 116, 1, 1, 1), (1, 64, 58, 116, 0, 0, 0),
                                                        (4, 120, 232, 232, 1, 1, 1), (4, 3072, 232, 232, 1, 1, 1), (2, 777, 232, 232, 0, 0, 0)])
def test_pwconv_bn_bwd(lib, G, Mg, K, N, relu, shuffle, xpro):
    """Backward of conv1x1 -> BN(train, per time slice) (+ReLU6) (+shuffled store) with the BN-backward apply fused
    into the operand loads of the backward-data and filter-gradient GEMMs: against torch autograd."""
    rng = np.random.default_rng(G * Mg + K + N)
    M = G * Mg
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    xst = np.stack([np.zeros((G, K)), np.ones((G, K)), rng.uniform(0.5, 1.5, (G, K)), rng.uniform(-0.5, 0.5, (G, K))]).astype(np.float32)
    p = {'b.gamma': torch.tensor(rng.uniform(0.5, 1.5, N), dtype=torch.float64).requires_grad_(True),
         'b.beta': torch.tensor(rng.uniform(1.0, 3.0, N), dtype=torch.float64).requires_grad_(True),
         'b.moving_mean': torch.zeros(N, dtype=torch.float64), 'b.moving_var': torch.ones(N, dtype=torch.float64)}
    xt = torch.tensor(x, dtype=torch.float64).reshape(G, Mg, K).requires_grad_(True)
    wt = torch.tensor(w, dtype=torch.float64).requires_grad_(True)
    bt = torch.tensor(bias, dtype=torch.float64).requires_grad_(True)
    xin = xt * torch.tensor(xst[2], dtype=torch.float64)[:, None, :] + torch.tensor(xst[3], dtype=torch.float64)[:, None, :] if xpro else xt
    yt = xin @ wt + bt                                                   # (G, Mg, N)
    out = OM.bn_slices(yt.permute(0, 2, 1)[:, None], p, 'b', True, True)   # (G,1,N,Mg)
    if relu:
        out = OM.relu6(out)
    ctot = 2 * N if shuffle else N
    coff = N if shuffle else 0
    dout = rng.standard_normal((M, ctot)).astype(np.float32)
    idx = [((coff + c) & 1) * (ctot // 2) + ((coff + c) >> 1) for c in range(N)] if shuffle else list(range(N))
    out.backward(torch.tensor(dout[:, idx], dtype=torch.float64).reshape(G, Mg, N).permute(0, 2, 1)[:, None])
    # GPU: forward statistics through the fused forward GEMM + finalize (cdrl_bn_train_fwd on y)
    X, Wd, Bd, XS, DO = dev(x), dev(w), dev(bias), dev(xst), dev(dout)
    y = dev(yt.detach().reshape(M, N).float().numpy())
    stats = torch.zeros(4 * G * N, device=DEV)
    tmp = torch.zeros((M, N), device=DEV)
    ws0 = torch.zeros(G * 256 * 2 * N, dtype=torch.float64, device=DEV)
    gam, bet = dev(p['b.gamma'].detach().float()), dev(p['b.beta'].detach().float())
    mm, mv = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
    _lib.check(lib.cdrl_bn_train_fwd(P(y), G, Mg, N, P(gam), P(bet), P(mm), P(mv), 1, relu, P(tmp), N, 0, 0, P(stats), P(ws0), S()))
    ws = torch.zeros(int(lib.cdrl_pwconv_bn_bwd_workspace_bytes(G, Mg, N, K)), dtype=torch.uint8, device=DEV)
    dg, dbt, coef = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV), torch.zeros(3 * G * N, device=DEV)
    dx = torch.full((M, K + 4), 2.0, device=DEV)
    dw, db = torch.zeros((K, N), device=DEV), torch.zeros(N, device=DEV)
    _lib.check(lib.cdrl_pwconv_bn_bwd(P(DO), ctot, coff, ctot if shuffle else 0, relu, P(y), P(stats), P(X), K, 0, P(XS) if xpro else None,
                                      P(Wd), G, Mg, N, K, P(dg), P(dbt), P(coef), P(dx), K + 4, 2, 1, P(dw), P(db), P(ws), S()))
    assert rel_err(dg.cpu().numpy(), p['b.gamma'].grad.numpy()) < 2e-5
    assert rel_err(dbt.cpu().numpy(), p['b.beta'].grad.numpy()) < 2e-5
    # dx is the gradient w.r.t. the conv INPUT after the optional affine: d(xin) = d(x) / scale
    got = dx.cpu().numpy()
    ref_dxin = (xt.grad.reshape(M, K).numpy() / xst[2].astype(np.float64).repeat(Mg, axis=0)) if xpro else xt.grad.reshape(M, K).numpy()
    assert rel_err(got[:, 2:2 + K] - 2.0, ref_dxin) < 3e-5
    assert np.all(got[:, :2] == 2.0) and np.all(got[:, 2 + K:] == 2.0)
    assert rel_err(dw.cpu().numpy(), wt.grad.numpy()) < 3e-5
    assert np.abs(db.cpu().numpy()).max() < 1e-4 * np.abs(dw.cpu().numpy()).max()


@pytest.mark.parametrize('B,T,H,W', [(2, 4, 41, 58), (3, 2, 90, 120)])
def test_stem(lib, B, T, H, W):
    rng = np.random.default_rng(1)
    x = rng.random((B, T, H, W, 3), dtype=np.float32)
    w = rng.standard_normal((3, 3, 3, 24)).astype(np.float32) * 0.2
    b = rng.standard_normal(24).astype(np.float32)
    Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    y = torch.zeros((T * B, Ho, Wo, 24), device=DEV)
    X, Wd, Bd = dev(x), dev(w), dev(b)
    _lib.check(lib.cdrl_stem_fwd(P(X), P(Wd), P(Bd), P(y), B, T, H, W, 24, S()))
    xt = torch.tensor(x, dtype=torch.float64).permute(1, 0, 4, 2, 3).reshape(T * B, 3, H, W).requires_grad_(False)
    wt = torch.tensor(w, dtype=torch.float64).permute(3, 2, 0, 1).requires_grad_(True)
    bt = torch.tensor(b, dtype=torch.float64).requires_grad_(True)
    ref = F.conv2d(xt, wt, bt, stride=2)
    assert rel_err(y.cpu().numpy(), ref.detach().permute(0, 2, 3, 1).numpy()) < 1e-5
    dy = rng.standard_normal((T * B, Ho, Wo, 24)).astype(np.float32)
    ref.backward(torch.tensor(dy, dtype=torch.float64).permute(0, 3, 1, 2))
    ws = torch.zeros(int(lib.cdrl_stem_bwd_workspace_doubles(B, T, H, W, 24)), dtype=torch.float64, device=DEV)
    dw = torch.zeros((3, 3, 3, 24), device=DEV)
    db = torch.zeros(24, device=DEV)
    DY = dev(dy)
    _lib.check(lib.cdrl_stem_bwd_filter(P(X), P(DY), P(dw), P(db), B, T, H, W, 24, P(ws), S()))
    assert rel_err(dw.cpu().numpy(), wt.grad.permute(2, 3, 1, 0).numpy()) < 1e-5
    assert rel_err(db.cpu().numpy(), bt.grad.numpy()) < 1e-5


@pytest.mark.parametrize('B,T,H,W,Cc', [(3, 4, 90, 120, 24), (2, 2, 31, 33, 24), (5, 1, 21, 28, 24), (2, 3, 41, 58, 24), (1, 4, 90, 360, 24),
                                        (2, 2, 17, 19, 32), (3, 1, 9, 7, 8)])
def test_stem_fwd_with_statistics(lib, B, T, H, W, Cc):
    """Stem conv + (sum, sum of squares) of its output per time slice in one kernel (round 5: image band staged in LDS): the conv output is
    BIT-identical to the plain conv kernel (same fmaf chain per element), the statistics agree with float64 sums of that output to 1e-12."""
    rng = np.random.default_rng(B * H + W + Cc)
    x = rng.uniform(0.0, 1.0, (B, T, H, W, 3)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, Cc)) * 0.4).astype(np.float32)
    b = rng.standard_normal(Cc).astype(np.float32)
    Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    X, Wd, Bd = dev(x), dev(w), dev(b)
    y0 = torch.zeros((T * B, Ho, Wo, Cc), device=DEV)
    _lib.check(lib.cdrl_stem_fwd(P(X), P(Wd), P(Bd), P(y0), B, T, H, W, Cc, S()))
    rows = int(lib.cdrl_stem_fwd_stats_rows(B, T, H, W, Cc))
    y1 = torch.full((T * B, Ho, Wo, Cc), 7.0, device=DEV)
    part = torch.full((T, rows, 2, Cc), float('nan'), dtype=torch.float64, device=DEV)
    _lib.check(lib.cdrl_stem_fwd_stats(P(X), P(Wd), P(Bd), P(y1), P(part), B, T, H, W, Cc, S()))
    assert torch.equal(y1, y0)
    yd = y0.double().view(T, B * Ho * Wo, Cc)
    s, q = part[:, :, 0].sum(1), part[:, :, 1].sum(1)
    assert not bool(torch.isnan(part).any())
    assert float(((s - yd.sum(1)).abs() / yd.abs().sum(1)).max()) < 1e-12
    assert float(((q - (yd * yd).sum(1)).abs() / (yd * yd).sum(1)).max()) < 1e-12


@pytest.mark.parametrize('N,H,W,Cc,stride', [(3, 22, 30, 58, 2), (2, 11, 15, 58, 1), (2, 6, 8, 116, 1), (3, 11, 15, 116, 2),
                                             (2, 3, 4, 232, 1), (2, 6, 23, 232, 2), (2, 5, 6, 24, 2)])
def test_dwconv(lib, N, H, W, Cc, stride):
    rng = np.random.default_rng(N * H + W + Cc)
    a = rng.standard_normal((N, H, W, Cc)).astype(np.float32)
    w = rng.standard_normal((3, 3, Cc, 1)).astype(np.float32)
    b = rng.standard_normal(Cc).astype(np.float32)
    Ho, Wo = -(-H // stride), -(-W // stride)
    at = torch.tensor(a, dtype=torch.float64).permute(0, 3, 1, 2)[None].requires_grad_(True)
    p = {'c.w': torch.tensor(w, dtype=torch.float64).requires_grad_(True),
         'c.b': torch.tensor(b, dtype=torch.float64).requires_grad_(True)}
    ref = OM.conv_dw(at, p, 'c', stride)[0]
    Ad, Wd, Bd = dev(a), dev(w), dev(b)
    y = torch.zeros((N, Ho, Wo, Cc), device=DEV)
    _lib.check(lib.cdrl_dwconv_fwd(P(Ad), P(Wd), P(Bd), P(y), N, H, W, Cc, stride, S()))
    assert tuple(ref.shape) == (N, Cc, Ho, Wo)
    assert rel_err(y.cpu().numpy(), ref.detach().permute(0, 2, 3, 1).numpy()) < 1e-5
    dy = rng.standard_normal((N, Ho, Wo, Cc)).astype(np.float32)
    ref.backward(torch.tensor(dy, dtype=torch.float64).permute(0, 3, 1, 2))
    DY = dev(dy)
    da = torch.zeros((N, H, W, Cc), device=DEV)
    _lib.check(lib.cdrl_dwconv_bwd_data(P(DY), P(Wd), P(da), N, H, W, Cc, stride, S()))
    assert rel_err(da.cpu().numpy(), at.grad[0].permute(0, 2, 3, 1).numpy()) < 1e-5
    ws = torch.zeros(int(lib.cdrl_dwconv_bwd_workspace_doubles(N, H, W, Cc, stride)), dtype=torch.float64, device=DEV)
    dw = torch.zeros((3, 3, Cc, 1), device=DEV)
    db = torch.zeros(Cc, device=DEV)
    _lib.check(lib.cdrl_dwconv_bwd_filter(P(Ad), P(DY), P(dw), P(db), N, H, W, Cc, stride, P(ws), S()))
    assert rel_err(dw.cpu().numpy(), p['c.w'].grad.numpy()) < 1e-5
    assert rel_err(db.cpu().numpy(), p['c.b'].grad.numpy()) < 1e-5


@pytest.mark.parametrize('T,B,H,W,Cc,stride,pre', [(4, 2, 11, 15, 58, 1, 1), (2, 4, 22, 30, 58, 2, 1), (4, 8, 3, 4, 232, 1, 1),
                                                  (2, 3, 6, 8, 116, 1, 1), (2, 2, 22, 30, 24, 2, 0), (4, 2, 11, 15, 116, 2, 0),
                                                  (2, 2, 6, 23, 232, 2, 1), (1, 2, 22, 90, 58, 2, 1), (2, 3, 6, 8, 116, 1, 2), (2, 4, 22, 30, 58, 2, 2)])
def test_dwconv_bn_fused(lib, T, B, H, W, Cc, stride, pre):
    """Fused depthwise block (frames in LDS): [BN+ReLU6 prologue] -> dw3x3 -> following BN statistics, and its
    backward (BN-backward prologue, filter/bias gradients, input gradient, ReLU6 mask, previous BN backward):
    against torch autograd of the unfused composition."""
    rng = np.random.default_rng(T * B + H * W + Cc + stride)
    N = T * B
    x = (rng.standard_normal((T, B, H, W, Cc)) * 1.5 + 0.4).astype(np.float32)
    w = rng.standard_normal((3, 3, Cc, 1)).astype(np.float32)
    b = rng.standard_normal(Cc).astype(np.float32)
    Ho, Wo = -(-H // stride), -(-W // stride)

    def bnp(name):
        return {f'{name}.gamma': torch.tensor(rng.uniform(0.5, 1.5, Cc), dtype=torch.float64).requires_grad_(True),
                f'{name}.beta': torch.tensor(rng.uniform(1.0, 3.0, Cc), dtype=torch.float64).requires_grad_(True),
                f'{name}.moving_mean': torch.tensor(rng.uniform(-0.2, 0.2, Cc), dtype=torch.float64),
                f'{name}.moving_var': torch.tensor(rng.uniform(0.5, 1.5, Cc), dtype=torch.float64)}
    p = {'c.w': torch.tensor(w, dtype=torch.float64).requires_grad_(True),
         'c.b': torch.tensor(b, dtype=torch.float64).requires_grad_(True), **bnp('pre'), **bnp('post')}
    if pre == 2:
        # channels the backward must NOT take xhat1 = (a - beta) / gamma for (tiny |gamma|, large |beta| / |gamma|: it re-reads y1 there),
        # mixed with ordinary ones inside a thread's channel pair
        with torch.no_grad():
            p['pre.gamma'][::5] = torch.tensor(rng.choice([-1.0, 1.0], len(p['pre.gamma'][::5])) * 0.01, dtype=torch.float64)
            p['pre.beta'][::5] = torch.tensor(rng.uniform(0.5, 2.5, len(p['pre.beta'][::5])), dtype=torch.float64)
            p['pre.gamma'][3::7] = 0.2
            p['pre.beta'][3::7] = 3.0 + 0.0 * p['pre.beta'][3::7]
    f32 = {k: v.detach().clone().float() for k, v in p.items()}
    xt = torch.tensor(x, dtype=torch.float64).permute(0, 1, 4, 2, 3).requires_grad_(True)        # (T,B,C,H,W)
    a = OM.relu6(OM.bn_slices(xt, p, 'pre', True, True)) if pre else xt
    y2 = OM.conv_dw(a, p, 'c', stride)
    out = OM.bn_slices(y2, p, 'post', True, True)
    dout = rng.standard_normal((T, B, Ho, Wo, Cc)).astype(np.float32)
    out.backward(torch.tensor(dout, dtype=torch.float64).permute(0, 1, 4, 2, 3))

    X, Wd, Bd, DO = dev(x), dev(w), dev(b), dev(dout)
    pre_stats = None
    if pre:
        pre_stats = torch.zeros(4 * T * Cc, device=DEV)
        tmp = torch.zeros((N * H * W, Cc), device=DEV)
        ws0 = torch.zeros(T * 256 * 2 * Cc, dtype=torch.float64, device=DEV)
        mm, mv = dev(f32['pre.moving_mean']), dev(f32['pre.moving_var'])
        g1, b1 = dev(f32['pre.gamma']), dev(f32['pre.beta'])          # keep alive: P() only passes the address
        _lib.check(lib.cdrl_bn_train_fwd(P(X), T, B * H * W, Cc, P(g1), P(b1), P(mm), P(mv),
                                         1, 1, P(tmp), Cc, 0, 0, P(pre_stats), P(ws0), S()))
    ws = torch.zeros(int(lib.cdrl_dwconv_bn_workspace_doubles(T, B, H, W, Cc, stride)), dtype=torch.float64, device=DEV)
    y = torch.zeros((N, Ho, Wo, Cc), device=DEV)
    post_stats = torch.zeros(4 * T * Cc, device=DEV)
    g2, b2 = dev(f32['post.gamma']), dev(f32['post.beta'])
    mm2, mv2 = dev(f32['post.moving_mean']), dev(f32['post.moving_var'])
    _lib.check(lib.cdrl_dwconv_bn_fwd(P(X), P(pre_stats), P(Wd), P(Bd), P(y), T, B, H, W, Cc, stride, P(g2), P(b2), P(mm2), P(mv2),
                                      1, P(post_stats), P(ws), S()))
    y2n = y2.detach().permute(0, 1, 3, 4, 2).reshape(N, Ho, Wo, Cc).numpy()
    assert rel_err(y.cpu().numpy(), y2n) < 1e-5
    st = post_stats.cpu().numpy().reshape(4, T, Cc)
    y2g = y2n.reshape(T, -1, Cc)
    assert rel_err(st[0], y2g.mean(axis=1)) < 1e-5
    assert rel_err(st[1], 1.0 / np.sqrt(y2g.var(axis=1) + 1e-3)) < 1e-5
    assert rel_err(mm2.cpu().numpy(), p['post.moving_mean'].numpy()) < 1e-5
    assert rel_err(mv2.cpu().numpy(), p['post.moving_var'].numpy()) < 1e-5
    # backward
    dx = torch.zeros((N, H, W, Cc), device=DEV)
    dw = torch.zeros((3, 3, Cc, 1), device=DEV)
    db = torch.zeros(Cc, device=DEV)
    vecs = [torch.zeros(Cc, device=DEV) for _ in range(4)]
    coefs = [torch.zeros(3 * T * Cc, device=DEV) for _ in range(2)]
    _lib.check(lib.cdrl_dwconv_bn_bwd(P(X), P(pre_stats), P(DO), P(y), P(post_stats), P(Wd), T, B, H, W, Cc, stride, P(dx), P(dw),
                                      P(db), P(vecs[0]), P(vecs[1]), P(coefs[0]), P(vecs[2]), P(vecs[3]), P(coefs[1]), P(ws), S()))
    assert rel_err(dx.cpu().numpy(), xt.grad.permute(0, 1, 3, 4, 2).reshape(N, H, W, Cc).numpy()) < 2e-5
    assert rel_err(dw.cpu().numpy(), p['c.w'].grad.numpy()) < 2e-5
    # the bias feeds a train-mode BN: its true gradient is 0, the computed one is rounding noise of the sums
    assert np.abs(db.cpu().numpy()).max() < 1e-4 * np.abs(dw.cpu().numpy()).max()
    assert rel_err(vecs[0].cpu().numpy(), p['post.gamma'].grad.numpy()) < 2e-5
    assert rel_err(vecs[1].cpu().numpy(), p['post.beta'].grad.numpy()) < 2e-5
    if pre:
        eg = np.abs(vecs[2].cpu().numpy() - p['pre.gamma'].grad.numpy()) / np.abs(p['pre.gamma'].grad.numpy()).max()
        assert eg.max() < 2e-5, (eg.max(), int(eg.argmax()), p['pre.gamma'][int(eg.argmax())].item(), p['pre.beta'][int(eg.argmax())].item())
        # (pre == 2: 2.4e-5 measured on the stride-2 case, with and without the xhat shortcut -- the tiny-gamma channels put 1 / gamma = 100 in front of the float32 dz)
        assert rel_err(vecs[3].cpu().numpy(), p['pre.beta'].grad.numpy()) < (5e-5 if pre == 2 else 2e-5)


@pytest.mark.parametrize('N,H,W', [(3, 44, 59), (2, 19, 27), (2, 20, 28)])
def test_maxpool(lib, N, H, W):
    rng = np.random.default_rng(H)
    a = rng.standard_normal((N, H, W, 24)).astype(np.float32)
    Ho, Wo = -(-H // 2), -(-W // 2)
    at = torch.tensor(a, dtype=torch.float64).permute(0, 3, 1, 2).requires_grad_(True)
    ph, pw = OM.same_pad(H, 3, 2), OM.same_pad(W, 3, 2)
    ref = F.max_pool2d(F.pad(at, (pw[0], pw[1], ph[0], ph[1]), value=float('-inf')), 3, 2)
    Ad = dev(a)
    p = torch.zeros((N, Ho, Wo, 24), device=DEV)
    am = torch.zeros((N, Ho, Wo, 24), dtype=torch.uint8, device=DEV)
    _lib.check(lib.cdrl_maxpool_fwd(P(Ad), P(p), P(am), N, H, W, 24, S()))
    assert np.array_equal(p.cpu().numpy(), ref.detach().permute(0, 2, 3, 1).numpy().astype(np.float32))
    dp = rng.standard_normal((N, Ho, Wo, 24)).astype(np.float32)
    ref.backward(torch.tensor(dp, dtype=torch.float64).permute(0, 3, 1, 2))
    da = torch.zeros((N, H, W, 24), device=DEV)
    DP = dev(dp)
    _lib.check(lib.cdrl_maxpool_bwd(P(am), P(DP), P(da), N, H, W, 24, S()))
    assert rel_err(da.cpu().numpy(), at.grad.permute(0, 2, 3, 1).numpy()) < 1e-6


@pytest.mark.parametrize('G,Mg,Cc,relu,shuffle', [(4, 330, 58, 1, 0), (4, 96, 116, 0, 0), (4, 50, 92, 1, 1), (1, 64, 352, 0, 0),
                                                  (4, 24, 768, 1, 0), (4, 1300, 24, 1, 0)])
def test_bn_train(lib, G, Mg, Cc, relu, shuffle):
    """Per-time-slice training BatchNorm (+ReLU6, + de-interleave shuffle on the store), fwd + bwd,
    incl. the T sequential moving-stat EMA updates with Bessel-corrected variance."""
    rng = np.random.default_rng(G * Mg + Cc)
    y = (rng.standard_normal((G, Mg, Cc)) * 2.0 + 0.7).astype(np.float32)
    gamma = rng.uniform(0.5, 1.5, Cc).astype(np.float32)
    beta = rng.uniform(-0.5, 0.5, Cc).astype(np.float32)
    mm0 = rng.uniform(-0.2, 0.2, Cc).astype(np.float32)
    mv0 = rng.uniform(0.5, 1.5, Cc).astype(np.float32)
    p = {'b.gamma': torch.tensor(gamma, dtype=torch.float64).requires_grad_(True),
         'b.beta': torch.tensor(beta, dtype=torch.float64).requires_grad_(True),
         'b.moving_mean': torch.tensor(mm0, dtype=torch.float64), 'b.moving_var': torch.tensor(mv0, dtype=torch.float64)}
    yt = torch.tensor(y, dtype=torch.float64).permute(0, 2, 1)[:, None].requires_grad_(True)     # (G,1,C,Mg)
    ref = OM.bn_slices(yt, p, 'b', True, True)
    if relu:
        ref = OM.relu6(ref)
    ctot = 2 * Cc if shuffle else Cc
    coff = Cc if shuffle else 0          # main-branch half of a concat buffer
    out = torch.zeros((G * Mg, ctot), device=DEV)
    stats = torch.zeros(4 * G * Cc, device=DEV)
    ws = torch.zeros(G * 256 * 2 * Cc, dtype=torch.float64, device=DEV)
    Y, Gm, Bt, MM, MV = dev(y), dev(gamma), dev(beta), dev(mm0), dev(mv0)
    _lib.check(lib.cdrl_bn_train_fwd(P(Y), G, Mg, Cc, P(Gm), P(Bt), P(MM), P(MV), 1, relu, P(out), ctot, coff,
                                     ctot if shuffle else 0, P(stats), P(ws), S()))
    refn = ref.detach()[:, 0].permute(0, 2, 1).reshape(G * Mg, Cc).numpy()
    got = out.cpu().numpy()
    if shuffle:
        idx = [((coff + c) & 1) * (ctot // 2) + ((coff + c) >> 1) for c in range(Cc)]
        got = got[:, idx]
    assert rel_err(got, refn) < 1e-5
    assert rel_err(MM.cpu().numpy(), p['b.moving_mean'].numpy()) < 1e-5
    assert rel_err(MV.cpu().numpy(), p['b.moving_var'].numpy()) < 1e-5
    # backward
    dout = rng.standard_normal((G * Mg, ctot)).astype(np.float32)
    dsel = dout[:, idx] if shuffle else dout
    ref.backward(torch.tensor(dsel, dtype=torch.float64).reshape(G, Mg, Cc).permute(0, 2, 1)[:, None])
    DO = dev(dout)
    dg = torch.zeros(Cc, device=DEV)
    dbt = torch.zeros(Cc, device=DEV)
    dy = torch.zeros((G * Mg, Cc), device=DEV)
    coef = torch.zeros(3 * G * Cc, device=DEV)
    _lib.check(lib.cdrl_bn_train_bwd(P(DO), ctot, coff, ctot if shuffle else 0, P(Y), G, Mg, Cc, P(stats), relu, P(dg),
                                     P(dbt), P(dy), P(coef), P(ws), S()))
    assert rel_err(dg.cpu().numpy(), p['b.gamma'].grad.numpy()) < 1e-5
    assert rel_err(dbt.cpu().numpy(), p['b.beta'].grad.numpy()) < 1e-5
    assert rel_err(dy.cpu().numpy(), yt.grad[:, 0].permute(0, 2, 1).reshape(G * Mg, Cc).numpy()) < 2e-5


@pytest.mark.parametrize('B,T,H,W', [(3, 4, 21, 28), (2, 2, 44, 59)])
def test_fused_stem_block(lib, B, T, H, W):
    """BN(train, per time slice) + ReLU6 + max-pool fused forward, and the BN backward that pulls its
    gradient through the pool argmax: against torch autograd of the unfused composition."""
    Cc, N = 24, B * T
    rng = np.random.default_rng(H * W)
    y = (rng.standard_normal((T, B, H, W, Cc)) * 1.5 + 0.3).astype(np.float32)
    gamma = rng.uniform(0.5, 1.5, Cc).astype(np.float32)
    gamma[::5] *= -1.0                                    # negative scale: max must be taken AFTER the affine
    beta = rng.uniform(-0.5, 0.5, Cc).astype(np.float32)
    Ho, Wo = -(-H // 2), -(-W // 2)
    p = {'b.gamma': torch.tensor(gamma, dtype=torch.float64).requires_grad_(True),
         'b.beta': torch.tensor(beta, dtype=torch.float64).requires_grad_(True),
         'b.moving_mean': torch.zeros(Cc, dtype=torch.float64), 'b.moving_var': torch.ones(Cc, dtype=torch.float64)}
    yt = torch.tensor(y, dtype=torch.float64).permute(0, 1, 4, 2, 3).requires_grad_(True)     # (T,B,C,H,W)
    a = OM.relu6(OM.bn_slices(yt, p, 'b', True, True))
    ph, pw = OM.same_pad(H, 3, 2), OM.same_pad(W, 3, 2)
    ref = F.max_pool2d(F.pad(a.reshape(N, Cc, H, W), (pw[0], pw[1], ph[0], ph[1]), value=float('-inf')), 3, 2)
    G, Mg = T, B * H * W
    Y, Gm, Bt = dev(y.reshape(N, H, W, Cc)), dev(gamma), dev(beta)
    MM, MV = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
    stats = torch.zeros(4 * G * Cc, device=DEV)
    ws = torch.zeros(G * 256 * 2 * Cc, dtype=torch.float64, device=DEV)
    scratch = torch.zeros((N * H * W, Cc), device=DEV)
    _lib.check(lib.cdrl_bn_train_fwd(P(Y), G, Mg, Cc, P(Gm), P(Bt), P(MM), P(MV), 1, 1, P(scratch), Cc, 0, 0, P(stats), P(ws), S()))
    pool = torch.zeros((N, Ho, Wo, Cc), device=DEV)
    am = torch.zeros((N, Ho, Wo, Cc), dtype=torch.uint8, device=DEV)
    _lib.check(lib.cdrl_maxpool_bn_fwd(P(Y), P(stats), G, B, P(pool), P(am), N, H, W, Cc, S()))
    assert rel_err(pool.cpu().numpy(), ref.detach().permute(0, 2, 3, 1).numpy()) < 1e-5
    dp = rng.standard_normal((N, Ho, Wo, Cc)).astype(np.float32)
    ref.backward(torch.tensor(dp, dtype=torch.float64).permute(0, 3, 1, 2))
    DP = dev(dp)
    dg, dbt = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV)
    dy = torch.zeros((N * H * W, Cc), device=DEV)
    coef = torch.zeros(3 * G * Cc, device=DEV)
    _lib.check(lib.cdrl_bn_train_bwd_pooled(P(am), P(DP), H, W, P(Y), G, Mg, Cc, P(stats), P(dg), P(dbt), P(dy), P(coef), P(ws), S()))
    assert rel_err(dg.cpu().numpy(), p['b.gamma'].grad.numpy()) < 1e-5
    assert rel_err(dbt.cpu().numpy(), p['b.beta'].grad.numpy()) < 1e-5
    assert rel_err(dy.cpu().numpy(), yt.grad.permute(0, 1, 3, 4, 2).reshape(N * H * W, Cc).numpy()) < 2e-5


@pytest.mark.parametrize('B,T,H,W', [(3, 4, 41, 58), (2, 2, 90, 120), (5, 1, 31, 33)])
def test_stem_block_bwd(lib, B, T, H, W):
    """Backward of conv3x3/s2 -> BN+ReLU6 -> max-pool from the pooled gradient only (scatter-form BN sums, BN-backward
    apply + pool gather fused into the filter-gradient GEMM): against torch autograd of the unfused composition."""
    Cc, N = 24, B * T
    rng = np.random.default_rng(B * H + W)
    x = rng.uniform(0.0, 1.0, (B, T, H, W, 3)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, Cc)) * 0.4).astype(np.float32)
    b = rng.standard_normal(Cc).astype(np.float32)
    gamma = rng.uniform(0.5, 1.5, Cc).astype(np.float32)
    gamma[::5] *= -1.0
    beta = rng.uniform(0.5, 2.5, Cc).astype(np.float32)
    Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    Hp, Wp = -(-Ho // 2), -(-Wo // 2)
    p = {'b.gamma': torch.tensor(gamma, dtype=torch.float64).requires_grad_(True),
         'b.beta': torch.tensor(beta, dtype=torch.float64).requires_grad_(True),
         'b.moving_mean': torch.zeros(Cc, dtype=torch.float64), 'b.moving_var': torch.ones(Cc, dtype=torch.float64)}
    wt = torch.tensor(w, dtype=torch.float64).requires_grad_(True)
    bt = torch.tensor(b, dtype=torch.float64).requires_grad_(True)
    xt = torch.tensor(x, dtype=torch.float64).permute(1, 0, 4, 2, 3).reshape(N, 3, H, W)          # frames f = t*B + b
    yt = F.conv2d(xt, wt.permute(3, 2, 0, 1), bt, stride=2)                                         # (N,Cc,Ho,Wo)
    a = OM.relu6(OM.bn_slices(yt.reshape(T, B, Cc, Ho, Wo), p, 'b', True, True))
    ph, pw = OM.same_pad(Ho, 3, 2), OM.same_pad(Wo, 3, 2)
    ref = F.max_pool2d(F.pad(a.reshape(N, Cc, Ho, Wo), (pw[0], pw[1], ph[0], ph[1]), value=float('-inf')), 3, 2)
    dp = rng.standard_normal((N, Hp, Wp, Cc)).astype(np.float32)
    ref.backward(torch.tensor(dp, dtype=torch.float64).permute(0, 3, 1, 2))
    # GPU forward pieces (stem conv, statistics, fused pool) to get y / stats / argmax, then the fused backward
    X, Wd, Bd, Gm, Bt, DP = dev(x), dev(w), dev(b), dev(gamma), dev(beta), dev(dp)
    y = torch.zeros((N, Ho, Wo, Cc), device=DEV)
    _lib.check(lib.cdrl_stem_fwd(P(X), P(Wd), P(Bd), P(y), B, T, H, W, Cc, S()))
    MM, MV = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
    stats = torch.zeros(4 * T * Cc, device=DEV)
    ws0 = torch.zeros(T * 256 * 2 * Cc, dtype=torch.float64, device=DEV)
    scratch = torch.zeros((N * Ho * Wo, Cc), device=DEV)
    _lib.check(lib.cdrl_bn_train_fwd(P(y), T, B * Ho * Wo, Cc, P(Gm), P(Bt), P(MM), P(MV), 1, 1, P(scratch), Cc, 0, 0, P(stats), P(ws0), S()))
    pool = torch.zeros((N, Hp, Wp, Cc), device=DEV)
    am = torch.zeros((N, Hp, Wp, Cc), dtype=torch.uint8, device=DEV)
    _lib.check(lib.cdrl_maxpool_bn_fwd(P(y), P(stats), T, B, P(pool), P(am), N, Ho, Wo, Cc, S()))
    assert rel_err(pool.cpu().numpy(), ref.detach().permute(0, 2, 3, 1).numpy()) < 1e-5
    ws = torch.zeros(int(lib.cdrl_stem_block_bwd_workspace_doubles(B, T, H, W, Cc)), dtype=torch.float64, device=DEV)
    dg, dbt, coef = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV), torch.zeros(3 * T * Cc, device=DEV)
    dw, db = torch.zeros((3, 3, 3, Cc), device=DEV), torch.zeros(Cc, device=DEV)
    _lib.check(lib.cdrl_stem_block_bwd(P(X), P(y), P(stats), P(am), P(DP), B, T, H, W, Cc, P(dg), P(dbt), P(coef), P(dw), P(db), P(ws), S()))
    assert rel_err(dg.cpu().numpy(), p['b.gamma'].grad.numpy()) < 2e-5
    assert rel_err(dbt.cpu().numpy(), p['b.beta'].grad.numpy()) < 2e-5
    assert rel_err(dw.cpu().numpy(), wt.grad.numpy()) < 3e-5
    assert np.abs(db.cpu().numpy()).max() < 1e-4 * np.abs(dw.cpu().numpy()).max()
    # the pooled-output form (mask and xhat from the pooled activated output instead of the argmax gather: what the float32 engine runs)
    dg2, dbt2, coef2 = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV), torch.zeros(3 * T * Cc, device=DEV)
    dw2, db2 = torch.zeros((3, 3, 3, Cc), device=DEV), torch.zeros(Cc, device=DEV)
    _lib.check(lib.cdrl_stem_block_bwd_pooled(P(X), P(y), P(stats), P(am), P(DP), P(pool), B, T, H, W, Cc, P(dg2), P(dbt2), P(coef2), P(dw2),
                                              P(db2), P(ws), S()))
    assert rel_err(dg2.cpu().numpy(), p['b.gamma'].grad.numpy()) < 2e-5
    assert rel_err(dbt2.cpu().numpy(), p['b.beta'].grad.numpy()) < 2e-5
    assert rel_err(dw2.cpu().numpy(), wt.grad.numpy()) < 3e-5
    assert torch.equal(dbt, dbt2)                                       # sum dz: same decisions, same addends
    assert int((am >= 128).sum()) > 0 and int((am.to(torch.int32) & 127).max()) <= 8        # ReLU6 flag in bit 7 of the codes


@pytest.mark.parametrize('M,Cc', [(256, 512), (256, 320), (37, 352), (1024, 16), (5, 3)])
def test_bn_small(lib, M, Cc):
    """Single-group dense BatchNorm as one launch per direction (reference core/networks.py:59-66: BatchNormalization on
    (B, C) activations, training mode) against fp64 numpy: outputs, moving statistics, dgamma / dbeta, input gradient."""
    rng = np.random.default_rng(M + Cc)
    y = (rng.standard_normal((M, Cc)) * rng.uniform(0.5, 2.0, Cc) + rng.uniform(-1, 1, Cc)).astype(np.float32)
    gam, bet = rng.uniform(0.5, 1.5, Cc).astype(np.float32), rng.uniform(-1, 1, Cc).astype(np.float32)
    mm0, mv0 = rng.standard_normal(Cc).astype(np.float32), rng.uniform(0.5, 2.0, Cc).astype(np.float32)
    dout = rng.standard_normal((M, Cc)).astype(np.float32)
    Y, G_, B_, MM, MV, DO = dev(y), dev(gam), dev(bet), dev(mm0), dev(mv0), dev(dout)
    stats, out = torch.zeros(4 * Cc, device=DEV), torch.zeros((M, Cc), device=DEV)
    _lib.check(lib.cdrl_bn_small_fwd(P(Y), M, Cc, P(G_), P(B_), P(MM), P(MV), P(stats), P(out), S()))
    y64 = y.astype(np.float64)
    mean, var = y64.mean(0), y64.var(0)
    inv = 1.0 / np.sqrt(var + 1e-3)
    xh = (y64 - mean) * inv
    assert rel_err(out.cpu().numpy(), xh * gam + bet) < 1e-5
    assert rel_err(MM.cpu().numpy(), mm0 - (mm0 - mean) * 0.01) < 1e-5
    assert rel_err(MV.cpu().numpy(), mv0 - (mv0 - var) * 0.01) < 1e-5
    dg, db, coef, dx = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV), torch.zeros(3 * Cc, device=DEV), torch.zeros((M, Cc), device=DEV)
    _lib.check(lib.cdrl_bn_small_bwd(P(DO), P(Y), M, Cc, P(stats), P(dg), P(db), P(coef), P(dx), S()))
    d64 = dout.astype(np.float64)
    assert rel_err(db.cpu().numpy(), d64.sum(0)) < 2e-5
    assert rel_err(dg.cpu().numpy(), (d64 * xh).sum(0)) < 2e-5
    ref_dx = gam * inv * (d64 - d64.mean(0) - xh * (d64 * xh).mean(0))
    assert rel_err(dx.cpu().numpy(), ref_dx) < 2e-5


@pytest.mark.parametrize('B,K,dims', [(256, 320, (2, 2, 1, 1)), (256, 320, (1, 1, 1, 1)), (37, 64, (3, 3, 1, 1)), (1, 320, (2, 2, 1, 1)),
                                      (300, 33, (2,))])
def test_linear_heads(lib, B, K, dims):
    """All linear output heads of a control branch in one launch per direction (core/networks.py:128-137, 267-275)
    against fp64 numpy: outputs, input gradient, weight and bias gradients."""
    rng = np.random.default_rng(B + K + sum(dims))
    a = rng.standard_normal((B, K)).astype(np.float32)
    ws = [(rng.standard_normal((K, n)) / np.sqrt(K)).astype(np.float32) for n in dims]
    bs = [rng.standard_normal(n).astype(np.float32) for n in dims]
    L = sum(dims)
    dlin = rng.standard_normal((B, L)).astype(np.float32)
    A_, DL = dev(a), dev(dlin)
    Wd, Bd = [dev(w) for w in ws], [dev(b) for b in bs]
    nh = len(dims)
    n_arr = (C.c_int * nh)(*dims)
    ptrs = lambda ts: (C.c_void_p * nh)(*[t.data_ptr() for t in ts])
    lin = torch.zeros((B, L), device=DEV)
    _lib.check(lib.cdrl_linear_heads_fwd(P(A_), nh, n_arr, ptrs(Wd), ptrs(Bd), P(lin), B, K, S()))
    a64 = a.astype(np.float64)
    ref = np.concatenate([a64 @ w.astype(np.float64) + b for w, b in zip(ws, bs)], axis=1)
    assert rel_err(lin.cpu().numpy(), ref) < 1e-5
    da = torch.full((B, K), 7.0, device=DEV)
    dW, dB = [torch.zeros_like(w) for w in Wd], [torch.zeros_like(b) for b in Bd]
    _lib.check(lib.cdrl_linear_heads_bwd(P(A_), nh, n_arr, ptrs(Wd), P(DL), P(da), ptrs(dW), ptrs(dB), B, K, S()))
    d64 = dlin.astype(np.float64)
    off = 0
    ref_da = np.zeros((B, K))
    for h, n in enumerate(dims):
        dh = d64[:, off:off + n]
        ref_da += dh @ ws[h].astype(np.float64).T
        assert rel_err(dW[h].cpu().numpy(), a64.T @ dh) < 1e-5
        assert rel_err(dB[h].cpu().numpy(), dh.sum(0)) < 1e-5
        off += n
    assert rel_err(da.cpu().numpy(), ref_da) < 1e-5


@pytest.mark.parametrize('B,A,faithful', [(256, 2, True), (37, 3, False), (1024, 2, True)])
def test_policy_loss(lib, B, A, faithful):
    rng = np.random.default_rng(B + A)
    L = 2 * A + 2
    lin = (rng.standard_normal((B, L)) * 1.5).astype(np.float32)
    adv = rng.standard_normal(B).astype(np.float32)
    u = np.clip(rng.beta(2, 2, (B, A)), 1e-4, 1 - 1e-4).astype(np.float32)
    u[0, 0] = 0.0       # exercises _clip_actions (clipped sample -> no pathwise gradient)
    u[1, 0] = 1.0
    old = (rng.standard_normal((B, A)) * 0.3).astype(np.float32)
    spd = rng.uniform(0, 0.3, B).astype(np.float32)
    sim = rng.uniform(-1, 1, B).astype(np.float32)
    ja = rng.uniform(-0.3, 0.3, (B, A)).astype(np.float32) if faithful else None
    jb = rng.uniform(-0.3, 0.3, (B, A)).astype(np.float32) if faithful else None
    lt = torch.tensor(lin, dtype=torch.float64, requires_grad=True)
    alpha = OM.softplus101(lt[:, :A])
    beta = OM.softplus101(lt[:, A:2 * A])
    tt = lambda x: torch.tensor(x, dtype=torch.float64)
    z = np.zeros((B, A), np.float32)
    us = OM._InjectedSample.apply(alpha, beta, tt(u), tt(ja if faithful else z), tt(jb if faithful else z))
    x = torch.clamp(us, OM.EPSILON, 1.0 - OM.EPSILON)
    logp = OM.beta_log_prob(x, alpha, beta)
    ent = OM.beta_entropy(alpha, beta).mean()
    ratio = torch.exp(logp - tt(old)).mean(dim=1)
    c = 0.2
    advt = tt(adv)
    min_adv = torch.where(advt > 0, (1 + c) * advt, (1 - c) * advt)
    simp = torch.tanh(lt[:, 2 * A])
    spdp = 2.0 * torch.sigmoid(lt[:, 2 * A + 1])
    pl = -torch.minimum(ratio * advt, min_adv).mean()
    total = pl - 0.7 * ent + 0.5 * ((tt(spd) - spdp) ** 2).mean() + 0.5 * ((tt(sim) - simp) ** 2).mean()
    total.backward()
    dlin = torch.zeros((B, L), device=DEV)
    metrics = torch.zeros(16, device=DEV)
    aux = torch.zeros((B, 4 * A), device=DEV)
    hp = torch.zeros(16, device=DEV)
    args = [dev(lin), dev(adv), dev(old), dev(spd), dev(sim), dev(u), dev(ja) if faithful else None, dev(jb) if faithful else None]
    _lib.check(lib.cdrl_beta_ppo_loss(*[P(a) for a in args], 0.2, 0.7, B, A, 1.0, P(dlin), P(metrics), P(aux), P(hp), S()))
    m = metrics.cpu().numpy()
    assert abs(m[0] - total.item()) < 1e-5 * max(1.0, abs(total.item()))
    assert abs(m[1] - pl.item()) < 1e-5 * max(1.0, abs(pl.item()))
    assert abs(m[2] - ent.item()) < 1e-5
    assert rel_err(dlin.cpu().numpy(), lt.grad.numpy()) < 1e-5
    ax = aux.cpu().numpy()
    assert rel_err(ax[:, :A], alpha.detach().numpy()) < 1e-6
    assert rel_err(ax[:, 2 * A:3 * A], logp.detach().numpy()) < 1e-5


def test_value_loss(lib):
    B = 200
    rng = np.random.default_rng(3)
    lin = rng.standard_normal((B, 4)).astype(np.float32)
    ret = np.stack([rng.uniform(-1, 1, B), rng.integers(0, 6, B)], 1).astype(np.float32)
    spd = rng.uniform(0, 0.3, B).astype(np.float32)
    sim = rng.uniform(-1, 1, B).astype(np.float32)
    lt = torch.tensor(lin, dtype=torch.float64, requires_grad=True)
    tt = lambda x: torch.tensor(x, dtype=torch.float64)
    base, ex = torch.tanh(lt[:, 0]), 6.0 * torch.sigmoid(lt[:, 1])
    sp, si = 2.0 * torch.sigmoid(lt[:, 2]), torch.tanh(lt[:, 3])
    vl = 0.25 * ((tt(ret[:, 0]) - base) ** 2).mean() + ((tt(ret[:, 1]) - ex) ** 2).mean() / 36.0
    total = 0.25 * (vl + ((tt(spd) - sp) ** 2).mean() + ((tt(sim) - si) ** 2).mean())
    total.backward()
    dlin = torch.zeros((B, 4), device=DEV)
    metrics = torch.zeros(16, device=DEV)
    vals = torch.zeros((B, 2), device=DEV)
    keep = [dev(lin), dev(ret), dev(spd), dev(sim)]      # keep the device tensors alive across the launch
    _lib.check(lib.cdrl_value_loss(*[P(k) for k in keep], B, 6.0, 1.0, P(dlin), P(metrics), P(vals), S()))
    assert abs(metrics.cpu().numpy()[0] - total.item()) < 1e-6 * max(1.0, abs(total.item()))
    assert rel_err(dlin.cpu().numpy(), lt.grad.numpy()) < 1e-5


@pytest.mark.parametrize('B,In,u,T', [(256, 768, 256, 4), (37, 16, 32, 4), (1, 16, 32, 3), (64, 40, 96, 2)])
def test_gru_steps_vs_fp64_autograd(lib, B, In, u, T):
    """Fused GRU time-step kernels (cdrl_gru_step_fwd / _bwd) composed into a T-step GRU against float64 autograd of the
    oracle's Keras-GRU restatement (oracle/model.py::gru_last): last hidden state, every saved gate tensor, and the
    gradients w.r.t. the input projection, the recurrent pre-activations (-> dK, dR, db) and the input."""
    rng = np.random.default_rng(B + In + u)
    x = rng.standard_normal((T, B, In))
    p64 = {'g.kernel': torch.tensor(rng.standard_normal((In, 3 * u)) / np.sqrt(In), requires_grad=True),
           'g.recurrent': torch.tensor(rng.standard_normal((u, 3 * u)) / np.sqrt(u), requires_grad=True),
           'g.bias': torch.tensor(rng.standard_normal((2, 3 * u)) * 0.1, requires_grad=True)}
    x64 = torch.tensor(x, requires_grad=True)
    h_ref = OM.gru_last(x64, p64, 'g')
    gout = rng.standard_normal((B, u))
    h_ref.backward(torch.tensor(gout))
    K, R, b = (p64[k].detach().float().to(DEV).contiguous() for k in ('g.kernel', 'g.recurrent', 'g.bias'))
    xd = dev(x.astype(np.float32))
    XP = (xd.reshape(T * B, In) @ K + b[0]).reshape(T, B, 3 * u).contiguous()      # input projection: one batched GEMM in the engine
    Hs = torch.zeros((T + 1, B, u), device=DEV)
    Z, Rg, HH = (torch.empty((T, B, u), device=DEV) for _ in range(3))
    HP = torch.empty((T, B, 3 * u), device=DEV)
    b1 = b[1].contiguous()
    for t in range(T):
        _lib.check(lib.cdrl_gru_step_fwd(P(XP[t]), P(Hs[t]), P(R), P(b1), P(Z[t]), P(Rg[t]), P(HH[t]), P(HP[t]), P(Hs[t + 1]), B, u, S()))
    assert rel_err(Hs[T].cpu().numpy(), h_ref.detach().numpy()) < 1e-5
    RT = R.t().contiguous()
    dXP = torch.zeros((T, B, 3 * u), device=DEV)
    dHP = torch.zeros((T, B, 3 * u), device=DEV)
    dh = [torch.empty((B, u + 3), device=DEV), torch.empty((B, u), device=DEV), torch.empty((B, u), device=DEV)]
    dh[0][:, :u] = dev(gout.astype(np.float32))                                     # strided first gradient (the concat slot)
    cur, ld = dh[0], u + 3
    for t in range(T - 1, -1, -1):
        nxt = dh[1] if cur is not dh[1] else dh[2]
        _lib.check(lib.cdrl_gru_step_bwd(P(cur), ld, P(Z[t]), P(Rg[t]), P(HH[t]), P(HP[t]), P(Hs[t]), P(RT), P(dXP[t]), P(dHP[t]),
                                         P(nxt) if t > 0 else None, B, u, S()))
        cur, ld = nxt, u
    dXPf, dHPf = dXP.reshape(T * B, 3 * u).double(), dHP.reshape(T * B, 3 * u).double()
    got = {'g.kernel': xd.reshape(T * B, In).double().t() @ dXPf,
           'g.recurrent': Hs[:T].reshape(T * B, u).double().t() @ dHPf,
           'g.bias': torch.stack([dXPf.sum(0), dHPf.sum(0)])}
    for k, g in got.items():
        assert rel_err(g.cpu().numpy(), p64[k].grad.numpy()) < 3e-5, k
    dx = (dXPf @ K.double().t()).reshape(T, B, In)
    assert rel_err(dx.cpu().numpy(), x64.grad.numpy()) < 3e-5


@pytest.mark.parametrize('G,Mg,K,N,pro', [(4, 1000, 116, 116, True), (4, 777, 116, 116, False), (2, 515, 24, 56, True), (1, 4100, 60, 92, False),
                                          (3, 64, 28, 28, False), (4, 12288, 116, 116, True),
                                          # K or N above 128: the one-tile-per-workgroup form of the 232-channel convs (stage 2)
                                          (4, 3072, 232, 232, True), (4, 3072, 232, 232, False), (2, 1001, 232, 140, True), (3, 333, 116, 232, False)])
def test_pwconv_x3_split(lib, G, Mg, K, N, pro):
    """Float32 1x1 conv on the bf16 matrix pipe (exact 3-way bf16 split of both operands, six MFMAs per K = 16 step): float32
    accuracy against the float64 reference (same bar as the float32-MFMA kernel), statistics epilogue, untouched padding."""
    rng = np.random.default_rng(G + Mg + K + N)
    M = G * Mg
    lda, a_coff, ldc, c_coff = K + 12, 4, N + 8, 4
    a = dev(rng.standard_normal((M, lda)).astype(np.float32))
    w = dev((rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32))
    bias = dev((rng.standard_normal(N) * 0.1).astype(np.float32))
    stats = None
    if pro:
        stats = dev(rng.uniform(0.5, 1.5, (4, G, K)).astype(np.float32))
        stats[3] = dev(rng.uniform(-0.3, 0.3, (G, K)).astype(np.float32))
    wp = torch.zeros(int(lib.cdrl_pwconv_x3_packed_bytes_n(K, N)), dtype=torch.uint8, device=DEV)
    _lib.check(lib.cdrl_pwconv_x3_pack(P(w), K, N, N, 1, P(wp), S()))
    c = torch.full((M, ldc), 7.0, device=DEV)
    nb = int(lib.cdrl_pwconv_x3_partial_rows(G, Mg, N, K))
    part = torch.zeros((G, nb, 2, N), dtype=torch.float64, device=DEV)
    _lib.check(lib.cdrl_pwconv_x3(P(a), lda, a_coff, P(stats), P(wp), P(bias), P(c), ldc, c_coff, G, Mg, N, K, P(part), S()))
    av = a[:, a_coff:a_coff + K].double().view(G, Mg, K)
    if pro:
        av = (stats[2].view(G, 1, K).double() * av + stats[3].view(G, 1, K).double()).float().double()     # fmaf, one float32 rounding
    ref = (av.view(M, K) @ w.double() + bias.double()).cpu().numpy()
    got = c[:, c_coff:c_coff + N]
    assert rel_err(got.cpu().numpy(), ref) < 1e-5
    assert bool((c[:, :c_coff] == 7.0).all()) and bool((c[:, c_coff + N:] == 7.0).all())
    sums = part.sum(dim=1)
    g64 = got.double().view(G, Mg, N)
    assert torch.allclose(sums[:, 0], g64.sum(1), rtol=1e-9, atol=1e-6) and torch.allclose(sums[:, 1], (g64 * g64).sum(1), rtol=1e-9, atol=1e-6)
    # transposed operand (backward-data orientation): B(k, n) = W[n][k]
    wp2 = torch.zeros(int(lib.cdrl_pwconv_x3_packed_bytes_n(N, K)), dtype=torch.uint8, device=DEV)
    if N % 4 == 0:
        _lib.check(lib.cdrl_pwconv_x3_pack(P(w), N, K, 1, N, P(wp2), S()))
        d = torch.zeros((M, K), device=DEV)
        gc = got.contiguous()
        _lib.check(lib.cdrl_pwconv_x3(P(gc), N, 0, None, P(wp2), None, P(d), K, 0, G, Mg, K, N, None, S()))
        assert rel_err(d.cpu().numpy(), (gc.double() @ w.double().t()).cpu().numpy()) < 1e-5


@pytest.mark.parametrize('G,Mg,K,N,pro', [(4, 1237, 58, 58, True), (4, 1000, 58, 58, False), (2, 777, 58, 92, True), (3, 130, 26, 58, False)])
def test_pwconv_x3_dword_aligned_rows(lib, G, Mg, K, N, pro):
    """Round 6: the split-precision conv on rows that are only DWORD aligned -- the 58-channel halves of stage 0 (unit tensors of 116
    floats = 464-byte rows, branch half at channel offset 58; core/architectures.py:120-145 at num_channels 116): 16-byte buffer loads at
    4-byte alignment, the two columns beyond K inside the last chunk belong to the NEIGHBOURING half (non-finite here on purpose) and
    must not reach the product."""
    rng = np.random.default_rng(G + Mg + K + N)
    M = G * Mg
    lda, a_coff, ldc, c_coff = 2 * K, K, N + 6, 2
    a_np = rng.standard_normal((M, lda)).astype(np.float32)
    a_np[:, :a_coff] = np.nan                                       # the other half of the tensor: never read into the product
    a = dev(a_np)
    a = torch.cat([a, torch.full((1, lda), float('nan'), device=DEV)])[:M]       # (keeps the allocation's tail defined)
    w = dev((rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32))
    bias = dev((rng.standard_normal(N) * 0.1).astype(np.float32))
    stats = None
    if pro:
        stats = dev(rng.uniform(0.5, 1.5, (4, G, K)).astype(np.float32))
        stats[3] = dev(rng.uniform(-0.3, 0.3, (G, K)).astype(np.float32))
    wp = torch.zeros(int(lib.cdrl_pwconv_x3_packed_bytes_n(K, N)), dtype=torch.uint8, device=DEV)
    _lib.check(lib.cdrl_pwconv_x3_pack(P(w), K, N, N, 1, P(wp), S()))
    c = torch.full((M, ldc), 7.0, device=DEV)
    nb = int(lib.cdrl_pwconv_x3_partial_rows(G, Mg, N, K))
    part = torch.zeros((G, nb, 2, N), dtype=torch.float64, device=DEV)
    _lib.check(lib.cdrl_pwconv_x3(P(a), lda, a_coff, P(stats), P(wp), P(bias), P(c), ldc, c_coff, G, Mg, N, K, P(part), S()))
    av = a[:, a_coff:a_coff + K].double().view(G, Mg, K)
    if pro:
        av = (stats[2].view(G, 1, K).double() * av + stats[3].view(G, 1, K).double()).float().double()
    ref = (av.view(M, K) @ w.double() + bias.double()).cpu().numpy()
    got = c[:, c_coff:c_coff + N]
    assert bool(torch.isfinite(got).all())
    assert rel_err(got.cpu().numpy(), ref) < 1e-5
    assert bool((c[:, :c_coff] == 7.0).all()) and bool((c[:, c_coff + N:] == 7.0).all())
    sums = part.sum(dim=1)
    g64 = got.double().view(G, Mg, N)
    assert torch.allclose(sums[:, 0], g64.sum(1), rtol=1e-9, atol=1e-6) and torch.allclose(sums[:, 1], (g64 * g64).sum(1), rtol=1e-9, atol=1e-6)


@pytest.mark.parametrize('G,Mg,Cin,Cout,shuffle,act,epi,acc', [(4, 3072, 232, 232, 1, 1, 1, 0), (4, 3072, 232, 232, 0, 0, 0, 0), (4, 1000, 232, 232, 0, 0, 0, 1),
                                                               (2, 333, 232, 140, 1, 1, 0, 1), (3, 97, 116, 232, 0, 1, 1, 0), (1, 64, 232, 232, 1, 0, 0, 0)])
def test_pwconv_x3_wide_bwd(lib, G, Mg, Cin, Cout, shuffle, act, epi, acc):
    """Round 6: backward-data of the 232-channel convs (stage 2) on the one-tile-per-workgroup split-precision kernel, with the
    BatchNorm-backward prologue (shuffle gather, ReLU6 mask), the column sums of dy, the BatchNorm-sum epilogue and the accumulate
    variant -- against a float64 evaluation of the same formulas (reference: tape.gradient through Conv2D(k=1) -> BatchNormalization,
    core/architectures.py:130-141)."""
    rng = np.random.default_rng(G * Mg + Cin + Cout + 7 * shuffle + 3 * act + epi + 2 * acc)
    M = G * Mg
    ctot = 2 * Cout if shuffle else Cout + 8
    coff = Cout if shuffle else 4
    dz = rng.standard_normal((M, ctot)).astype(np.float32)
    y = (rng.standard_normal((M, Cout)) * 1.5 + 1.0).astype(np.float32)
    stats = np.stack([rng.uniform(0.5, 1.5, (G, Cout)), rng.uniform(0.5, 1.5, (G, Cout)), rng.uniform(0.5, 1.5, (G, Cout)),
                      rng.uniform(-0.5, 2.0, (G, Cout))]).astype(np.float32)
    coef = np.stack([rng.uniform(0.5, 1.5, (G, Cout)), rng.uniform(-0.1, 0.1, (G, Cout)), rng.uniform(-0.1, 0.1, (G, Cout))]).astype(np.float32)
    w = (rng.standard_normal((Cin, Cout)) / np.sqrt(Cout)).astype(np.float32)
    idx = [((coff + c) & 1) * (ctot // 2) + ((coff + c) >> 1) for c in range(Cout)] if shuffle else [coff + c for c in range(Cout)]
    d64 = dz[:, idx].astype(np.float64).reshape(G, Mg, Cout)
    y64 = y.astype(np.float64).reshape(G, Mg, Cout)
    st = stats.astype(np.float64)[:, :, None, :]
    cf = coef.astype(np.float64)[:, :, None, :]
    if act:
        z = (st[2] * y64 + st[3]).astype(np.float32)        # fmaf(scale, y, shift): one float32 rounding
        d64 = d64 * ((z > 0) & (z < 6))
    xh = (y64 - st[0]) * st[1]
    dy = cf[0] * (d64 - cf[1] - xh * cf[2])
    ref = (dy.reshape(M, Cout) @ w.astype(np.float64).T)
    ldda, dco = Cin + 8, 4
    da0 = rng.standard_normal((M, ldda)).astype(np.float32)
    if acc:
        ref_out = ref + da0[:, dco:dco + Cin]
    else:
        ref_out = ref
    DZ, Y, ST, CF, Wd, DA = dev(dz), dev(y), dev(stats), dev(coef), dev(w), dev(da0)
    wp = torch.zeros(int(lib.cdrl_pwconv_x3_packed_bytes_n(Cout, Cin)), dtype=torch.uint8, device=DEV)
    _lib.check(lib.cdrl_pwconv_x3_pack(P(Wd), Cout, Cin, 1, Cout, P(wp), S()))
    rows = int(lib.cdrl_pwconv_x3_wide_bwd_rows(Mg))
    part2 = torch.zeros((G, rows, Cout), dtype=torch.float64, device=DEV)
    ey = est = part = None
    if epi:
        ey_np = rng.standard_normal((M, Cin)).astype(np.float32)
        est_np = np.stack([rng.uniform(-0.5, 0.5, (G, Cin)), rng.uniform(0.5, 1.5, (G, Cin)), np.ones((G, Cin)), np.zeros((G, Cin))]).astype(np.float32)
        ey, est = dev(ey_np), dev(est_np)
        part = torch.zeros((G, rows, 2, Cin), dtype=torch.float64, device=DEV)
    _lib.check(lib.cdrl_pwconv_x3_wide_bwd(P(DZ), ctot, coff, ctot if shuffle else 0, act, P(Y), P(ST), P(CF), P(wp), P(DA), ldda, dco, acc, G, Mg,
                                           Cin, Cout, P(part2), P(ey), P(est), P(part), S()))
    got = DA.cpu().numpy()
    assert rel_err(got[:, dco:dco + Cin], ref_out) < 1e-5
    assert np.array_equal(got[:, :dco], da0[:, :dco]) and np.array_equal(got[:, dco + Cin:], da0[:, dco + Cin:])        # padding untouched
    assert rel_err(part2.sum(dim=1).cpu().numpy(), dy.sum(axis=1)) < 1e-6
    if epi:
        g3 = (got[:, dco:dco + Cin].astype(np.float64) - (da0[:, dco:dco + Cin] if acc else 0.0)).reshape(G, Mg, Cin)
        xe = (ey_np.astype(np.float64).reshape(G, Mg, Cin) - est_np[0].astype(np.float64)[:, None, :]) * est_np[1].astype(np.float64)[:, None, :]
        ps = part.sum(dim=1).cpu().numpy()
        # (xhat is evaluated in float32 by the kernel, as by pw_nn's epilogue: 3-4e-8 measured)
        assert rel_err(ps[:, 0], g3.sum(axis=1)) < 1e-9 and rel_err(ps[:, 1], (g3 * xe).sum(axis=1)) < 1e-6
    # bit-wise reproducible
    DA2 = dev(da0)
    part2b = torch.zeros_like(part2)
    _lib.check(lib.cdrl_pwconv_x3_wide_bwd(P(DZ), ctot, coff, ctot if shuffle else 0, act, P(Y), P(ST), P(CF), P(wp), P(DA2), ldda, dco, acc, G, Mg,
                                           Cin, Cout, P(part2b), P(ey), P(est), P(part), S()))
    assert torch.equal(DA, DA2) and torch.equal(part2, part2b)


@pytest.mark.parametrize('M,K,N', [(12288, 464, 768), (12288, 768, 464), (1000, 232, 232), (130, 464, 768), (4099, 60, 92), (257, 16, 8)])
def test_gemm_x3_split(lib, M, K, N):
    """General split-precision GEMM (head conv shapes, forward and backward-data orientation, ragged edges): float32 accuracy
    vs float64, bias, accumulate into a strided view."""
    rng = np.random.default_rng(M + K + N)
    lda, a_coff = K + 8, 4
    a = dev(rng.standard_normal((M, lda)).astype(np.float32))
    b = dev((rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32))
    bias = dev(rng.standard_normal(N).astype(np.float32))
    bp = torch.zeros(int(lib.cdrl_gemm_x3_packed_bytes(N, K)), dtype=torch.uint8, device=DEV)
    _lib.check(lib.cdrl_gemm_x3_pack(P(b), K, N, N, 1, P(bp), S()))
    c = torch.full((M, N + 5), 3.0, device=DEV)
    _lib.check(lib.cdrl_gemm_x3(P(a), lda, a_coff, P(bp), P(bias), P(c), N + 5, 2, M, N, K, 0, S()))
    ref = a[:, a_coff:a_coff + K].double() @ b.double() + bias.double()
    assert rel_err(c[:, 2:2 + N].cpu().numpy(), ref.cpu().numpy()) < 1e-5
    assert bool((c[:, :2] == 3.0).all()) and bool((c[:, 2 + N:] == 3.0).all())
    # transposed operand + accumulate: d += c_block @ b^T
    if N % 4 == 0:
        bt = torch.zeros(int(lib.cdrl_gemm_x3_packed_bytes(K, N)), dtype=torch.uint8, device=DEV)
        _lib.check(lib.cdrl_gemm_x3_pack(P(b), N, K, 1, N, P(bt), S()))
        g = dev(rng.standard_normal((M, N)).astype(np.float32))
        d = torch.ones((M, K), device=DEV)
        _lib.check(lib.cdrl_gemm_x3(P(g), N, 0, P(bt), None, P(d), K, 0, M, K, N, 1, S()))
        assert rel_err(d.cpu().numpy(), (g.double() @ b.double().t() + 1.0).cpu().numpy()) < 1e-5


@pytest.mark.parametrize('G,Mg,K,N,relu,shuffle,anorm,acc,generic', [
    (4, 330, 58, 58, 1, 1, 1, 0, 0), (4, 96, 116, 116, 1, 1, 1, 0, 0), (4, 4100, 116, 116, 1, 0, 0, 1, 0), (2, 77, 58, 58, 0, 0, 0, 1, 1),
    (4, 100, 116, 116, 1, 1, 1, 0, 1), (1, 64, 40, 60, 0, 0, 1, 0, 1), (2, 8300, 58, 58, 1, 1, 1, 0, 0), (4, 1500, 116, 116, 1, 1, 0, 0, 1),
    (4, 12288, 116, 116, 1, 1, 1, 0, 0), (4, 330, 58, 92, 1, 1, 1, 0, 0), (2, 4300, 58, 92, 1, 0, 0, 1, 1), (4, 77, 36, 100, 0, 1, 1, 0, 1),
    (4, 2700, 24, 58, 1, 0, 0, 1, 0),      # (the first unit's conv on the pooled stem output, accumulating -- fused since round 5)
    (4, 2700, 24, 24, 1, 1, 1, 0, 0), (2, 333, 32, 16, 0, 0, 1, 0, 1)])     # (round 6: the 24-channel shortcut conv; N <= 32: a weight pack of two K steps)
def test_pwconv_bwd_fused(lib, G, Mg, K, N, relu, shuffle, anorm, acc, generic):
    """cdrl_pwconv_bwd_fused: BatchNorm-backward apply on load + backward-data + filter / bias gradient (+ the backward sums of
    the BatchNorm in FRONT of the conv, derived from the filter product) in one pass, against a float64 numpy evaluation of
    the same formulas (core/architectures.py:130-141 under autograd).  generic = 1: arbitrary (k2, k3) coefficients, so that
    the bias gradient -- analytically zero behind a train-mode BatchNorm -- and every term that rides on it is exercised."""
    rng = np.random.default_rng(G * Mg + K + 3 * N + relu + 2 * shuffle + 4 * anorm)
    M = G * Mg
    f64 = np.float64
    x = rng.standard_normal((M, K)).astype(np.float32) * rng.uniform(0.5, 2.0, K).astype(np.float32) + rng.uniform(-1, 1, K).astype(np.float32)
    w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    xg = x.astype(f64).reshape(G, Mg, K)
    ga, ba = rng.uniform(0.5, 1.5, K).astype(np.float32), rng.uniform(-0.5, 0.5, K).astype(np.float32)
    amean = xg.mean(axis=1).astype(np.float32)
    ainv = (1.0 / np.sqrt(xg.var(axis=1) + 1e-3)).astype(np.float32)
    ast = np.stack([amean, ainv, ga[None] * ainv, ba[None] - amean * (ga[None] * ainv)]).astype(np.float32)      # [4][G][K]
    xh_a = (xg - ast[0].astype(f64)[:, None, :]) * ast[1].astype(f64)[:, None, :]
    a = xh_a * ga.astype(f64) + ba.astype(f64) if anorm else xg
    y64 = a @ w.astype(f64) + bias.astype(f64)
    y = y64.astype(np.float32)                                          # the stored raw conv output
    yg = y.astype(f64)
    gy, by = rng.uniform(0.5, 1.5, N).astype(np.float32), rng.uniform(1.0, 3.0, N).astype(np.float32)
    ymean = yg.mean(axis=1).astype(np.float32)
    yinv = (1.0 / np.sqrt(yg.var(axis=1) + 1e-3)).astype(np.float32)
    ysc = (gy[None] * yinv).astype(np.float32)
    ysh = (by[None] - ymean * ysc).astype(np.float32)
    yst = np.stack([ymean, yinv, ysc, ysh]).astype(np.float32)          # [4][G][N]
    ctot, coff = (2 * N, N) if shuffle else (N, 0)
    dout = rng.standard_normal((M, ctot)).astype(np.float32)
    idx = [((coff + c) & 1) * (ctot // 2) + ((coff + c) >> 1) for c in range(N)] if shuffle else list(range(N))
    dz = dout[:, idx].astype(f64).reshape(G, Mg, N)
    if relu:
        z = np.float32(ysc)[:, None, :] * y.reshape(G, Mg, N) + np.float32(ysh)[:, None, :]       # float32 fmaf region (ties are measure zero)
        z64 = ysc.astype(f64)[:, None, :] * yg + ysh.astype(f64)[:, None, :]
        safe = (np.abs(z64) > 1e-4) & (np.abs(z64 - 6.0) > 1e-4)       # elements away from the kinks decide identically
        dz = dz * ((z64 > 0) & (z64 < 6))
    xh_y = (yg - ymean.astype(f64)[:, None, :]) * yinv.astype(f64)[:, None, :]
    k2 = dz.mean(axis=1)
    k3 = (dz * xh_y).mean(axis=1)
    if generic:
        k2 = k2 + rng.uniform(-0.3, 0.3, k2.shape)
        k3 = k3 + rng.uniform(-0.3, 0.3, k3.shape)
    coef = np.stack([ysc, k2.astype(np.float32), k3.astype(np.float32)]).astype(np.float32)      # [3][G][N]
    dy = coef[0].astype(f64)[:, None, :] * (dz - coef[1].astype(f64)[:, None, :] - xh_y * coef[2].astype(f64)[:, None, :])
    if relu:            # kink-adjacent elements: use whatever side the float32 expression takes (both are valid float32 evaluations)
        m32 = (z > 0) & (z < 6)
        flip = ~safe & (m32 != ((z64 > 0) & (z64 < 6)))
        if flip.any():
            dzf = dout[:, idx].astype(f64).reshape(G, Mg, N) * m32
            dy = np.where(flip, coef[0].astype(f64)[:, None, :] * (dzf - coef[1].astype(f64)[:, None, :] - xh_y * coef[2].astype(f64)[:, None, :]), dy)
    da_ref = dy @ w.astype(f64).T                                       # (G, Mg, K)
    dw_ref = np.einsum('gmk,gmn->kn', a, dy)
    db_ref = dy.sum(axis=(0, 1))
    # device
    X, Wd, Y, YS, CF, DO = dev(x), dev(w), dev(y.reshape(M, N)), dev(yst), dev(coef), dev(dout)
    AS, GA, BA = dev(ast), dev(ga), dev(ba)
    wp = torch.zeros(int(lib.cdrl_pwconv_x3_packed_bytes(N)), dtype=torch.uint8, device=DEV)
    _lib.check(lib.cdrl_pwconv_x3_pack(P(Wd), N, K, 1, N, P(wp), S()))            # B(k = n_out, n = k_in) = W[n][k]
    qpart = torch.zeros(int(lib.cdrl_pwconv_bwd_fused_workspace(G, Mg, N, K, 0)), device=DEV)
    dbpart = torch.zeros(int(lib.cdrl_pwconv_bwd_fused_workspace(G, Mg, N, K, 1)), dtype=torch.float64, device=DEV)
    base = rng.standard_normal((M, K + 4)).astype(np.float32)
    dA = dev(base.copy())
    dW, dB = torch.full((K, N), 7.0, device=DEV), torch.full((N,), 7.0, device=DEV)
    adg, adb, acf = torch.zeros(K, device=DEV), torch.zeros(K, device=DEV), torch.zeros(3 * G * K, device=DEV)
    _lib.check(lib.cdrl_pwconv_bwd_fused(P(DO), ctot, coff, ctot if shuffle else 0, relu, P(Y), P(YS), P(CF), P(X), K, 0,
                                         P(AS) if anorm else None, P(GA) if anorm else None, P(BA) if anorm else None,
                                         P(adg) if anorm else None, P(adb) if anorm else None, P(acf) if anorm else None, P(Wd), P(wp),
                                         P(dA), K + 4, 2, acc, P(dW), P(dB), P(qpart), P(dbpart), G, Mg, N, K, S()))
    torch.cuda.synchronize()
    got = dA.cpu().numpy()
    exp = da_ref.reshape(M, K) + (base[:, 2:2 + K].astype(f64) if acc else 0.0)
    assert rel_err(got[:, 2:2 + K], exp) < 1e-5
    assert np.array_equal(got[:, :2], base[:, :2]) and np.array_equal(got[:, 2 + K:], base[:, 2 + K:])
    assert rel_err(dW.cpu().numpy(), dw_ref) < 1e-5
    scale_db = max(np.abs(db_ref).max(), 1e-4 * np.abs(dw_ref).max())
    assert np.abs(dB.cpu().numpy() - db_ref).max() < 2e-5 * scale_db + 1e-5 * np.abs(dw_ref).max()
    if anorm:
        s1 = da_ref.sum(axis=1)                                         # (G, K)
        s2 = (da_ref * xh_a).sum(axis=1)
        cf = acf.cpu().numpy().reshape(3, G, K)
        assert np.array_equal(cf[0], ast[2])
        sc = max(np.abs(s2).max(), np.abs(s1).max()) / Mg
        assert np.abs(cf[1] - s1 / Mg).max() < 2e-5 * sc
        assert np.abs(cf[2] - s2 / Mg).max() < 2e-5 * sc
        assert rel_err(adg.cpu().numpy(), s2.sum(axis=0)) < 2e-5
        assert np.abs(adb.cpu().numpy() - s1.sum(axis=0)).max() < 2e-5 * np.abs(s2.sum(axis=0)).max()
    # bit-wise reproducible (fixed-order partial sums, no atomics)
    dW2, dB2 = torch.zeros((K, N), device=DEV), torch.zeros(N, device=DEV)
    dA2 = dev(base.copy())
    _lib.check(lib.cdrl_pwconv_bwd_fused(P(DO), ctot, coff, ctot if shuffle else 0, relu, P(Y), P(YS), P(CF), P(X), K, 0,
                                         P(AS) if anorm else None, P(GA) if anorm else None, P(BA) if anorm else None,
                                         P(adg) if anorm else None, P(adb) if anorm else None, P(acf) if anorm else None, P(Wd), P(wp),
                                         P(dA2), K + 4, 2, acc, P(dW2), P(dB2), P(qpart), P(dbpart), G, Mg, N, K, S()))
    assert torch.equal(dW, dW2) and torch.equal(dB, dB2) and torch.equal(dA, dA2)


def _bf(x):
    """round-to-nearest-even to bf16, returned as float64 (numpy in / out)"""
    return torch.as_tensor(np.asarray(x, dtype=np.float32)).to(torch.bfloat16).to(torch.float64).numpy()


@pytest.mark.parametrize('G,Mg,K,N,relu,shuffle,anorm,acc', [
    (4, 330, 58, 58, 1, 1, 1, 0), (4, 96, 116, 116, 1, 1, 1, 0), (4, 4100, 116, 116, 1, 0, 0, 1), (2, 77, 58, 58, 0, 0, 0, 1),
    (4, 100, 116, 116, 1, 1, 1, 0), (2, 8300, 58, 58, 1, 1, 1, 0), (4, 1500, 116, 116, 1, 1, 0, 0), (4, 12288, 116, 116, 1, 1, 1, 0),
    (4, 2700, 24, 24, 1, 1, 1, 0)])
def test_pwconv_bwd_fused_bf16_storage(lib, G, Mg, K, N, relu, shuffle, anorm, acc):
    """The bf16-storage form of cdrl_pwconv_bwd_fused (configuration 3): bf16 tensors in HBM, float32 BatchNorm-backward prologue,
    ONE bf16 plane per MFMA operand.  Reference: float64 evaluation of the same contract -- dy, xhat / a and W rounded to bf16
    where the kernel rounds them, exact products and sums.  What remains is float32-vs-float64 in front of a rounding: an element
    within 1e-7 of a bf16 boundary lands on the other side (one bf16 ulp for that element), hence bounds at bf16 level for the
    stored tensor and much tighter ones for the sums."""
    rng = np.random.default_rng(7 + G * Mg + K + 3 * N + relu + 2 * shuffle + 4 * anorm)
    M = G * Mg
    f64 = np.float64
    x = _bf(rng.standard_normal((M, K)) * rng.uniform(0.5, 2.0, K) + rng.uniform(-1, 1, K))            # stored bf16 tensors
    w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    xg = x.reshape(G, Mg, K)
    ga, ba = rng.uniform(0.5, 1.5, K).astype(np.float32), rng.uniform(-0.5, 0.5, K).astype(np.float32)
    amean = xg.mean(axis=1).astype(np.float32)
    ainv = (1.0 / np.sqrt(xg.var(axis=1) + 1e-3)).astype(np.float32)
    ast = np.stack([amean, ainv, ga[None] * ainv, ba[None] - amean * (ga[None] * ainv)]).astype(np.float32)
    xh_a = (xg - ast[0].astype(f64)[:, None, :]) * ast[1].astype(f64)[:, None, :]
    a_op = _bf(xh_a.astype(np.float32)).reshape(G, Mg, K) if anorm else xg                              # the kernel's second operand
    a_val = xh_a * ga.astype(f64) + ba.astype(f64) if anorm else xg
    y = _bf(a_val @ w.astype(f64) + rng.standard_normal(N))                                              # stored raw conv output
    yg = y.reshape(G, Mg, N)
    gy, by = rng.uniform(0.5, 1.5, N).astype(np.float32), rng.uniform(1.0, 3.0, N).astype(np.float32)
    ymean = yg.mean(axis=1).astype(np.float32)
    yinv = (1.0 / np.sqrt(yg.var(axis=1) + 1e-3)).astype(np.float32)
    ysc = (gy[None] * yinv).astype(np.float32)
    ysh = (by[None] - ymean * ysc).astype(np.float32)
    yst = np.stack([ymean, yinv, ysc, ysh]).astype(np.float32)
    ctot, coff = (2 * N, N) if shuffle else (N, 0)
    dout = _bf(rng.standard_normal((M, ctot)))
    idx = [((coff + c) & 1) * (ctot // 2) + ((coff + c) >> 1) for c in range(N)] if shuffle else list(range(N))
    dz = dout[:, idx].reshape(G, Mg, N)
    if relu:
        z32 = ysc[:, None, :] * yg.astype(np.float32) + ysh[:, None, :]
        z64 = ysc.astype(f64)[:, None, :] * yg + ysh.astype(f64)[:, None, :]
        m = np.where(np.abs(z64 - np.round(z64 / 6.0) * 6.0) > 1e-4, (z64 > 0) & (z64 < 6), (z32 > 0) & (z32 < 6))
        dz = dz * m
    xh_y = (yg - ymean.astype(f64)[:, None, :]) * yinv.astype(f64)[:, None, :]
    coef = np.stack([ysc, dz.mean(axis=1).astype(np.float32) + rng.uniform(-0.3, 0.3, (G, N)).astype(np.float32),
                     (dz * xh_y).mean(axis=1).astype(np.float32) + rng.uniform(-0.3, 0.3, (G, N)).astype(np.float32)]).astype(np.float32)
    dy = coef[0].astype(f64)[:, None, :] * (dz - coef[1].astype(f64)[:, None, :] - xh_y * coef[2].astype(f64)[:, None, :])
    dyb = _bf(dy.astype(np.float32)).reshape(G, Mg, N)                                                  # the MFMA operand
    wb = _bf(w)
    da_ref = dyb @ wb.T
    q_ref = np.einsum('gmk,gmn->gkn', a_op, dyb)
    db_g = dy.sum(axis=1)                                                                               # float32 dy, summed in double
    db_ref = db_g.sum(axis=0)
    dw_ref = (ga.astype(f64)[:, None] * q_ref.sum(axis=0) + ba.astype(f64)[:, None] * db_ref[None, :]) if anorm else q_ref.sum(axis=0)
    bf16 = torch.bfloat16
    X, Wd, Y, YS, CF = dev(torch.as_tensor(x).to(bf16)), dev(w), dev(torch.as_tensor(y).to(bf16)), dev(yst), dev(coef)
    DO = dev(torch.as_tensor(dout).to(bf16))
    AS, GA, BA = dev(ast), dev(ga), dev(ba)
    wp = torch.zeros(int(lib.cdrl_pwconv_x3_packed_bytes(N)), dtype=torch.uint8, device=DEV)
    _lib.check(lib.cdrl_pwconv_x3_pack(P(Wd), N, K, 1, N, P(wp), S()))
    lib.cdrl_set_op_activation_type(1)
    try:
        qpart = torch.zeros(int(lib.cdrl_pwconv_bwd_fused_workspace(G, Mg, N, K, 0)), device=DEV)
        dbpart = torch.zeros(int(lib.cdrl_pwconv_bwd_fused_workspace(G, Mg, N, K, 1)), dtype=torch.float64, device=DEV)
        base = _bf(rng.standard_normal((M, K + 4)))
        outs = []
        for _ in range(2):
            dA = dev(torch.as_tensor(base).to(bf16))
            dW, dB = torch.full((K, N), 7.0, device=DEV), torch.full((N,), 7.0, device=DEV)
            adg, adb, acf = torch.zeros(K, device=DEV), torch.zeros(K, device=DEV), torch.zeros(3 * G * K, device=DEV)
            _lib.check(lib.cdrl_pwconv_bwd_fused(P(DO), ctot, coff, ctot if shuffle else 0, relu, P(Y), P(YS), P(CF), P(X), K, 0,
                                                 P(AS) if anorm else None, P(GA) if anorm else None, P(BA) if anorm else None,
                                                 P(adg) if anorm else None, P(adb) if anorm else None, P(acf) if anorm else None, P(Wd), P(wp),
                                                 P(dA), K + 4, 2, acc, P(dW), P(dB), P(qpart), P(dbpart), G, Mg, N, K, S()))
            torch.cuda.synchronize()
            outs.append((dA.clone(), dW.clone(), dB.clone(), adg.clone(), adb.clone(), acf.clone()))
    finally:
        lib.cdrl_set_op_activation_type(0)
    dA, dW, dB, adg, adb, acf = outs[0]
    for u, v in zip(outs[0], outs[1]):
        assert torch.equal(u, v)                                                                        # bit-wise reproducible
    got = dA.double().cpu().numpy()
    exp = da_ref.reshape(M, K) + (base[:, 2:2 + K] if acc else 0.0)
    assert rel_err(got[:, 2:2 + K], exp) < 6e-3                                                         # stored as bf16: one ulp of the maximum = 3.9e-3
    assert np.array_equal(got[:, :2], base[:, :2]) and np.array_equal(got[:, 2 + K:], base[:, 2 + K:])
    assert rel_err(dW.cpu().numpy(), dw_ref) < 5e-4                                                     # sums: single-element ulp flips average out
    assert np.abs(dB.cpu().numpy() - db_ref).max() < 1e-4 * max(np.abs(db_ref).max(), 1e-3 * np.abs(dw_ref).max())
    if anorm:
        # bf16 storage: the sums are taken directly from the float32 backward-data accumulators and the xhat operand plane
        # (float32 over a lane's 16 rows of a tile, double beyond), not derived from the filter product
        da_u = dyb @ wb.T
        s1 = da_u.sum(axis=1)
        s2 = (da_u * a_op).sum(axis=1)
        cf = acf.cpu().numpy().reshape(3, G, K)
        sc = max(np.abs(s2).max(), np.abs(s1).max()) / Mg
        assert np.array_equal(cf[0], ast[2])
        assert np.abs(cf[1] - s1 / Mg).max() < 5e-4 * sc and np.abs(cf[2] - s2 / Mg).max() < 5e-4 * sc
        assert rel_err(adg.cpu().numpy(), s2.sum(axis=0)) < 5e-4
        # ... which are the BatchNorm-backward sums of the (unrounded) da up to the operand rounding of xhat
        true_s2 = (da_u * xh_a).sum(axis=1)
        assert np.abs(s2 - true_s2).max() < 2e-2 * np.abs(true_s2).max()
        assert np.abs(adb.cpu().numpy() - s1.sum(axis=0)).max() < 5e-4 * np.abs(s2.sum(axis=0)).max()
