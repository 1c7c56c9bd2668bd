"""bf16 ACTIVATION STORAGE (BASELINE.json configs[2] in full: `cdrl_config.compute = CDRL_COMPUTE_BF16_STORAGE`, LearnerEngine(compute=
'bf16s')): every activation / activation-gradient tensor of the image tower is bf16 in HBM, everything a kernel computes with stays
float32 / double.

Kernel level -- the storage contract is exact, so the tests are exact: a bf16-storage kernel run on bf16 tensors must produce
  * for every activation output: bit for bit the round-to-nearest-even bf16 of what the SAME kernel in its float32-tensor form
    produces from the same values widened to float32 (the float32-tensor forms are the ones tests/test_gpu_ops.py and
    tests/test_gpu_bf16.py hold against float64);
  * for every float32 / double output (statistics blocks, coefficient blocks, dgamma / dbeta, filter / bias gradients): the same
    numbers -- except where the contract says the statistics are those of the STORED (rounded) values (conv / depthwise / stem
    outputs feeding a BatchNorm), which are checked against float64 sums of the rounded outputs.
Engine level: forward quantities against the bf16-operand engine (same arithmetic, float32 tensors), determinism, training that
tracks the float32 loss curves, and the size-independent properties at configs[2]'s own size (B = 1024, 90x120)."""
import ctypes as C

import numpy as np
import pytest
import torch

from carla_driving_rl_agent_amd import _lib

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
BF = torch.bfloat16


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def dev(x, dt=torch.float32):
    return torch.tensor(np.asarray(x, np.float32), device=DEV).to(dt)


class storage:
    """with storage(lib, 1): op-level entry points take / produce bf16 activation tensors."""

    def __init__(self, lib, at):
        self.lib, self.at = lib, at

    def __enter__(self):
        _lib.check(self.lib.cdrl_set_op_activation_type(self.at))

    def __exit__(self, *a):
        _lib.check(self.lib.cdrl_set_op_activation_type(0))


def same_bits(a_bf16, ref_f32):
    return torch.equal(a_bf16, ref_f32.to(BF))


@pytest.mark.parametrize('G,Mg,Cc,relu,shuffle', [(4, 700, 116, 1, 1), (4, 333, 58, 1, 1), (2, 1000, 24, 0, 0), (4, 96, 232, 1, 0), (3, 50, 768, 1, 0)])
def test_bn_train_bf16_storage(lib, G, Mg, Cc, relu, shuffle):
    rng = np.random.default_rng(G * Mg + Cc)
    yb = dev(rng.standard_normal((G * Mg, Cc)) * 2.0 + 0.7, BF)
    y32 = yb.float()
    gamma, beta = dev(rng.uniform(0.5, 1.5, Cc)), dev(rng.uniform(-0.5, 0.5, Cc))
    ctot, coff = (2 * Cc, Cc) if shuffle else (Cc, 0)
    dob = dev(rng.standard_normal((G * Mg, ctot)), BF)
    res = {}
    for at, Y, DO, dt in ((0, y32, dob.float(), torch.float32), (1, yb, dob, BF)):
        mm, mv = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
        out = torch.zeros((G * Mg, ctot), dtype=dt, device=DEV)
        stats = torch.zeros(4 * G * Cc, device=DEV)
        ws = torch.zeros(G * 256 * 2 * Cc, dtype=torch.float64, device=DEV)
        dg, dbt, coef = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV), torch.zeros(3 * G * Cc, device=DEV)
        dy = torch.zeros((G * Mg, Cc), dtype=dt, device=DEV)
        with storage(lib, at):
            _lib.check(lib.cdrl_bn_train_fwd(P(Y), G, Mg, Cc, P(gamma), P(beta), P(mm), P(mv), 1, relu, P(out), ctot, coff,
                                             ctot if shuffle else 0, P(stats), P(ws), S()))
            _lib.check(lib.cdrl_bn_train_bwd(P(DO), ctot, coff, ctot if shuffle else 0, P(Y), G, Mg, Cc, P(stats), relu, P(dg), P(dbt),
                                             P(dy), P(coef), P(ws), S()))
        res[at] = (out, stats, mm, mv, dg, dbt, coef, dy)
    r0, r1 = res[0], res[1]
    assert same_bits(r1[0], r0[0]) and same_bits(r1[7], r0[7])              # activations: the rounded float32 results
    for i in (1, 2, 3, 4, 5, 6):                                            # statistics, moving statistics, dgamma, dbeta, coefficients
        assert torch.equal(r0[i], r1[i]), i


@pytest.mark.parametrize('T,B,H,W,Cc,stride,pre', [(4, 8, 6, 8, 116, 1, True), (2, 6, 11, 15, 58, 1, True), (2, 4, 22, 30, 24, 2, False),
                                                    (4, 4, 11, 15, 116, 2, True), (2, 8, 3, 4, 232, 1, True)])
def test_dwconv_bn_bf16_storage(lib, T, B, H, W, Cc, stride, pre):
    rng = np.random.default_rng(T * B + H * W + Cc)
    N, Ho, Wo = T * B, -(-H // stride), -(-W // stride)
    xb = dev(rng.standard_normal((N, H, W, Cc)) * 1.5 + 0.4, BF)
    w, b = dev(rng.standard_normal((3, 3, Cc, 1))), dev(rng.standard_normal(Cc))
    dob = dev(rng.standard_normal((N, Ho, Wo, Cc)), BF)
    g1, b1 = dev(rng.uniform(0.5, 1.5, Cc)), dev(rng.uniform(1.0, 3.0, Cc))
    g2, b2 = dev(rng.uniform(0.5, 1.5, Cc)), dev(rng.uniform(-0.5, 0.5, Cc))
    res = {}
    for at, X, DO, dt in ((0, xb.float(), dob.float(), torch.float32), (1, xb, dob, BF)):
        with storage(lib, at):
            pre_stats = None
            if pre:
                pre_stats = torch.zeros(4 * T * Cc, device=DEV)
                tmp = torch.zeros((N * H * W, Cc), dtype=dt, device=DEV)
                ws0 = torch.zeros(T * 256 * 2 * Cc, dtype=torch.float64, device=DEV)
                mm, mv = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
                _lib.check(lib.cdrl_bn_train_fwd(P(X), T, B * H * W, Cc, P(g1), P(b1), P(mm), P(mv), 1, 1, P(tmp), Cc, 0, 0, P(pre_stats),
                                                 P(ws0), S()))
            ws = torch.zeros(int(lib.cdrl_dwconv_bn_workspace_doubles(T, B, H, W, Cc, stride)), dtype=torch.float64, device=DEV)
            y = torch.zeros((N, Ho, Wo, Cc), dtype=dt, device=DEV)
            post_stats = torch.zeros(4 * T * Cc, device=DEV)
            mm2, mv2 = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
            _lib.check(lib.cdrl_dwconv_bn_fwd(P(X), P(pre_stats), P(w), P(b), P(y), T, B, H, W, Cc, stride, P(g2), P(b2), P(mm2), P(mv2), 1,
                                              P(post_stats), P(ws), S()))
            res[(at, 'y')], res[(at, 'st')] = y, post_stats.clone()
    # forward: y = rounded float32 y; the following BatchNorm's statistics are those of the ROUNDED y
    assert same_bits(res[(1, 'y')], res[(0, 'y')])
    yr = res[(1, 'y')].double().view(T, -1, Cc)
    st = res[(1, 'st')].double().view(4, T, Cc)
    assert torch.allclose(st[0], yr.mean(1), rtol=1e-6, atol=1e-6)
    assert torch.allclose(st[1], 1.0 / torch.sqrt(yr.var(1, unbiased=False) + 1e-3), rtol=1e-5)
    # backward from a COMMON state (the rounded y and its statistics): activations rounded, everything else identical
    yb2, post = res[(1, 'y')], res[(1, 'st')]
    out = {}
    for at, X, Y, DO, dt in ((0, xb.float(), yb2.float(), dob.float(), torch.float32), (1, xb, yb2, dob, BF)):
        with storage(lib, at):
            ws = torch.zeros(int(lib.cdrl_dwconv_bn_workspace_doubles(T, B, H, W, Cc, stride)), dtype=torch.float64, device=DEV)
            dx = torch.zeros((N, H, W, Cc), dtype=dt, device=DEV)
            dw, db = torch.zeros((3, 3, Cc, 1), device=DEV), torch.zeros(Cc, device=DEV)
            vecs = [torch.zeros(Cc, device=DEV) for _ in range(4)]
            coefs = [torch.zeros(3 * T * Cc, device=DEV) for _ in range(2)]
            _lib.check(lib.cdrl_dwconv_bn_bwd(P(X), P(pre_stats), P(DO), P(Y), P(post), P(w), T, B, H, W, Cc, stride, P(dx), P(dw), P(db),
                                              P(vecs[0]), P(vecs[1]), P(coefs[0]), P(vecs[2]), P(vecs[3]), P(coefs[1]), P(ws), S()))
            out[at] = (dx, dw, db, vecs, coefs)
    o0, o1 = out[0], out[1]
    assert torch.equal(o0[1], o1[1]) and torch.equal(o0[2], o1[2])                  # filter / bias gradients
    assert torch.equal(o0[3][0], o1[3][0]) and torch.equal(o0[3][1], o1[3][1]) and torch.equal(o0[4][0], o1[4][0])
    if pre:
        # dz1 (the masked gradient at the pre-BN's output) is the kernel's activation output; the op wrapper then applies the
        # pre-BN backward IN PLACE on it (reads the stored dz1): float32 vs bf16 storage differ by that one rounding
        assert torch.equal(o0[3][2], o1[3][2]) and torch.equal(o0[3][3], o1[3][3])  # BN1 sums come from the unrounded registers
        e = (o1[0].float() - o0[0]).abs().max().item() / o0[0].abs().max().item()
        assert e < 2.0 ** -7, e
    else:
        assert same_bits(o1[0], o0[0])


@pytest.mark.parametrize('G,Mg,K,N,pro,epi,bt', [(4, 330, 58, 58, 1, 1, 0), (4, 1000, 116, 116, 1, 1, 0), (2, 515, 24, 58, 0, 1, 0),
                                                  (4, 257, 116, 116, 0, 2, 1), (4, 120, 232, 232, 1, 1, 0), (1, 77, 232, 232, 0, 2, 1),
                                                  (4, 96, 58, 24, 0, 0, 1), (3, 200, 116, 58, 0, 0, 0)])
def test_pwconv_fused_bf16_storage(lib, G, Mg, K, N, pro, epi, bt):
    rng = np.random.default_rng(G * Mg + K + N)
    M = G * Mg
    lda, coff = K + 6, 2
    ab = dev(rng.standard_normal((M, lda)), BF)
    w = dev(rng.standard_normal((N, K) if bt else (K, N)) / np.sqrt(K))
    bias = dev(rng.standard_normal(N))
    PS, ES = dev(rng.uniform(0.5, 1.5, (4, G, K))), dev(rng.uniform(0.5, 1.5, (4, G, N)))
    eyb = dev(rng.standard_normal((M, N)), BF)
    c0b = dev(rng.standard_normal((M, N + 4)), BF)                          # previous content (accumulate variant)
    sbk, sbn = (1, K) if bt else (N, 1)
    nb = int(lib.cdrl_pwconv_fused_partial_rows(G, Mg, N, K))
    wp = torch.zeros(int(lib.cdrl_pwconv_pack_elems(N, K)), device=DEV)
    _lib.check(lib.cdrl_pwconv_pack(P(w), K, N, sbk, sbn, P(wp), 1, S()))
    res = {}
    for at, A, EY, C0 in ((0, ab.float(), eyb.float(), c0b.float()), (1, ab, eyb, c0b)):
        out = C0.clone()
        part = torch.zeros((G, nb, 2, N), dtype=torch.float64, device=DEV)
        with storage(lib, at):
            _lib.check(lib.cdrl_pwconv_fused_packed(P(A), lda, coff, P(PS) if pro else None, P(w), sbk, sbn, None if bt else P(bias), P(out),
                                                    N + 4, 2, 1 if bt else 0, G, Mg, N, K, epi, P(EY), P(ES), P(part), P(wp), 1, S()))
        res[at] = (out, part.sum(1))
    o0, o1 = res[0][0], res[1][0]
    assert same_bits(o1[:, 2:2 + N], o0[:, 2:2 + N])
    assert torch.equal(o1[:, :2], c0b[:, :2]) and torch.equal(o1[:, 2 + N:], c0b[:, 2 + N:])       # neighbours untouched
    if epi == 1:        # statistics of the STORED values
        r = o1[:, 2:2 + N].double().view(G, Mg, N)
        assert torch.allclose(res[1][1][:, 0], r.sum(1), rtol=1e-9, atol=1e-6)
        assert torch.allclose(res[1][1][:, 1], (r * r).sum(1), rtol=1e-9, atol=1e-6)
    elif epi == 2:      # BN-backward sums: taken from the float32 registers, identical in both storage modes
        assert torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize('G,Mg,K,N,relu,shuffle,xpro', [(4, 330, 58, 58, 1, 1, 0), (4, 96, 116, 116, 1, 1, 1), (2, 500, 24, 58, 1, 0, 0),
                                                        (4, 257, 58, 24, 0, 0, 1), (4, 120, 232, 232, 1, 1, 1)])
def test_pwconv_bn_bwd_bf16_storage(lib, G, Mg, K, N, relu, shuffle, xpro):
    rng = np.random.default_rng(G * Mg + K + N)
    M = G * Mg
    xb = dev(rng.standard_normal((M, K)), BF)
    yb = dev(rng.standard_normal((M, N)) * 1.3 + 0.2, BF)
    w = dev(rng.standard_normal((K, N)) / np.sqrt(K))
    xst = dev(np.stack([np.zeros((G, K)), np.ones((G, K)), rng.uniform(0.5, 1.5, (G, K)), rng.uniform(-0.5, 0.5, (G, K))]))
    gam, bet = dev(rng.uniform(0.5, 1.5, N)), dev(rng.uniform(1.0, 3.0, N))
    ctot, coff = (2 * N, N) if shuffle else (N, 0)
    dob = dev(rng.standard_normal((M, ctot)), BF)
    dx0b = dev(rng.standard_normal((M, K + 4)), BF)
    wtp = torch.zeros(int(lib.cdrl_pwconv_pack_elems(K, N)), device=DEV)
    _lib.check(lib.cdrl_pwconv_pack(P(w), N, K, 1, N, P(wtp), 1, S()))
    res = {}
    for at, X, Y, DO, DX0, dt in ((0, xb.float(), yb.float(), dob.float(), dx0b.float(), torch.float32), (1, xb, yb, dob, dx0b, BF)):
        with storage(lib, at):
            stats = torch.zeros(4 * G * N, device=DEV)
            tmp = torch.zeros((M, N), dtype=dt, device=DEV)
            ws0 = torch.zeros(G * 256 * 2 * N, dtype=torch.float64, device=DEV)
            mm, mv = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
            _lib.check(lib.cdrl_bn_train_fwd(P(Y), G, Mg, N, P(gam), P(bet), P(mm), P(mv), 1, relu, P(tmp), N, 0, 0, P(stats), P(ws0), S()))
            ws = torch.zeros(int(lib.cdrl_pwconv_bn_bwd_workspace_bytes(G, Mg, N, K)), dtype=torch.uint8, device=DEV)
            dg, dbt, coef = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV), torch.zeros(3 * G * N, device=DEV)
            dx = DX0.clone()
            dw, db = torch.zeros((K, N), device=DEV), torch.zeros(N, device=DEV)
            _lib.check(lib.cdrl_pwconv_bn_bwd_packed(P(DO), ctot, coff, ctot if shuffle else 0, relu, P(Y), P(stats), P(X), K, 0,
                                                     P(xst) if xpro else None, P(w), G, Mg, N, K, P(dg), P(dbt), P(coef), P(dx), K + 4, 2, 1,
                                                     P(dw), P(db), P(ws), P(wtp), 1, S()))
        res[at] = (dx, dg, dbt, coef, dw, db)
    r0, r1 = res[0], res[1]
    assert same_bits(r1[0][:, 2:2 + K], r0[0][:, 2:2 + K])                  # input gradient (accumulated onto the old content)
    assert torch.equal(r1[0][:, :2], dx0b[:, :2]) and torch.equal(r1[0][:, 2 + K:], dx0b[:, 2 + K:])
    for i in (1, 2, 3, 4, 5):                                               # dgamma, dbeta, coefficients, filter and bias gradients
        assert torch.equal(r0[i], r1[i]), (i, float((r0[i] - r1[i]).abs().max()), float(r0[i].abs().max()))


@pytest.mark.parametrize('at', [1, 0])
def test_fused_backward_conv_is_reproducible_next_to_the_filter_gradient_gemm(lib, at):
    """Co-execution: the fused BN-backward + 1x1 backward-data GEMM (stage-2 shape, K = N = 232, one workgroup per CU) on one stream
    while another stream runs the LDS filter-gradient GEMM with several column blocks (the head conv's shape) -- what the engine's
    main and side streams do.  Every repetition must reproduce the result of the kernel running alone, bit for bit.  (Round 3: with
    the two-sided ReLU6 mask `z > 0 && z < 6` about 70 of 49152 rows per launch came out with masked elements unmasked in the top
    lanes of a wave -- never alone; tools/det_co.py, DESIGN.md "What round 3 found".)"""
    rng = np.random.default_rng(0)
    dt = BF if at else torch.float32
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    SA, SB = C.c_void_p(sA.cuda_stream), C.c_void_p(sB.cuda_stream)
    G, Mg, K, N = 4, 12288, 232, 232
    M = G * Mg
    xb, yb = dev(rng.standard_normal((M, K)), BF).to(dt), dev(rng.standard_normal((M, N)) * 1.3 + 0.2, BF).to(dt)
    w = dev(rng.standard_normal((K, N)) / np.sqrt(K))
    gam, bet = dev(rng.uniform(0.5, 1.5, N)), dev(rng.uniform(1.0, 3.0, N))
    ctot, coff = 2 * N, N
    dob = dev(rng.standard_normal((M, ctot)), BF).to(dt)
    M2, K2, N2 = 49152, 464, 768
    a2, d2 = torch.randn(M2, K2, device=DEV).to(BF), torch.randn(M2, N2, device=DEV).to(BF)
    with storage(lib, 1):
        ws2 = torch.zeros(int(lib.cdrl_gemm_tn_workspace_elems(M2, N2, K2)), device=DEV)
    out2 = torch.zeros(K2, N2, device=DEV)
    wtp = torch.zeros(int(lib.cdrl_pwconv_pack_elems(K, N)), device=DEV)
    stats = torch.zeros(4 * G * N, device=DEV)
    with storage(lib, at):
        _lib.check(lib.cdrl_pwconv_pack(P(w), N, K, 1, N, P(wtp), 1 if at else 0, SA))
        tmp = torch.zeros((M, N), dtype=dt, device=DEV)
        ws0 = torch.zeros(G * 256 * 2 * N, dtype=torch.float64, device=DEV)
        mm, mv = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
        _lib.check(lib.cdrl_bn_train_fwd(P(yb), G, Mg, N, P(gam), P(bet), P(mm), P(mv), 1, 1, P(tmp), N, 0, 0, P(stats), P(ws0), SA))
    torch.cuda.synchronize()
    outs = []
    for rep in range(7):
        ws = torch.zeros(int(lib.cdrl_pwconv_bn_bwd_workspace_bytes(G, Mg, N, K)), dtype=torch.uint8, device=DEV)
        dg, dbt, coef = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV), torch.zeros(3 * G * N, device=DEV)
        dx = torch.zeros((M, K), dtype=dt, device=DEV)
        dw, db = torch.zeros((K, N), device=DEV), torch.zeros(N, device=DEV)
        torch.cuda.synchronize()
        if rep > 0:                                         # repetition 0 runs alone
            with storage(lib, 1):
                for _ in range(3):
                    _lib.check(lib.cdrl_gemm_tn(P(a2), K2, 0, P(d2), N2, 0, P(out2), M2, N2, K2, P(ws2), 0, SB))
        with storage(lib, at):
            _lib.check(lib.cdrl_pwconv_bn_bwd_packed(P(dob), ctot, coff, ctot, 1, P(yb), P(stats), P(xb), K, 0, None, P(w), G, Mg, N, K, P(dg),
                                                     P(dbt), P(coef), P(dx), K, 0, 0, P(dw), P(db), P(ws), P(wtp), 1 if at else 0, SA))
        torch.cuda.synchronize()
        outs.append((dx, db, dw, dg, out2.clone()))
    for rep in range(1, 7):
        for i, name in enumerate(('dx', 'db', 'dw', 'dgamma', 'co-runner')):
            if name == 'co-runner' and rep == 1:
                continue
            ref = outs[0][i] if name != 'co-runner' else outs[1][i]
            assert torch.equal(ref, outs[rep][i]), (name, rep, int((ref.float() != outs[rep][i].float()).sum()))


@pytest.mark.parametrize('M,K,N', [(4096, 116, 116), (1000, 232, 232), (777, 464, 768), (640, 24, 24), (5000, 24, 58), (3001, 58, 58), (1000, 58, 24), (777, 16, 64)])
def test_gemm_tn_and_x3_bf16_storage(lib, M, K, N):
    rng = np.random.default_rng(M + K + N)
    ab, db_ = dev(rng.standard_normal((M, K)), BF), dev(rng.standard_normal((M, N)), BF)
    w = dev(rng.standard_normal((K, N)) / np.sqrt(K))
    bias = dev(rng.standard_normal(N))
    ws = torch.zeros(int(lib.cdrl_gemm_tn_workspace_elems(M, N, K)), device=DEV)
    # filter gradient: float32 output from bf16 tensors == the bf16-operand product of the widened tensors (bit for bit) == float64 product
    out = torch.zeros((K, N), device=DEV)
    with storage(lib, 1):
        _lib.check(lib.cdrl_gemm_tn(P(ab), K, 0, P(db_), N, 0, P(out), M, N, K, P(ws), 0, S()))
    ref = ab.double().T @ db_.double()
    assert float((out.double() - ref).abs().max() / ref.abs().max()) < 1e-5
    if K % 4 != 0:
        return                                                              # (gemm_x3 takes 8-byte aligned rows)
    # general GEMM (head conv / shortcut convs): C = A W + bias, rounded on store
    wp = torch.zeros(int(lib.cdrl_gemm_x3_packed_bytes(N, K)), dtype=torch.uint8, device=DEV)
    _lib.check(lib.cdrl_gemm_x3_pack(P(w), K, N, N, 1, P(wp), S()))
    cb = torch.zeros((M, N), dtype=BF, device=DEV)
    with storage(lib, 1):
        _lib.check(lib.cdrl_gemm_x3(P(ab), K, 0, P(wp), P(bias), P(cb), N, 0, M, N, K, 0, S()))
    ref = ab.double() @ w.to(BF).double() + bias.double()
    err = (cb.double() - ref).abs()
    assert bool((err <= ref.abs() * 2.0 ** -8 + 1e-3).all()), float(err.max())


@pytest.mark.parametrize('B,T,H,W', [(3, 4, 41, 58), (2, 2, 90, 120)])
def test_stem_block_bf16_storage(lib, B, T, H, W):
    """Fused BN + ReLU6 + max-pool forward and the stem block's backward from the pooled gradient, bf16 y / pool / pooled gradient."""
    Cc, N = 24, B * T
    rng = np.random.default_rng(B * H + W)
    x = dev(rng.uniform(0.0, 1.0, (B, T, H, W, 3)))
    w, b = dev(rng.standard_normal((3, 3, 3, Cc)) * 0.4), dev(rng.standard_normal(Cc))
    gamma = rng.uniform(0.5, 1.5, Cc)
    gamma[::5] *= -1.0
    Gm, Bt = dev(gamma), dev(rng.uniform(0.5, 2.5, Cc))
    Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    Hp, Wp = -(-Ho // 2), -(-Wo // 2)
    y32 = torch.zeros((N, Ho, Wo, Cc), device=DEV)
    _lib.check(lib.cdrl_stem_fwd(P(x), P(w), P(b), P(y32), B, T, H, W, Cc, S()))
    yb = y32.to(BF)
    dpb = dev(rng.standard_normal((N, Hp, Wp, Cc)), BF)
    res = {}
    for at, Y, DP, dt in ((0, yb.float(), dpb.float(), torch.float32), (1, yb, dpb, BF)):
        with storage(lib, at):
            MM, MV = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
            stats = torch.zeros(4 * T * Cc, device=DEV)
            ws0 = torch.zeros(T * 256 * 2 * Cc, dtype=torch.float64, device=DEV)
            scratch = torch.zeros((N * Ho * Wo, Cc), dtype=dt, device=DEV)
            _lib.check(lib.cdrl_bn_train_fwd(P(Y), T, B * Ho * Wo, Cc, P(Gm), P(Bt), P(MM), P(MV), 1, 1, P(scratch), Cc, 0, 0, P(stats), P(ws0), S()))
            pool = torch.zeros((N, Hp, Wp, Cc), dtype=dt, device=DEV)
            am = torch.zeros((N, Hp, Wp, Cc), dtype=torch.uint8, device=DEV)
            _lib.check(lib.cdrl_maxpool_bn_fwd(P(Y), P(stats), T, B, P(pool), P(am), N, Ho, Wo, Cc, S()))
            ws = torch.zeros(int(lib.cdrl_stem_block_bwd_workspace_doubles(B, T, H, W, Cc)), dtype=torch.float64, device=DEV)
            dg, dbt, coef = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV), torch.zeros(3 * T * Cc, device=DEV)
            dw, db = torch.zeros((3, 3, 3, Cc), device=DEV), torch.zeros(Cc, device=DEV)
            _lib.check(lib.cdrl_stem_block_bwd(P(x), P(Y), P(stats), P(am), P(DP), B, T, H, W, Cc, P(dg), P(dbt), P(coef), P(dw), P(db), P(ws), S()))
        res[at] = (pool, am, stats, dg, dbt, coef, dw, db)
    r0, r1 = res[0], res[1]
    assert same_bits(r1[0], r0[0]) and torch.equal(r0[1], r1[1])
    for i in (2, 3, 4, 5, 6, 7):
        assert torch.equal(r0[i], r1[i]), i


# ---------------------------------------------------------------------------------------------------------------------
# engine level
# ---------------------------------------------------------------------------------------------------------------------
def _flat(views, names):
    return np.concatenate([views[n].detach().cpu().numpy().astype(np.float64).ravel() for n in names])


def _cos(a, b):
    return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))


@pytest.mark.parametrize('B,H,W,A', [(64, 48, 64, 2), (32, 90, 120, 3), (16, 41, 58, 2)])
def test_bf16_storage_engine_vs_bf16_operand_engine(B, H, W, A):
    """Same weights, same batch: the bf16-storage engine against the bf16-operand engine (identical arithmetic, float32 tensors).
    Storing an activation as bf16 perturbs it by <= 2^-9 relative, the same size as the operand rounding the other engine already
    applies inside every 1x1 convolution, so the two agree the way two bf16 implementations do: FORWARD quantities closely
    (loss 5e-2, Beta parameters 3e-1 of their scale, worst element), gradients in
    direction and scale (cosine; see the discussion there).  Inference (predict) is compared as well."""
    from tests.util import make_pair, make_batches, to_dev, rel_err, is_zero_gradient
    _, es = make_pair(B, H, W, seed=5, A=A, compute='bf16s')
    _, eo = make_pair(B, H, W, seed=5, A=A, compute='bf16')
    pol, val = make_batches(B, H, W, seed=5, A=A, faithful=True)
    dpol, dval = to_dev(pol), to_dev(val)
    ps, po = es.predict(dpol['states']), eo.predict(dpol['states'])
    for k in ('alpha', 'beta', 'value'):
        assert rel_err(ps[k].cpu().numpy(), po[k].cpu().numpy()) < 3e-1, k
    es.policy_forward_backward(dpol)
    eo.policy_forward_backward(dpol)
    ls, lo = es.metrics('policy')['loss'], eo.metrics('policy')['loss']
    assert np.isfinite(ls) and abs(ls - lo) <= 5e-2 * max(1.0, abs(lo)), (ls, lo)       # measured 0.6e-2 .. 2.3e-2
    axs, axo = es.buffer(_lib.BUF_AUX_P, (B, 4, A)).cpu().numpy(), eo.buffer(_lib.BUF_AUX_P, (B, 4, A)).cpu().numpy()
    for i, k in enumerate(('alpha', 'beta')):
        assert rel_err(axs[:, i], axo[:, i]) <= 3e-1, k         # measured 0.12 .. 0.18 (worst element over the minibatch, relative to the largest)
    gs, go = es.grad_views('trunk'), eo.grad_views('trunk')
    names = [n for n in gs if not is_zero_gradient(n)]
    rep = {}
    for grp, ns in (('tower', [n for n in names if n.startswith('img.')]), ('tail', [n for n in names if not n.startswith('img.')])):
        a, b = _flat(gs, ns), _flat(go, ns)
        assert np.all(np.isfinite(a))
        rep[grp] = dict(cos=_cos(a, b), norm_ratio=float(np.linalg.norm(a) / np.linalg.norm(b)))
        assert 0.25 <= rep[grp]['norm_ratio'] <= 4.0, (grp, rep)
    # measured: tail 0.77 .. 0.8, tower 0.35 (the operand engine itself sits at 0.95 / 0.65 from the float32 engine): every stored
    # tensor is one more 2^-9 perturbation in front of ~50 train-mode BatchNorms
    assert rep['tail']['cos'] > 0.6 and rep['tower']['cos'] > 0.2, rep
    hs, ho = es.grad_views('policy'), eo.grad_views('policy')
    hn = [n for n in hs if not is_zero_gradient(n)]
    assert _cos(_flat(hs, hn), _flat(ho, hn)) > 0.7           # measured 0.81 .. 0.9
    es.value_forward_backward(dval)
    eo.value_forward_backward(dval)
    vs, vo = es.metrics('value')['loss'], eo.metrics('value')['loss']
    assert abs(vs - vo) <= 5e-2 * max(1.0, abs(vo)), (vs, vo)
    import json
    import os
    os.makedirs('gpurun_out', exist_ok=True)
    rep.update(loss_storage=ls, loss_operand=lo, value_loss_storage=vs, value_loss_operand=vo)
    json.dump(rep, open(f'gpurun_out/parity_report_bf16_storage_B{B}_{H}x{W}.json', 'w'), indent=1)


def test_bf16_storage_training_tracks_float32():
    """Loss-curve criterion: 12 update-steps (policy + value, re-sampled loss on the same Philox stream) from
    identical weights on the bf16-storage and the float32 engine; each loss within 15 % of the float32 curve's scale at every
    step, the value loss decreasing and the policy objective minimised on both."""
    from tests.util import make_pair, make_batches, to_dev
    B, H, W = 32, 48, 64
    _, e16 = make_pair(B, H, W, seed=9, compute='bf16s')
    _, e32 = make_pair(B, H, W, seed=9)
    pol, val = make_batches(B, H, W, seed=9)
    dpol, dval = to_dev(pol), to_dev(val)
    hist = {16: [], 32: []}
    for step in range(12):
        for tag, e in ((16, e16), (32, e32)):
            e.policy_forward_backward_resample(dpol, 7, step)
            lp = e.metrics('policy')['loss']
            e.policy_apply()
            e.value_forward_backward(dval)
            lv = e.metrics('value')['loss']
            e.value_apply()
            hist[tag].append((lp, lv))
    h16, h32 = np.array(hist[16]), np.array(hist[32])
    assert np.all(np.isfinite(h16))
    for j in range(2):
        scale = max(np.abs(h32[:, j]).max(), 1e-6)
        assert np.abs(h16[:, j] - h32[:, j]).max() <= 0.15 * scale, (j, h16[:, j], h32[:, j])
    assert h32[-1, 1] < h32[0, 1] and h16[-1, 1] < h16[0, 1]
    assert h16[-1, 0] < 0.2 * h16[0, 0]


def test_bf16_storage_converges_like_float32_over_200_steps():
    """configs[2]'s consequence: the bf16-storage tower gradient points 0.3-0.4 (cosine) away from the float32
    one (the rounding of the stored ACTIVATIONS, tests/test_oracle_bf16_ablation.py) -- does it train?  200 update-steps (re-sampled policy
    loss on one Philox stream + value step, lr 3e-4) on the 'signal' minibatch of test_config3_three_engines_at_batch_1024 (a per-sample
    brightness offset the advantages / returns follow) at the configuration's own batch, B = 1024, 4 x 90 x 120 x 3, from identical
    weights: the float32 and the bf16-storage engine must reach the same losses.  Curves -> gpurun_out/c3_convergence_B1024.json."""
    import json
    import os
    from carla_driving_rl_agent_amd.engine import LearnerEngine
    from carla_driving_rl_agent_amd.init import init_engine_parameters
    from carla_driving_rl_agent_amd import synthetic
    B, T, H, W, STEPS = 1024, 4, 90, 120, 200
    r = synthetic.make_rollout(B, T=T, H=H, W=W, seed=7)
    rng = np.random.default_rng(1)
    bright = rng.uniform(-0.3, 0.3, B).astype(np.float32)
    r['states']['state_image'] = np.clip(0.5 * r['states']['state_image'] + 0.25 + bright[:, None, None, None, None], 0.0, 1.0).astype(np.float32)
    adv_np = (bright / 0.3 * 1.5 + 0.2 * rng.standard_normal(B)).astype(np.float32)
    ret_np = np.stack([bright / 0.3, np.abs(bright) / 0.3], axis=1).astype(np.float32)
    states = {k: torch.as_tensor(v).cuda() for k, v in r['states'].items()}
    pol = dict(states=states, advantages=torch.as_tensor(adv_np).cuda(), old_log_prob=torch.as_tensor(r['old_log_prob']).cuda(),
               speed=(torch.as_tensor(r['speed'][:, 0]) / 100.0).cuda().contiguous(),
               similarity=torch.as_tensor(r['similarity'][:, 0]).cuda().contiguous(), u=torch.as_tensor(r['action']).cuda(),
               du_da=None, du_db=None)
    val = dict(states=states, returns=torch.as_tensor(ret_np).cuda(), speed=pol['speed'], similarity=pol['similarity'])
    curves = {}
    for compute in ('f32', 'bf16s'):
        eng = LearnerEngine(B, device=DEV, T=T, H=H, W=W, compute=compute)
        init_engine_parameters(eng, seed=42)
        hist = []
        for step in range(STEPS):
            eng.policy_forward_backward_resample(pol, seed=11, offset=step + 1)
            lp = eng.metrics('policy')['loss']
            eng.policy_apply()
            eng.value_forward_backward(val)
            lv = eng.metrics('value')['loss']
            eng.value_apply()
            hist.append((lp, lv))
        curves[compute] = np.asarray(hist, dtype=np.float64)
        del eng
        torch.cuda.empty_cache()
    a, b = curves['f32'], curves['bf16s']
    assert np.all(np.isfinite(a)) and np.all(np.isfinite(b))
    smooth = lambda x: np.convolve(x, np.ones(10) / 10.0, mode='valid')
    rep = dict(config=dict(B=B, T=T, H=H, W=W, steps=STEPS, rollout='signal', lr=3e-4), first=dict(f32=a[0].tolist(), bf16s=b[0].tolist()),
               last10_mean=dict(f32=a[-10:].mean(axis=0).tolist(), bf16s=b[-10:].mean(axis=0).tolist()), worst_smoothed_gap={},
               curve_every_5=dict(f32=a[::5].tolist(), bf16s=b[::5].tolist()))
    for j, name in enumerate(('policy', 'value')):
        sa, sb = smooth(a[:, j]), smooth(b[:, j])
        rng_a = max(sa.max() - sa.min(), 1e-6)
        rep['worst_smoothed_gap'][name] = float(np.abs(sa - sb).max() / rng_a)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(rep, open('gpurun_out/c3_convergence_B1024.json', 'w'), indent=1)
    print('[configs[2] convergence]', json.dumps({k: rep[k] for k in ('first', 'last10_mean', 'worst_smoothed_gap')}))
    # both train: the value loss falls by more than half, the policy objective is minimised
    for c in (a, b):
        assert c[-10:, 1].mean() < 0.5 * c[:5, 1].mean(), rep['last10_mean']
        assert c[-10:, 0].mean() < c[:5, 0].mean(), rep['last10_mean']
    # ... to the same place, along the same curve.  Measured (profiles/r05_c3_convergence_B1024.json): policy 0.894 -> 0.0162 (float32) /
    # 0.889 -> 0.0186 (bf16 storage); value 0.3269 -> 0.00047 / 0.3280 -> 0.0014 -- both remove > 99.5 % of the value loss, the bf16-storage
    # run lags in the fast phase (step 20: 0.0102 vs 0.0030) and keeps a 3x larger residual after 200 steps; the 10-step moving averages are
    # never further apart than 0.12 (policy) / 0.18 (value) of the float32 curve's range.  Gates: final losses within 2 % of the float32
    # run's total movement, moving averages within 25 % of its range, residual value loss below 1 % of the initial one.
    for j, name in enumerate(('policy', 'value')):
        move = max(abs(a[:5, j].mean() - a[-10:, j].mean()), 1e-6)
        assert abs(a[-10:, j].mean() - b[-10:, j].mean()) <= 0.02 * move, (name, rep['last10_mean'])
        assert rep['worst_smoothed_gap'][name] <= 0.25, rep['worst_smoothed_gap']
    assert b[-10:, 1].mean() < 0.01 * b[0, 1], rep['last10_mean']


def test_bf16_storage_mode_is_deterministic_and_smaller():
    from tests.util import make_pair, make_batches, to_dev
    B, H, W = 16, 90, 120
    pol, val = make_batches(B, H, W, seed=11)
    dpol, dval = to_dev(pol), to_dev(val)
    outs = []
    for _ in range(2):
        _, e = make_pair(B, H, W, seed=11, compute='bf16s')
        e.policy_forward_backward(dpol)
        lp = e.metrics('policy')['loss']
        gp = e.grads.clone()
        e.policy_apply()
        e.value_forward_backward(dval)
        outs.append((lp, e.metrics('value')['loss'], gp, e.grads.clone(), e.params.clone()))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
    for a, b in zip(outs[0][2:], outs[1][2:]):
        assert torch.equal(a, b)
    _, e32 = make_pair(B, H, W, seed=11)
    assert e.workspace.numel() * e.workspace.element_size() < 0.75 * e32.workspace.numel() * e32.workspace.element_size()


@pytest.mark.parametrize('compute', ['f32', 'bf16s'])
def test_config3_full_size_properties(compute):
    """BASELINE.json configs[2] at its own size -- B = 1024, T = 4, 90x120x3 -- in both storage modes, through the size-independent
    properties (the CPU oracle cannot run here): (1) bit-wise determinism over repeated policy / value passes, (2) exact linearity
    of every gradient in the data-parallel gradient scale, (3) invariance of loss and gradients under a permutation of the
    minibatch rows up to summation order, (4) equivariance of the trunk output."""
    from carla_driving_rl_agent_amd.engine import LearnerEngine
    from carla_driving_rl_agent_amd.init import init_engine_parameters
    from carla_driving_rl_agent_amd import synthetic
    B, T, H, W = 1024, 4, 90, 120
    eng = LearnerEngine(B, device=DEV, T=T, H=H, W=W, compute=compute)
    init_engine_parameters(eng, seed=42)
    r = synthetic.make_rollout(B, T=T, H=H, W=W, seed=7)
    states = {k: torch.as_tensor(v).cuda() for k, v in r['states'].items()}
    adv = torch.as_tensor(np.random.default_rng(1).standard_normal(B).astype(np.float32)).cuda()
    pol = dict(states=states, advantages=adv, old_log_prob=torch.as_tensor(r['old_log_prob']).cuda(),
               speed=(torch.as_tensor(r['speed'][:, 0]) / 100.0).cuda().contiguous(),
               similarity=torch.as_tensor(r['similarity'][:, 0]).cuda().contiguous(), u=torch.as_tensor(r['action']).cuda(),
               du_da=None, du_db=None)
    val = dict(states=states, returns=torch.as_tensor(np.random.default_rng(2).uniform(-1, 1, (B, 2)).astype(np.float32)).cuda(),
               speed=pol['speed'], similarity=pol['similarity'])
    moving = {k: v.clone() for k, v in eng.param_views('trunk').items() if 'moving' in k}

    def region(models):
        return torch.cat([v.reshape(-1) for m in models for v in eng.grad_views(m).values()]).clone()

    def grads(batch, scale=1.0):
        for k, v in eng.param_views('trunk').items():
            if 'moving' in k:
                v.copy_(moving[k])
        eng.policy_forward_backward(batch, grad_scale=scale)
        torch.cuda.synchronize()
        return region(('policy', 'trunk')), eng.metrics('policy')['loss']

    p1, l1 = grads(pol)
    assert torch.isfinite(p1).all() and np.isfinite(l1)
    eng.value_forward_backward(val)
    torch.cuda.synchronize()
    v1 = region(('trunk', 'value'))
    for _ in range(5):                                                      # (1) (repeated: cross-stream races are timing dependent)
        p2, l2 = grads(pol)
        assert torch.equal(p1, p2) and l1 == l2
        eng.value_forward_backward(val)
        torch.cuda.synchronize()
        assert torch.equal(v1, region(('trunk', 'value')))
    ph, _ = grads(pol, scale=0.5)
    if compute == 'f32':
        assert torch.equal(ph, p1 * 0.5)                                    # (2) power-of-two scale: exact in float32
    else:       # bf16-stored activation gradients: scaling by 0.5 is exact in bf16 too (no subnormals in range)
        assert torch.equal(ph, p1 * 0.5)
    perm = torch.as_tensor(np.random.default_rng(3).permutation(B)).cuda()
    ppol = {k: (v[perm].contiguous() if torch.is_tensor(v) else v) for k, v in pol.items() if k != 'states'}
    ppol['states'] = {k: v[perm].contiguous() for k, v in states.items()}
    gp, lp = grads(ppol)
    assert abs(lp - l1) < 1e-4 * max(1.0, abs(l1))
    worst = (gp - p1).abs().max().item() / p1.abs().max().item()
    # (3) identical decisions and roundings per element, only the reduction orders differ -> float32 summation noise
    assert worst < (2e-4 if compute == 'f32' else 2e-3), worst
    for k, v in eng.param_views('trunk').items():
        if 'moving' in k:
            v.copy_(moving[k])
    eng.trunk_forward_train(states)
    torch.cuda.synchronize()
    d0 = eng.buffer(0, (B, eng.cfg.dyn)).clone()
    eng.trunk_forward_train(ppol['states'])
    torch.cuda.synchronize()
    d1 = eng.buffer(0, (B, eng.cfg.dyn)).clone()
    assert torch.isfinite(d0).all()
    assert (d1 - d0[perm]).abs().max().item() < (1e-4 if compute == 'f32' else 2e-3) * max(1.0, d0.abs().max().item())     # (4)


# Gradient groups of the three-engine comparison at configs[2]'s own size
def _c3_group(name):
    if name.startswith('img.stem') or name.startswith('img.s0'):
        return 'tower: stem + stage 0'
    if name.startswith('img.s1'):
        return 'tower: stage 1'
    if name.startswith('img.'):
        return 'tower: stage 2 + head conv'
    if name.split('.')[0] in ('road', 'vehicle', 'navigation'):
        return 'feature nets'
    return 'GRUs' if name.startswith('gru_') else 'trunk tail'


# Thresholds = what was measured at B = 1024, 90x120 (profiles/r04_c3_three_engines_B1024_*.json), minus ~10 %; `cos` is the
# cosine of the concatenated gradient vectors of a group against the float32 engine, `ratio` the quotient of their norms.
# Two minibatches: 'noise' = the synthetic rollout as it is (iid-noise images, random advantages / returns) and 'signal' = the
# same rollout with a per-sample brightness offset in the images that the advantages / returns follow.  Measured (policy / value
# pass, noise | signal):
#   bf16 operands : tower 0.61 / 0.62 | 0.58 / 0.61 (stem + stage 0), 0.61 | 0.61 (stage 1), 0.74 | 0.76 (stage 2 + head conv);
#                   feature nets / GRUs / tail 0.94-0.98; heads 0.99; norm ratios 0.89-1.02; loss 2e-3 | 5e-3
#   bf16 storage  : tower 0.42 | 0.33 / 0.35, 0.42 | 0.41 / 0.43, 0.57 | 0.61; feature nets / GRUs / tail 0.83-0.95; heads 0.97-0.99;
#                   norm ratios 0.83-1.08; loss 1e-3 .. 3e-3
# i.e. the LOSSES and gradient NORMS of the three engines agree to a fraction of a percent / a few percent at this size, the
# DIRECTION of the tower's weight gradient does not: behind ~50 train-mode BatchNorms every weight gradient is the small residue of
# cancelling sums (mean and xhat-correlated parts removed at each layer), and a 2^-9 perturbation per stored activation moves it by
# an amount comparable to itself -- with a learnable signal in the batch as much as without.  It is a property of this network in
# bf16, not of the batch size (B <= 64: 0.35, tests above); what training needs is shown by test_bf16_storage_training_tracks_float32.
C3_GATES = {
    ('bf16', 'noise'): dict(loss=1e-2, dist=0.20, tower=(0.52, 0.55, 0.67), tail=0.90, heads=0.97),
    ('bf16s', 'noise'): dict(loss=1e-2, dist=0.36, tower=(0.29, 0.36, 0.51), tail=0.75, heads=0.94),
    ('bf16', 'signal'): dict(loss=1e-2, dist=0.20, tower=(0.52, 0.55, 0.67), tail=0.90, heads=0.97),
    ('bf16s', 'signal'): dict(loss=1e-2, dist=0.36, tower=(0.29, 0.36, 0.51), tail=0.75, heads=0.94),
}
C3_TOWER = ('tower: stem + stage 0', 'tower: stage 1', 'tower: stage 2 + head conv')


@pytest.mark.parametrize('scenario', ['noise', 'signal'])
def test_config3_three_engines_at_batch_1024(scenario):
    """configs[2] at ITS OWN batch (B = 1024, 4 x 90 x 120 x 3): the float32, the bf16-operand and the bf16-storage engine from
    identical weights on the identical minibatch -- loss, Beta parameters, values, and per-group gradient cosine / norm ratio
    against the float32 engine.  (the B <= 64 comparisons sit in the regime where a handful of decision flips
    dominate a 64-row BatchNorm; this is the size the configuration is quoted on, and it needs no CPU oracle.)"""
    import json
    import os
    from carla_driving_rl_agent_amd.engine import LearnerEngine
    from carla_driving_rl_agent_amd.init import init_engine_parameters
    from carla_driving_rl_agent_amd import synthetic
    from tests.util import is_zero_gradient, rel_err
    B, T, H, W, A = 1024, 4, 90, 120, 2
    r = synthetic.make_rollout(B, T=T, H=H, W=W, seed=7)
    rng = np.random.default_rng(1)
    if scenario == 'signal':
        bright = rng.uniform(-0.3, 0.3, B).astype(np.float32)                # per-sample brightness the targets follow
        img = np.clip(0.5 * r['states']['state_image'] + 0.25 + bright[:, None, None, None, None], 0.0, 1.0).astype(np.float32)
        r['states']['state_image'] = img
        adv_np = (bright / 0.3 * 1.5 + 0.2 * rng.standard_normal(B)).astype(np.float32)
        ret_np = np.stack([bright / 0.3, np.abs(bright) / 0.3], axis=1).astype(np.float32)
    else:
        adv_np = rng.standard_normal(B).astype(np.float32)
        ret_np = np.random.default_rng(2).uniform(-1, 1, (B, 2)).astype(np.float32)
    states = {k: torch.as_tensor(v).cuda() for k, v in r['states'].items()}
    pol = dict(states=states, advantages=torch.as_tensor(adv_np).cuda(), old_log_prob=torch.as_tensor(r['old_log_prob']).cuda(),
               speed=(torch.as_tensor(r['speed'][:, 0]) / 100.0).cuda().contiguous(),
               similarity=torch.as_tensor(r['similarity'][:, 0]).cuda().contiguous(), u=torch.as_tensor(r['action']).cuda(),
               du_da=None, du_db=None)
    val = dict(states=states, returns=torch.as_tensor(ret_np).cuda(), speed=pol['speed'], similarity=pol['similarity'])
    res = {}
    for compute in ('f32', 'bf16', 'bf16s'):
        eng = LearnerEngine(B, device=DEV, T=T, H=H, W=W, compute=compute)
        init_engine_parameters(eng, seed=42)
        eng.policy_forward_backward(pol)
        torch.cuda.synchronize()
        out = dict(loss=eng.metrics('policy')['loss'], dist=eng.buffer(_lib.BUF_AUX_P, (B, 4, A)).cpu().numpy().copy(),
                   trunk_p={k: v.cpu().numpy().astype(np.float64) for k, v in eng.grad_views('trunk').items()},
                   policy={k: v.cpu().numpy().astype(np.float64) for k, v in eng.grad_views('policy').items()})
        eng.value_forward_backward(val)
        torch.cuda.synchronize()
        out.update(vloss=eng.metrics('value')['loss'], values=eng.buffer(_lib.BUF_AUX_V, (B, 2)).cpu().numpy().copy(),
                   trunk_v={k: v.cpu().numpy().astype(np.float64) for k, v in eng.grad_views('trunk').items()},
                   value={k: v.cpu().numpy().astype(np.float64) for k, v in eng.grad_views('value').items()})
        res[compute] = out
        del eng
        torch.cuda.empty_cache()
    ref = res['f32']
    report = {}
    for compute in ('bf16', 'bf16s'):
        o = res[compute]
        rep = dict(loss=abs(o['loss'] - ref['loss']) / max(1.0, abs(ref['loss'])), value_loss=abs(o['vloss'] - ref['vloss']) / max(1.0, abs(ref['vloss'])),
                   alpha=rel_err(o['dist'][:, 0], ref['dist'][:, 0]), beta=rel_err(o['dist'][:, 1], ref['dist'][:, 1]),
                   values=rel_err(o['values'], ref['values']), policy_pass={}, value_pass={})
        for key, trunk, head, hname in (('policy_pass', 'trunk_p', 'policy', 'policy head'), ('value_pass', 'trunk_v', 'value', 'value head')):
            groups = {}
            for name in ref[trunk]:
                if not is_zero_gradient(name):
                    groups.setdefault(_c3_group(name), []).append((trunk, name))
            groups[hname] = [(head, n) for n in ref[head] if not is_zero_gradient(n)]
            for grp, items in groups.items():
                a = np.concatenate([o[m][n].ravel() for m, n in items])
                b = np.concatenate([ref[m][n].ravel() for m, n in items])
                assert np.all(np.isfinite(a)), (compute, grp)
                rep[key][grp] = dict(cos=_cos(a, b), ratio=float(np.linalg.norm(a) / np.linalg.norm(b)))
        report[compute] = rep
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(report, open(f'gpurun_out/c3_three_engines_B1024_{scenario}.json', 'w'), indent=1)
    print(f'[configs[2] at B = 1024, {scenario}] vs the float32 engine:', json.dumps(report))
    for compute in ('bf16', 'bf16s'):
        rep, gate = report[compute], C3_GATES[(compute, scenario)]
        assert rep['loss'] <= gate['loss'] and rep['value_loss'] <= gate['loss'], (compute, rep)
        assert max(rep['alpha'], rep['beta'], rep['values']) <= gate['dist'], (compute, rep)
        for key in ('policy_pass', 'value_pass'):
            for grp, g in rep[key].items():
                lo = gate['tower'][C3_TOWER.index(grp)] if grp in C3_TOWER else (gate['heads'] if grp.endswith('head') else gate['tail'])
                assert g['cos'] >= lo, (compute, key, grp, g)
                assert 0.75 <= g['ratio'] <= 1.33, (compute, key, grp, g)


@pytest.mark.parametrize('B,H,W', [(64, 48, 64)])
def test_bf16_storage_engine_vs_oracle_with_the_storage_rule(B, H, W):
    """END-TO-END HIP vs ORACLE for configs[2]: the float64 oracle with the bf16-storage contract restated
    (oracle/model.py::_st: round-to-nearest-even at every tensor the engine stores, forward and gradient; bf16 MFMA operands in the
    1x1 convolutions), evaluated on the engine's own ReLU6 / max-pool decisions.  The rule itself is validated unit by unit at bf16
    rounding level by test_every_unit_backward_against_the_oracle_locally.  END TO END the two cannot agree better than two bf16
    implementations of this network do: an element whose float32 and float64 pre-rounding values straddle a bf16 rounding boundary
    (~5e-5 of all elements) differs by one bf16 ulp, and ~50 train-mode BatchNorms amplify that a thousandfold (the float32 engine
    meets the float64 oracle at 1e-4 = 1e-7 x 1e3).  Measured at B = 64, 48x64: loss 1e-2, alpha / beta 8-9 %, per-tensor gradient
    error median 0.16-0.26 in every group -- the same distance as between the bf16-operand engine and ITS rule oracle (round 2:
    4 % / 0.5 l2), i.e. rounding chaos, not wiring.  The gates only keep that level from drifting."""
    import json
    import os
    from oracle import model as OM
    from tests.util import make_pair, make_batches, to_dev, oracle_batch, rel_err, is_zero_gradient, engine_decisions
    A = 2
    oracle, eng = make_pair(B, H, W, seed=5, A=A, compute='bf16s', with64=True)
    o64 = oracle.o64
    pol, val = make_batches(B, H, W, seed=5, A=A, faithful=True)
    dpol, dval = to_dev(pol), to_dev(val)

    def ruled(fn, batch):
        OM.DEC.items = engine_decisions(eng, oracle.cfg)
        OM.PW_BF16_OPERANDS = OM.BF16_STORAGE = True
        try:
            OM.DEC.start('replay')
            out = fn(batch)
            assert OM.DEC.cursor == len(OM.DEC.items)
            return out
        finally:
            OM.DEC.start('off')
            OM.PW_BF16_OPERANDS = OM.BF16_STORAGE = False

    report = {}
    for kind in ('policy', 'value'):
        if kind == 'policy':
            eng.policy_forward_backward(dpol)
            loss, gh, gt, aux = ruled(o64.policy_grads, oracle_batch(pol))
            ax = eng.buffer(_lib.BUF_AUX_P, (B, 4, A)).cpu().numpy()
            fw = dict(alpha=rel_err(ax[:, 0], aux['alpha'].detach().numpy()), beta=rel_err(ax[:, 1], aux['beta'].detach().numpy()))
        else:
            eng.value_forward_backward(dval)
            loss, gh, gt, aux = ruled(o64.value_grads, oracle_batch(val))
            fw = dict(values=rel_err(eng.buffer(_lib.BUF_AUX_V, (B, 2)).cpu().numpy(), aux['values'].detach().numpy()))
        le = eng.metrics(kind)['loss']
        fw['loss'] = abs(le - float(loss.detach())) / max(1.0, abs(float(loss.detach())))
        rep = dict(forward=fw, groups={})
        for views, ref in ((eng.grad_views('trunk'), gt), (eng.grad_views(kind), gh)):
            gmax = max(float(g.abs().max()) for g in ref.values())
            for name, g in ref.items():
                if is_zero_gradient(name):
                    continue
                grp = _c3_group(name) if name.split('.')[0] not in ('pi', 'v') else 'head'
                e = float((views[name].cpu().double() - g.detach()).abs().max()) / max(float(g.abs().max()), 1e-3 * gmax)
                r = rep['groups'].setdefault(grp, dict(worst=0.0, tensor='', errs=[]))
                r['errs'].append(e)
                if e >= r['worst']:
                    r['worst'], r['tensor'] = e, name
        for r in rep['groups'].values():
            r['median'] = float(np.median(r.pop('errs')))
        report[kind] = rep
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(report, open(f'gpurun_out/parity_report_bf16_storage_rule_B{B}_{H}x{W}.json', 'w'), indent=1)
    print(f'[bf16 storage engine vs storage-rule oracle, B={B} {H}x{W}]', json.dumps(report))
    for kind, rep in report.items():
        for k, v in rep['forward'].items():
            assert v <= (3e-2 if k == 'loss' else 2e-1), (kind, k, v)
        for grp, r in rep['groups'].items():
            assert r['median'] <= 0.4, (kind, grp, r)


@pytest.mark.parametrize('compute', ['f32', 'f32_nofin', 'bf16s', 'bf16s_fused'])
def test_every_unit_backward_against_the_oracle_locally(compute, monkeypatch):
    """Unit-by-unit HIP-vs-oracle check of the tower's backward in both storage modes (the engine-level plumbing of
    the bf16-storage mode -- tens_a, element-sized slots, the `at` flag through ~40 call sites -- was only covered by cosine gates).
    End to end, ~50 train-mode BatchNorms amplify a perturbation of one bf16 ulp a thousandfold (the float32 engine meets the
    float64 oracle at 1e-4, i.e. 1e-7 rounding x 1e3), so an end-to-end comparison of two bf16 implementations can only agree in
    direction.  LOCALLY the amplification is that of one unit: every ShuffleNet unit is replayed in the float64 oracle FROM THE
    ENGINE'S OWN stored input X and stored output gradient G (oracle/model.py::shufflenet_unit with the storage rule `_st` and the
    bf16-operand rule for configs[2]; the engine's own ReLU6 decisions), and the engine's stored unit output, its stored input
    gradient and the unit's weight gradients must match per tensor: 1e-4 in float32, bf16 rounding level in bf16 storage."""
    import json
    import os
    from oracle import model as OM
    from oracle.spec import unit_plan
    from tests.util import make_pair, make_batches, to_dev, is_zero_gradient, engine_decisions
    B, H, W, A = 32, 48, 64, 2
    label = compute
    if compute == 'f32_nofin':      # the fused conv backward finalizes the BatchNorm behind it on load by default (round 5): here the
        # stand-alone bn_bwd_finalize launches of rounds 1-4
        monkeypatch.setenv('CDRL_FIN_ON_LOAD', '0')
        compute = 'f32'
    if compute == 'bf16s_fused':        # the engine takes the fused backward in bf16 storage from B = 512 on its own: forced here
        monkeypatch.setenv('CDRL_FUSED_BWD', '1')
        compute = 'bf16s'
    oracle, eng = make_pair(B, H, W, seed=5, A=A, compute=compute, with64=True)
    o64 = oracle.o64
    pol, _ = make_batches(B, H, W, seed=5, A=A, faithful=True)
    eng.policy_forward_backward(to_dev(pol))
    torch.cuda.synchronize()
    T = eng.cfg.T
    dt = torch.bfloat16 if compute == 'bf16s' else torch.float32
    items = engine_decisions(eng, oracle.cfg)
    grads = eng.grad_views('trunk')

    def tensor(name, h, w, c):
        return eng.named_buffer(name, dt).view(T, B, h, w, c).permute(0, 1, 4, 2, 3).double().cpu().contiguous()

    hs, ws = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    h, w = -(-hs // 2), -(-ws // 2)
    prev, prev_c = 'img.stem.pool.out', oracle.cfg.stem_channels
    cursor = 2
    report = {}
    for u in unit_plan(oracle.cfg):
        pre = f"img.s{u['stage']}.u{u['unit']}"
        ho, wo = (-(-h // 2), -(-w // 2)) if u['stride'] == 2 else (h, w)
        C = oracle.cfg.stage_channels[u['stage']]
        n_items = 3 if u['stride'] == 2 else 2
        x = tensor(prev, h, w, prev_c).requires_grad_(True)
        G = tensor(pre + '.out.g', ho, wo, C)
        names = [n for n in o64.trunk if n.startswith(pre + '.') and 'moving' not in n]
        params = [o64.trunk[n] for n in names]
        OM.DEC.items = items[cursor:cursor + n_items]
        OM.PW_BF16_OPERANDS = OM.BF16_STORAGE = compute == 'bf16s'
        try:
            OM.DEC.start('replay')
            out = OM.shufflenet_unit(x, o64.trunk, u, True)
            assert OM.DEC.cursor == n_items
            gs = torch.autograd.grad(out, [x] + params, grad_outputs=G, allow_unused=True)
        finally:
            OM.DEC.start('off')
            OM.PW_BF16_OPERANDS = OM.BF16_STORAGE = False
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
        rep = dict(out=rel(tensor(pre + '.out', ho, wo, C), out.detach()), dx=rel(tensor(prev + '.g', h, w, prev_c), gs[0]))
        gmax = max(float(g.abs().max()) for n, g in zip(names, gs[1:]) if not is_zero_gradient(n))
        worst = ('', 0.0)
        for n, g in zip(names, gs[1:]):
            if is_zero_gradient(n):
                continue
            e = float((grads[n].cpu().double() - g).abs().max()) / max(float(g.abs().max()), 1e-3 * gmax)
            if e >= worst[1]:
                worst = (n, e)
        rep['weights'] = worst[1]
        rep['worst_weight'] = worst[0]
        report[pre] = rep
        cursor += n_items
        prev, prev_c, h, w = pre + '.out', C, ho, wo
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(report, open(f'gpurun_out/parity_report_units_local_{label}_B{B}_{H}x{W}.json', 'w'), indent=1)
    w_out, w_dx, w_w = (max(r[k] for r in report.values()) for k in ('out', 'dx', 'weights'))
    print(f'[unit-local parity, {label}] worst over 16 units: out {w_out:.2e}, input gradient {w_dx:.2e}, weight gradients {w_w:.2e}')
    # measured worst over the 16 units: float32 5.6e-7 / 7.1e-7 / 5.6e-6; bf16 storage 2.8e-3 (under one bf16 ulp of the tensor maximum) /
    # 3.5e-3 / 2.7e-2 (always bn1.gamma, every other tensor <= 1e-2: the engine takes the BatchNorm-backward sums from the float32 gradient BEFORE it is rounded for
    # storage, the oracle's autograd from the rounded one)
    tol = dict(out=5e-6, dx=5e-6, weights=3e-5) if compute == 'f32' else dict(out=4e-3, dx=5e-3, weights=4e-2)
    for pre, r in report.items():
        for k in ('out', 'dx', 'weights'):
            assert r[k] <= tol[k], (pre, k, r)
