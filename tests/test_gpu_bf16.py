"""bf16 path (BASELINE.json configuration 3), kernel level: the 1x1 convolution with bf16 activations and bf16 MFMA
(cdrl_pwconv_bf16) against a float64 reference evaluated on the SAME bf16-rounded operands.

Tolerance (stated, as the bf16 path cannot meet the float32 bar of 1e-4): the kernel accumulates exact bf16 x bf16 products in
float32 and rounds the result once to bf16, so |err| <= 2^-9 |c| (half a bf16 ulp) + float32 accumulation noise; the statistics
epilogue is taken from the rounded outputs in double and must match the float64 sums of those outputs to 1e-6."""
import ctypes as C

import numpy as np
import pytest
import torch

from carla_driving_rl_agent_amd import _lib

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def test_bf16_round_trip_matches_torch(lib):
    x = torch.randn(100003, device=DEV) * 37.0
    y = torch.empty(100003, dtype=torch.bfloat16, device=DEV)
    _lib.check(lib.cdrl_f32_to_bf16(P(x), P(y), x.numel(), S()))
    assert torch.equal(y, x.to(torch.bfloat16))                          # round to nearest even
    z = torch.empty_like(x)
    _lib.check(lib.cdrl_bf16_to_f32(P(y), P(z), x.numel(), S()))
    assert torch.equal(z, y.float())


@pytest.mark.parametrize('G,Mg,K,N,pro', [(4, 1000, 116, 116, True), (4, 777, 116, 116, False), (2, 515, 24, 56, True), (1, 4100, 60, 92, False),
                                          (4, 333, 116, 120, True), (3, 64, 28, 28, False), (4, 49152, 116, 116, True)])
@pytest.mark.parametrize('packed', [False, True])
def test_pwconv_bf16(lib, G, Mg, K, N, pro, packed):
    rng = np.random.default_rng(G + Mg + K + N)
    M = G * Mg
    lda, a_coff, ldc, c_coff = K + 12, 4, N + 8, 4
    a = torch.tensor(rng.standard_normal((M, lda)), dtype=torch.float32, device=DEV).to(torch.bfloat16)
    w = torch.tensor(rng.standard_normal((K, N)) / np.sqrt(K), dtype=torch.float32, device=DEV)
    bias = torch.tensor(rng.standard_normal(N) * 0.1, dtype=torch.float32, device=DEV)
    stats = None
    if pro:
        stats = torch.tensor(rng.uniform(0.5, 1.5, (4, G, K)), dtype=torch.float32, device=DEV)
        stats[3] = torch.tensor(rng.uniform(-0.3, 0.3, (G, K)), dtype=torch.float32, device=DEV)
    c = torch.full((M, ldc), 7.0, dtype=torch.bfloat16, device=DEV)
    nb = int(lib.cdrl_pwconv_bf16_partial_rows(G, Mg, N, K))
    part = torch.zeros((G, nb, 2, N), dtype=torch.float64, device=DEV)
    wp = None
    if packed:
        wp = torch.zeros(int(lib.cdrl_pwconv_bf16_packed_elems(K)), dtype=torch.bfloat16, device=DEV)
        _lib.check(lib.cdrl_pwconv_bf16_pack(P(w), K, N, P(wp), S()))
    _lib.check(lib.cdrl_pwconv_bf16(P(a), lda, a_coff, P(stats), None if packed else P(w), P(wp), P(bias), P(c), ldc, c_coff, G, Mg,
                                    N, K, P(part), S()))
    # reference on the operands the kernel's MFMA sees: bf16 A (after the float32 BN-apply, re-rounded to bf16), bf16 W
    av = a[:, a_coff:a_coff + K].float().view(G, Mg, K)
    if pro:
        # fmaf = exact product + sum, ONE rounding to float32 (float64 evaluates it exactly), then the bf16 rounding of the MFMA operand
        av = (stats[2].view(G, 1, K).double() * av.double() + stats[3].view(G, 1, K).double()).float().to(torch.bfloat16).float()
    ref = av.double().view(M, K) @ w.to(torch.bfloat16).double() + bias.double()
    got = c[:, c_coff:c_coff + N].double()
    err = (got - ref).abs()
    bound = ref.abs() * 2.0 ** -8 + 1e-3
    assert bool((err <= bound).all()), float((err / bound).max())
    assert float(err.max() / ref.abs().max()) < 4e-3
    # untouched padding columns
    assert bool((c[:, :c_coff] == 7.0).all()) and bool((c[:, c_coff + N:] == 7.0).all())
    sums = part.sum(dim=1)                                                # [G][2][N]
    g64 = got.view(G, Mg, N)
    assert torch.allclose(sums[:, 0], g64.sum(1), rtol=1e-9, atol=1e-6)
    assert torch.allclose(sums[:, 1], (g64 * g64).sum(1), rtol=1e-9, atol=1e-6)


# ------------------------------------------------------------------------------------------
# bf16-OPERAND compute mode of the float32 engine kernels (float32 tensors, bf16 MFMA operands): cdrl_pwconv_fused_packed /
# cdrl_pwconv_bn_bwd_packed with packed_bf16 = 1.  Forward: against float64 on the operands the MFMA sees (A after the float32
# prologue, rounded to bf16 like torch's .bfloat16(); W rounded to bf16) -> float32-accumulation accuracy (1e-5).  Backward-data and filter gradient:
# against float64 autograd of the UNROUNDED layer -> the stated bf16-operand tolerance 6e-3 (and > 1e-4: the bf16 pipe was used).
# ------------------------------------------------------------------------------------------
def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _bf(x):
    return torch.tensor(np.asarray(x, np.float32)).to(torch.bfloat16).to(torch.float64).numpy()


@pytest.mark.parametrize('G,Mg,K,N,pro,epi,bt', [(4, 330, 58, 58, 1, 1, 0), (4, 1000, 116, 116, 1, 1, 0), (2, 515, 24, 58, 0, 1, 0),
                                                  (4, 257, 116, 116, 0, 2, 1), (4, 120, 232, 232, 1, 1, 0), (1, 77, 232, 232, 0, 2, 1),
                                                  (4, 96, 58, 24, 0, 0, 1), (3, 200, 116, 58, 0, 0, 0)])
@pytest.mark.parametrize('bf16', [0, 1])
def test_pwconv_fused_packed(lib, G, Mg, K, N, pro, epi, bt, bf16):
    rng = np.random.default_rng(G * Mg + K + N)
    M = G * Mg
    lda, coff = K + 6, 2
    a = rng.standard_normal((M, lda)).astype(np.float32)
    w = (rng.standard_normal((N, K) if bt else (K, N)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    pst = rng.uniform(0.5, 1.5, (4, G, K)).astype(np.float32)
    est = rng.uniform(0.5, 1.5, (4, G, N)).astype(np.float32)
    ey = rng.standard_normal((M, N)).astype(np.float32)
    dev = lambda x: torch.tensor(x, device=DEV)
    A, Wd, Bd, PS, ES, EY = dev(a), dev(w), dev(bias), dev(pst), dev(est), dev(ey)
    sbk, sbn = (1, K) if bt else (N, 1)
    nb = int(lib.cdrl_pwconv_fused_partial_rows(G, Mg, N, K))
    wp = torch.zeros(int(lib.cdrl_pwconv_pack_elems(N, K)), device=DEV)
    _lib.check(lib.cdrl_pwconv_pack(P(Wd), K, N, sbk, sbn, P(wp), bf16, S()))

    def run(packed):
        out = torch.full((M, N + 3), 1.0, device=DEV)
        part = torch.zeros((G, nb, 2, N), dtype=torch.float64, device=DEV)
        args = (P(A), lda, coff, P(PS) if pro else None, P(Wd), sbk, sbn, None if bt else P(Bd), P(out), N + 3, 1, 1 if bt else 0, G, Mg, N,
                K, epi, P(EY), P(ES), P(part))
        if packed:
            _lib.check(lib.cdrl_pwconv_fused_packed(*args, P(wp), bf16, S()))
        else:
            _lib.check(lib.cdrl_pwconv_fused(*args, S()))
        return out.cpu().numpy(), part.sum(dim=1).cpu().numpy()

    got, ps = run(True)
    assert np.all(got[:, 0] == 1.0) and np.all(got[:, N + 1:] == 1.0)
    if not bf16:                                        # float32 fragments: the same products in the same order
        ref, rps = run(False)
        assert np.array_equal(got, ref) and np.array_equal(ps, rps)
        return
    a32 = a[:, coff:coff + K].reshape(G, Mg, K)
    if pro:     # the kernel's prologue: one float32 fma per element
        a32 = (a32.astype(np.float64) * pst[2][:, None, :].astype(np.float64) + pst[3][:, None, :].astype(np.float64)).astype(np.float32)
    w64 = _bf(w).T if bt else _bf(w)
    ref = (_bf(a32) @ w64 + (0.0 if bt else bias.astype(np.float64))).reshape(M, N)
    assert _rel(got[:, 1:1 + N], ref + (1.0 if bt else 0.0)) < 1e-5
    if epi:
        r3 = ref.reshape(G, Mg, N)
        assert _rel(ps[:, 0], r3.sum(axis=1)) < 1e-5
        if epi == 1:
            assert _rel(ps[:, 1], (r3 * r3).sum(axis=1)) < 1e-5
        else:
            xh = (ey.astype(np.float64).reshape(G, Mg, N) - est[0][:, None, :]) * est[1][:, None, :]
            assert _rel(ps[:, 1], (r3 * xh).sum(axis=1)) < 2e-5
    # and it is the bf16 pipe: the unrounded float64 product differs at the bf16 operand level
    full = (a32.astype(np.float64) @ (w.astype(np.float64).T if bt else w.astype(np.float64)) + (0.0 if bt else bias)).reshape(M, N)
    assert 1e-4 < _rel(got[:, 1:1 + N], full + (1.0 if bt else 0.0)) < 1e-2


@pytest.mark.parametrize('G,Mg,K,N,relu,shuffle,xpro', [(4, 330, 58, 58, 1, 1, 0), (4, 96, 116, 116, 1, 1, 1), (2, 500, 24, 58, 1, 0, 0),
                                                        (4, 257, 58, 24, 0, 0, 1), (4, 120, 232, 232, 1, 1, 1)])
@pytest.mark.parametrize('bf16', [0, 1])
def test_pwconv_bn_bwd_packed(lib, G, Mg, K, N, relu, shuffle, xpro, bf16):
    from oracle import model as OM
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
    yt = xin @ wt + bt
    out = OM.bn_slices(yt.permute(0, 2, 1)[:, None], p, 'b', True, True)
    if relu:
        out = OM.relu6(out)
    ctot = 2 * N if shuffle else N
    coff = N if shuffle else 0
    dout = rng.standard_normal((M, ctot)).astype(np.float32)
    idx = [((coff + c) & 1) * (ctot // 2) + ((coff + c) >> 1) for c in range(N)] if shuffle else list(range(N))
    out.backward(torch.tensor(dout[:, idx], dtype=torch.float64).reshape(G, Mg, N).permute(0, 2, 1)[:, None])
    dev = lambda v: torch.tensor(np.asarray(v, np.float32), device=DEV)
    X, Wd, XS, DO = dev(x), dev(w), dev(xst), dev(dout)
    y = dev(yt.detach().reshape(M, N).float().numpy())
    stats = torch.zeros(4 * G * N, device=DEV)
    tmp = torch.zeros((M, N), device=DEV)
    ws0 = torch.zeros(G * 256 * 2 * N, dtype=torch.float64, device=DEV)
    gam, bet = dev(p['b.gamma'].detach().float()), dev(p['b.beta'].detach().float())
    mm, mv = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
    _lib.check(lib.cdrl_bn_train_fwd(P(y), G, Mg, N, P(gam), P(bet), P(mm), P(mv), 1, relu, P(tmp), N, 0, 0, P(stats), P(ws0), S()))
    wtp = torch.zeros(int(lib.cdrl_pwconv_pack_elems(K, N)), device=DEV)
    _lib.check(lib.cdrl_pwconv_pack(P(Wd), N, K, 1, N, P(wtp), bf16, S()))       # B(k = n_out, n = k_in) = W[k_in * N + n_out]

    def run(packed):
        ws = torch.zeros(int(lib.cdrl_pwconv_bn_bwd_workspace_bytes(G, Mg, N, K)), dtype=torch.uint8, device=DEV)
        dg, dbt, coef = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV), torch.zeros(3 * G * N, device=DEV)
        dx = torch.full((M, K + 4), 2.0, device=DEV)
        dw, db = torch.zeros((K, N), device=DEV), torch.zeros(N, device=DEV)
        args = (P(DO), ctot, coff, ctot if shuffle else 0, relu, P(y), P(stats), P(X), K, 0, P(XS) if xpro else None, P(Wd), G, Mg, N, K,
                P(dg), P(dbt), P(coef), P(dx), K + 4, 2, 1, P(dw), P(db), P(ws))
        if packed:
            _lib.check(lib.cdrl_pwconv_bn_bwd_packed(*args, P(wtp), bf16, S()))
        else:
            _lib.check(lib.cdrl_pwconv_bn_bwd(*args, S()))
        return [t.cpu().numpy() for t in (dg, dbt, dx, dw, db)]

    dg, dbt, dx, dw, db = run(True)
    if not bf16:
        for u, v in zip((dg, dbt, dx, dw, db), run(False)):
            assert np.array_equal(u, v)
        return
    assert _rel(dg, p['b.gamma'].grad.numpy()) < 2e-5 and _rel(dbt, p['b.beta'].grad.numpy()) < 2e-5      # float32 reductions
    ew = _rel(dw, wt.grad.numpy())
    assert 1e-5 < ew < 6e-3, ew                                  # filter gradient from bf16 operands as well (sum over G*Mg rows)
    ref_dxin = (xt.grad.reshape(M, K).numpy() / xst[2].astype(np.float64).repeat(Mg, axis=0)) if xpro else xt.grad.reshape(M, K).numpy()
    e = _rel(dx[:, 2:2 + K] - 2.0, ref_dxin)
    assert 1e-4 < e < 6e-3, e
    assert np.all(dx[:, :2] == 2.0) and np.all(dx[:, 2 + K:] == 2.0)


# ------------------------------------------------------------------------------------------
# bf16-operand mode end to end: LearnerEngine(compute='bf16').
#
# What can be asserted, and what cannot (measured, profiles/r02_bf16_operand_parity.json):
#   * The kernels implement the rounding rule exactly (tests above: 1e-5 against float64 on the same rounded operands).
#   * FORWARD quantities of the whole network agree with the float64 oracle evaluated under the same rule
#     (oracle/model.py::_PwBf16Operands) -- loss to 2e-2 (measured 7e-5 .. 5e-3), Beta parameters to 1.5e-1 of their scale (measured
#     4e-2 .. 6e-2; the rule's oracle itself sits 8e-2 from the exact one) -- and with the float32 engine.
#   * GRADIENTS of the 50-layer train-mode-BatchNorm tower do NOT have a meaningful bf16 tolerance on this workload: bf16 operand
#     rounding (2^-9 relative) is a discrete decision like a ReLU6 mask, the tower amplifies perturbations by ~1e5 (its float32
#     gradients already sit 1e-2..1e-1 from float64, tests/test_gpu_learner.py), and the float64 oracle WITH the rule differs from
#     the float64 oracle WITHOUT it by 50-100 % per tensor (L2) at the benchmark's random-init / random-input state.  That is a
#     property of the reference network under bf16, not of a kernel; it is recorded, and the test asserts what a user can rely
#     on: finite gradients of the right scale that point the same way as the float32 engine's (cosine), and training that
#     follows the float32 loss curve (next test).
# ------------------------------------------------------------------------------------------
def _flat(views, names):
    return np.concatenate([views[n].detach().cpu().numpy().astype(np.float64).ravel() for n in names])


def _cos(a, b):
    return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))


# (round 6: the (32, 90, 120, 2) case, 52 s of CPU oracle, left the suite -- the operand mode at the real image size is held by
#  test_config3_three_engines_at_batch_1024 and the full-size property tests; the suite stays under 15 minutes on the slowest boxes)
@pytest.mark.parametrize('B,H,W,A', [(64, 48, 64, 2)])
def test_bf16_operand_engine_vs_oracle(B, H, W, A):
    from oracle import model as OM
    from tests.util import make_pair, make_batches, oracle_batch, to_dev, rel_err, is_zero_gradient
    oracle, eng = make_pair(B, H, W, seed=5, A=A, with64=True, compute='bf16')
    _, eng32 = make_pair(B, H, W, seed=5, A=A)
    o64 = oracle.o64
    pol, val = make_batches(B, H, W, seed=5, A=A, faithful=True)
    dpol = to_dev(pol)
    eng.policy_forward_backward(dpol)
    eng32.policy_forward_backward(dpol)
    OM.PW_BF16_OPERANDS = True
    try:
        loss_b, gp_b, gt_b, aux_b = o64.policy_grads(oracle_batch(pol))
    finally:
        OM.PW_BF16_OPERANDS = False
    loss_x, gp_x, gt_x, aux_x = o64.policy_grads(oracle_batch(pol))
    l16, l32 = eng.metrics('policy')['loss'], eng32.metrics('policy')['loss']
    report = dict(loss=dict(engine_bf16=l16, oracle_bf16_rule=float(loss_b.detach()), engine_f32=l32, oracle_exact=float(loss_x.detach())))
    assert abs(l16 - float(loss_b.detach())) <= 2e-2 * max(1.0, abs(float(loss_b.detach())))
    assert abs(l16 - l32) <= 2e-2 * max(1.0, abs(l32))
    ax = eng.buffer(_lib.BUF_AUX_P, (B, 4, A)).cpu().numpy()
    for i, k in enumerate(('alpha', 'beta')):
        e = rel_err(ax[:, i], aux_b[k].detach().numpy())
        report[k] = dict(engine_bf16_vs_oracle_bf16_rule=e, oracle_bf16_rule_vs_exact=rel_err(aux_b[k].detach().numpy(), aux_x[k].detach().numpy()))
        assert e <= 1.5e-1, (k, e)
    names = [n for n in gt_b if not is_zero_gradient(n)]
    tower = [n for n in names if n.startswith('img.')]
    tail = [n for n in names if not n.startswith('img.')]
    g16, g32 = eng.grad_views('trunk'), eng32.grad_views('trunk')
    ob = {n: g for n, g in gt_b.items()}
    ox = {n: g for n, g in gt_x.items()}
    for grp, ns in (('tower', tower), ('tail', tail)):
        a16, a32, ab, ax_ = _flat(g16, ns), _flat(g32, ns), _flat(ob, ns), _flat(ox, ns)
        report[grp] = dict(cos_engine_bf16_vs_engine_f32=_cos(a16, a32), cos_engine_bf16_vs_oracle_bf16_rule=_cos(a16, ab),
                           cos_oracle_bf16_rule_vs_oracle_exact=_cos(ab, ax_), norm_ratio_engine_bf16_over_f32=float(np.linalg.norm(a16) / np.linalg.norm(a32)),
                           l2_rel_oracle_bf16_rule_vs_exact=float(np.linalg.norm(ab - ax_) / np.linalg.norm(ax_)),
                           l2_rel_engine_bf16_vs_oracle_bf16_rule=float(np.linalg.norm(a16 - ab) / np.linalg.norm(ab)))
        assert np.all(np.isfinite(a16))
    hp16, hp32 = eng.grad_views('policy'), eng32.grad_views('policy')
    hn = [n for n in gp_b if not is_zero_gradient(n)]
    report['policy_head'] = dict(cos_engine_bf16_vs_engine_f32=_cos(_flat(hp16, hn), _flat(hp32, hn)),
                                 cos_engine_bf16_vs_oracle_bf16_rule=_cos(_flat(hp16, hn), _flat({n: g for n, g in gp_b.items()}, hn)))
    import json
    import os
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(report, open(f'gpurun_out/parity_report_bf16_B{B}_{H}x{W}.json', 'w'), indent=1)
    # the engine is as close to the rule's oracle as that oracle is to the exact one (it implements the rule, nothing worse)
    for grp in ('tower', 'tail'):
        r = report[grp]
        assert 0.25 <= r['norm_ratio_engine_bf16_over_f32'] <= 4.0, (grp, r)
        assert r['cos_engine_bf16_vs_oracle_bf16_rule'] >= r['cos_oracle_bf16_rule_vs_oracle_exact'] - 0.15, (grp, r)


def test_bf16_operand_training_tracks_float32():
    """Loss-curve agreement over a short run: the same 12 update-steps (policy + value, re-sampled loss with the same Philox
    stream) on the bf16-operand and the float32 engine from identical weights.  Stated tolerance: each loss within 15 % of the
    float32 curve's scale at every step (measured 8 %: the two runs take different rounding decisions from step 1 on and the
    policy loss falls by two orders of magnitude in 12 steps), both value losses decreasing, everything finite."""
    from tests.util import make_pair, make_batches, to_dev
    B, H, W = 32, 48, 64
    _, e16 = make_pair(B, H, W, seed=9, compute='bf16')
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
    assert h32[-1, 1] < h32[0, 1] and h16[-1, 1] < h16[0, 1]          # the value loss goes down on both
    assert h16[-1, 0] < 0.2 * h16[0, 0]                                # and the bf16 run minimises the policy objective as well


def test_bf16_operand_mode_is_deterministic():
    """Same inputs, same weights -> bit-identical losses and gradient arenas (fixed-order reductions in every BF kernel)."""
    from tests.util import make_pair, make_batches, to_dev
    B, H, W = 16, 90, 120
    pol, val = make_batches(B, H, W, seed=11)
    dpol, dval = to_dev(pol), to_dev(val)
    outs = []
    for _ in range(2):
        _, e = make_pair(B, H, W, seed=11, compute='bf16')
        e.policy_forward_backward(dpol)
        lp = e.metrics('policy')['loss']
        gp = e.grads.clone()
        e.policy_apply()
        e.value_forward_backward(dval)
        outs.append((lp, e.metrics('value')['loss'], gp, e.grads.clone(), e.params.clone()))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
    for a, b in zip(outs[0][2:], outs[1][2:]):
        assert torch.equal(a, b)
