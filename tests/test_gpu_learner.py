"""End-to-end parity of the HIP learner (through the C ABI) against the CPU oracle on identical
seeded synthetic inputs: dynamics output, action 'logits' (alpha, beta), values, losses, every
gradient tensor, updated weights, Adam state, BN moving statistics, old-policy copy.

Tolerance: 1e-4 relative to each tensor's scale (BASELINE.json north_star, fp32)."""
import numpy as np
import pytest
import torch

from carla_driving_rl_agent_amd import _lib
from tests.util import make_pair, make_batches, oracle_batch, to_dev, rel_err, is_degenerate_bias

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _compare_grads(eng_views, oracle_grads, tol, scale_floor=0.0):
    worst = (0.0, None)
    gmax = max(float(g.abs().max()) for g in oracle_grads.values())
    for name, g in oracle_grads.items():
        got = eng_views[name].detach().cpu().numpy().astype(np.float64)
        ref = g.detach().numpy().astype(np.float64)
        denom = max(np.abs(ref).max(), scale_floor * gmax, 1e-30)
        e = float(np.abs(got - ref).max() / denom)
        if e > worst[0]:
            worst = (e, name)
    assert worst[0] < tol, f'gradient mismatch {worst}'
    return worst


@pytest.mark.parametrize('B,H,W,A', [(6, 48, 64, 2), (4, 41, 58, 3)])
def test_trunk_forward_and_predict(B, H, W, A):
    oracle, eng = make_pair(B, H, W, seed=7, A=A)
    pol, _ = make_batches(B, H, W, seed=7, A=A)
    dstates = to_dev(pol['states'])
    # inference path first (moving statistics, old_policy + value heads) - CARLANetwork.predict
    alpha, beta, value, dyn = oracle.predict(pol['states'])
    out = eng.predict(dstates)
    assert rel_err(out['dynamics'].cpu().numpy(), dyn.numpy()) < TOL
    assert rel_err(out['alpha'].cpu().numpy(), alpha.numpy()) < TOL
    assert rel_err(out['beta'].cpu().numpy(), beta.numpy()) < TOL
    assert rel_err(out['value'].cpu().numpy(), value.numpy()) < TOL
    # training-mode forward (per-time-slice batch statistics)
    from oracle import model as OM
    taps = {}
    st = {k: torch.as_tensor(v) for k, v in pol['states'].items()}
    with torch.no_grad():
        d_ref = OM.dynamics_forward(st, oracle.trunk, oracle.cfg, True, taps)
    d = eng.trunk_forward_train(dstates)
    feat = eng.buffer(_lib.BUF_IMG_FEAT, (eng.cfg.T, B, eng.cfg.last))
    assert rel_err(feat.cpu().numpy(), taps['img_feat'].numpy()) < TOL
    assert rel_err(d.cpu().numpy(), d_ref.numpy()) < TOL
    # BN moving statistics after one training forward (T sequential EMA updates, Bessel for rank 4)
    pv = eng.param_views('trunk')
    for name in ('img.stem.bn.moving_mean', 'img.stem.bn.moving_var', 'img.s1.u3.bn2.moving_var', 'img.head.bn.moving_var',
                 'road.bn1.moving_var', 'dyn.bn.moving_mean'):
        assert rel_err(pv[name].cpu().numpy(), oracle.trunk[name].numpy()) < TOL, name


@pytest.mark.parametrize('B,H,W,A,faithful', [(6, 48, 64, 2, True), (5, 41, 58, 3, False)])
def test_policy_and_value_step(B, H, W, A, faithful):
    oracle, eng = make_pair(B, H, W, seed=3, A=A)
    pol, val = make_batches(B, H, W, seed=3, A=A, faithful=faithful)
    dpol, dval = to_dev(pol), to_dev(val)

    # ---------------- policy minibatch step
    loss, gp, gt, aux = oracle.policy_grads(oracle_batch(pol))
    eng.policy_forward_backward(dpol)
    m = eng.metrics('policy')
    assert abs(m['loss'] - float(loss)) < TOL * max(1.0, abs(float(loss)))
    ax = eng.buffer(_lib.BUF_AUX_P, (B, 4, A)).cpu().numpy()
    assert rel_err(ax[:, 0], aux['alpha'].detach().numpy()) < TOL          # action "logits"
    assert rel_err(ax[:, 1], aux['beta'].detach().numpy()) < TOL
    assert rel_err(ax[:, 2], aux['log_prob'].detach().numpy()) < TOL
    _compare_grads(eng.grad_views('policy'), gp, TOL)
    # trunk: compare relative to each tensor's own scale, with a floor for the analytically-zero
    # conv-bias gradients (pure rounding noise in any implementation)
    _compare_grads(eng.grad_views('trunk'), gt, 2 * TOL, scale_floor=1e-3)
    oracle.policy_step(None, grads=(loss, gp, gt, aux))
    eng.policy_apply()
    for model, ref in (('trunk', oracle.trunk), ('policy', oracle.policy), ('old_policy', oracle.old_policy)):
        views = eng.param_views(model)
        for name, r in ref.items():
            if is_degenerate_bias(name):
                continue
            e = rel_err(views[name].cpu().numpy(), r.detach().numpy())
            assert e < TOL, (model, name, e)
    m_t, v_t = eng.adam_views('trunk')
    for name in ('img.s0.u0.pw1.w', 'gru_image.kernel', 'dyn.fc.w'):
        assert rel_err(m_t[name].cpu().numpy(), oracle.opt_trunk.m[name].numpy()) < 2 * TOL, name
        assert rel_err(v_t[name].cpu().numpy(), oracle.opt_trunk.v[name].numpy()) < 2 * TOL, name

    # ---------------- value minibatch step (second trunk Adam step, t = 2)
    loss, gv, gt, aux = oracle.value_grads(oracle_batch(val))
    eng.value_forward_backward(dval)
    m = eng.metrics('value')
    assert abs(m['loss'] - float(loss)) < TOL * max(1.0, abs(float(loss)))
    vals = eng.buffer(_lib.BUF_AUX_V, (B, 2)).cpu().numpy()
    assert rel_err(vals, aux['values'].detach().numpy()) < TOL
    _compare_grads(eng.grad_views('value'), gv, TOL)
    _compare_grads(eng.grad_views('trunk'), gt, 2 * TOL, scale_floor=1e-3)
    oracle.value_step(None, grads=(loss, gv, gt, aux))
    eng.value_apply()
    for model, ref in (('trunk', oracle.trunk), ('value', oracle.value)):
        views = eng.param_views(model)
        for name, r in ref.items():
            if is_degenerate_bias(name):
                continue
            e = rel_err(views[name].cpu().numpy(), r.detach().numpy())
            assert e < TOL, (model, name, e)


def test_determinism():
    """Same inputs -> bit-identical gradients (no float atomics anywhere on the path)."""
    B, H, W = 4, 48, 64
    _, eng = make_pair(B, H, W, seed=11)
    pol, _ = make_batches(B, H, W, seed=11)
    dpol = to_dev(pol)
    mm = {k: v.clone() for k, v in eng.param_views('trunk').items() if 'moving' in k}
    eng.policy_forward_backward(dpol)
    g1 = eng.grads.clone()
    for k, v in eng.param_views('trunk').items():
        if 'moving' in k:
            v.copy_(mm[k])
    eng.policy_forward_backward(dpol)
    assert torch.equal(g1, eng.grads)


def test_bad_inputs_fail_loudly():
    B, H, W = 2, 48, 64
    _, eng = make_pair(B, H, W, seed=1)
    pol, _ = make_batches(B, H, W, seed=1)
    dpol = to_dev(pol)
    bad = dict(dpol)
    bad['advantages'] = dpol['advantages'][:1]
    with pytest.raises(ValueError):
        eng.policy_forward_backward(bad)
    with pytest.raises(_lib.CdrlError):
        _lib.check(eng.lib.cdrl_learner_policy_forward_backward(eng.h, None, 1.0, None), 'null batch')
