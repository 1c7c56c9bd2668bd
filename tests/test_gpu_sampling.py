"""On-device Beta sampling with pathwise derivatives (cdrl_beta_sample): distribution, reproducibility,
and the implicit Gamma / Beta derivatives against scipy (quantile-function finite differences)."""
import ctypes as C

import numpy as np
import pytest
import torch
from scipy import stats

from carla_driving_rl_agent_amd import _lib

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def test_gamma_implicit_gradient_matches_quantile_finite_difference(lib):
    a = np.array([1.01, 1.5, 2.0, 5.0, 20.0, 80.0] * 5)
    p = np.repeat([0.001, 0.1, 0.5, 0.9, 0.999], 6)
    g = stats.gamma.ppf(p, a)
    h = 1e-5 * a
    fd = (stats.gamma.ppf(p, a + h) - stats.gamma.ppf(p, a - h)) / (2 * h)
    A, G = torch.tensor(a, device=DEV), torch.tensor(g, device=DEV)
    out = torch.zeros_like(A)
    _lib.check(lib.cdrl_gamma_implicit_grad(P(A), P(G), len(a), P(out), S()))
    assert np.allclose(out.cpu().numpy(), fd, rtol=1e-7)


def test_beta_sample_distribution_and_determinism(lib):
    rows, A = 200000, 2
    al = np.tile(np.array([[1.01, 3.5]], np.float32), (rows, 1))
    be = np.tile(np.array([[7.0, 1.3]], np.float32), (rows, 1))
    ab = torch.tensor(np.concatenate([al, be], 1), device=DEV)          # [rows][2A]: alpha | beta, ld = 2A
    u = torch.zeros((rows, A), device=DEV)
    ja, jb = torch.zeros_like(u), torch.zeros_like(u)
    _lib.check(lib.cdrl_beta_sample(P(ab), C.c_void_p(ab.data_ptr() + 4 * A), rows, A, 2 * A, 1234, 7, P(u), P(ja), P(jb), S()))
    un = u.cpu().numpy()
    assert (un > 0).all() and (un < 1).all()
    for col in range(A):
        a, b = float(al[0, col]), float(be[0, col])
        assert abs(un[:, col].mean() - a / (a + b)) < 4 * stats.beta.std(a, b) / np.sqrt(rows)
        assert stats.kstest(un[:20000, col], 'beta', args=(a, b)).pvalue > 1e-3
    u2 = torch.zeros_like(u)
    _lib.check(lib.cdrl_beta_sample(P(ab), C.c_void_p(ab.data_ptr() + 4 * A), rows, A, 2 * A, 1234, 7, P(u2), None, None, S()))
    assert torch.equal(u, u2)                                           # same (seed, offset) -> same stream
    _lib.check(lib.cdrl_beta_sample(P(ab), C.c_void_p(ab.data_ptr() + 4 * A), rows, A, 2 * A, 1234, 8, P(u2), None, None, S()))
    assert not torch.equal(u, u2)
    # pathwise derivatives: E[d/dalpha f(u)] == d/dalpha E[f(u)] for f(u) = u:  d/dalpha (a/(a+b)) = b/(a+b)^2
    jan, jbn = ja.cpu().numpy().astype(np.float64), jb.cpu().numpy().astype(np.float64)
    for col in range(A):
        a, b = float(al[0, col]), float(be[0, col])
        assert abs(jan[:, col].mean() - b / (a + b) ** 2) < 5 * jan[:, col].std() / np.sqrt(rows)
        assert abs(jbn[:, col].mean() + a / (a + b) ** 2) < 5 * jbn[:, col].std() / np.sqrt(rows)


def test_resampled_policy_step_runs_and_uses_new_policy_sample():
    from tests.util import make_pair, make_batches, to_dev
    B, H, W = 8, 48, 64
    _, eng = make_pair(B, H, W, seed=2)
    pol, _ = make_batches(B, H, W, seed=2)
    d = to_dev(pol)
    eng.policy_forward_backward_resample(d, seed=5, offset=1)
    g1 = eng.grads.clone()
    u1 = eng.buffer(_lib.BUF_SAMPLE, (B, 2)).clone()
    aux = eng.buffer(_lib.BUF_AUX_P, (B, 4, 2)).cpu().numpy()
    assert (u1 > 0).all() and (u1 < 1).all() and np.isfinite(eng.metrics('policy')['loss'])
    # the sample follows the NEW policy's (alpha, beta): compare with the Beta mean on average
    mean = aux[:, 0] / (aux[:, 0] + aux[:, 1])
    assert abs(float(u1.mean()) - float(mean.mean())) < 0.25
    eng.policy_forward_backward_resample(d, seed=5, offset=2)          # another offset -> another sample / gradient
    assert not torch.equal(u1, eng.buffer(_lib.BUF_SAMPLE, (B, 2)))
    assert not torch.equal(g1, eng.grads)
