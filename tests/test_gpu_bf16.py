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
