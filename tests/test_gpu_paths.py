"""Full-size consistency of the fused kernels with the unfused ones (which the small-size oracle parity tests also
cover): the same policy pass at the benchmark shape (B=256, T=4, 90x120x3) through (a) the default engine -- fused
depthwise block, persistent pointwise GEMMs with BatchNorm prologues / epilogues, BN-backward operand prologue, fused stem
block, identity half riding on the BatchNorm ops, side / aux streams -- and (b) the engine with every fusion and the side
stream switched off, in two subprocesses (the switches are per process).  Every reduction on the path accumulates in
double over float data and every GEMM is a k-ordered fmaf chain, so the two paths agree to float rounding everywhere except on
the analytically-zero gradients (pure rounding noise, tests/util.py)."""
import os
import subprocess
import sys

import pytest
import torch

from tests.util import is_degenerate_bias

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(path, **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, os.path.join(HERE, 'path_consistency.py'), 'run', path], env=e, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return torch.load(path)


def _run_steps(path, **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, os.path.join(HERE, 'path_consistency.py'), 'steps', path], env=e, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return torch.load(path)


def test_stream_synchronisation_modes_are_bit_identical_over_update_steps(tmp_path):
    """Round 6 changed HOW the engine's streams are ordered against each other, not what they compute: events without the system-scope
    fence, forks through stop events bound to the kernels in front of them (learned per body: the first run of a body records the old
    way), numbered side-stream records with lagged waits, one hand-over per update-step.  Twelve update-steps at the benchmark shape
    (where the kernels take their real durations and the side stream really lags) must leave the parameters, the last gradients and the
    losses bit-identical to (a) rounds 1-5's synchronisation -- default events, recorded forks, a wait per scratch claim, a hand-over
    per call -- and (b) the most eager setting (waits for the newest record).  A missing edge shows up here as a different bit."""
    new = _run_steps(str(tmp_path / 'new.pt'))
    old = _run_steps(str(tmp_path / 'old.pt'), CDRL_EVENT_FENCE=1, CDRL_TAIL_EVENTS=0, CDRL_SIDE_LAG=-1, PC_SEQ=0)
    lag0 = _run_steps(str(tmp_path / 'lag0.pt'), CDRL_SIDE_LAG=0, PC_SEQ=0)
    for other in (old, lag0):
        assert torch.equal(new['losses'], other['losses'])
        assert torch.equal(new['params'], other['params'])
        assert torch.equal(new['grads'], other['grads'])
    assert torch.isfinite(new['params']).all() and len(new['losses']) == 2 * (int(os.environ.get('PC_STEPS', 12)) // 4)     # (PC_STEPS: soak runs)


def _zero_gradient(name):
    """Parameters whose gradient is analytically zero (pure rounding residue in any implementation): biases in front of a
    train-mode BatchNorm, and the beta of a BatchNorm that is directly followed by another train-mode BatchNorm / conv + BN."""
    return is_degenerate_bias(name) or name.endswith('.bn2.beta') or name.endswith('.sc_bn1.beta') or name == 'dyn.bn.beta'


def _worst(a, b, floor=0.0, skip_zero_gradients=False):
    worst = {}
    for k in a:
        name = k.split('/', 1)[-1]
        if k == 'loss' or is_degenerate_bias(name) or (skip_zero_gradients and _zero_gradient(name)):
            continue
        x, y = a[k].double(), b[k].double()
        worst[k] = (x - y).abs().max().item() / max(y.abs().max().item(), floor, 1e-30)
    return worst


def test_fused_and_unfused_paths_agree_at_full_size(tmp_path):
    """(a0) every fusion on, float32-MFMA GEMMs  vs  (b) everything off: same arithmetic (k-ordered fmaf chains, double
    reductions) except the fused depthwise backward's xhat1, different kernels -> 5e-5 of every tensor's scale (measured worst
    1.3e-5: the stem BatchNorm's beta, the end of the backward chain; bit for bit until round 3)."""
    a0 = _run(str(tmp_path / 'fused.pt'), CDRL_PW_X3=0, CDRL_FUSED_BWD=0)
    b = _run(str(tmp_path / 'plain.pt'), CDRL_FUSED_DW=0, CDRL_FUSED_PW=0, CDRL_FUSED_STEM=0, CDRL_FUSED_PASS=0, CDRL_FUSED_BB=0,
             CDRL_SIDE_STREAM=0, CDRL_PW_X3=0)
    assert abs(a0['loss'].item() - b['loss'].item()) <= 1e-6 * max(1.0, abs(b['loss'].item()))
    # (the analytically-zero gradients are rounding residue of the others: since the fused depthwise backward takes xhat1 from the
    #  activated tile -- (a - beta) / gamma instead of (y1 - mean) invstd -- they are no longer bit-identical between the paths; they
    #  are held to "negligible next to the real gradients" instead)
    bad = {k: v for k, v in _worst(a0, b, skip_zero_gradients=True).items() if v > 5e-5}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
    gmax = max(v.abs().max().item() for k, v in b.items() if k.startswith('trunk/'))
    for k in a0:
        if k.startswith('trunk/') and _zero_gradient(k.split('/', 1)[-1]):
            assert a0[k].abs().max().item() < 1e-5 * gmax and b[k].abs().max().item() < 1e-5 * gmax, k


def test_fused_conv_backward_agrees_with_the_two_kernel_backward_at_full_size(tmp_path):
    """Round 4: backward-data + filter gradient + bias gradient + the BatchNorm-backward sums of the BatchNorm in front of the conv
    from ONE pass over the operands (gemm_pw_bwd.hip, three-way bf16 operand split) vs the float32-MFMA backward-data kernel on
    the critical stream + the filter-gradient GEMM on the side stream, at the benchmark shape.  Both runs take the SAME forward
    (CDRL_PW_X3=0: float32 MFMA), hence the same ReLU6 / max-pool decisions, so every gradient tensor must agree to float32
    accuracy: 5e-5 of the tensor's scale (measured worst 2.3e-5); the analytically-zero gradients are rounding residue in both."""
    a1 = _run(str(tmp_path / 'fbwd.pt'), CDRL_PW_X3=0)
    a0 = _run(str(tmp_path / 'two.pt'), CDRL_PW_X3=0, CDRL_FUSED_BWD=0)
    assert a1['loss'].item() == a0['loss'].item()
    assert torch.equal(a1['dyn'], a0['dyn'])
    w = _worst(a1, a0, skip_zero_gradients=True)
    worst = sorted(w.items(), key=lambda kv: -kv[1])[:6]
    print('[fused conv backward vs two-kernel backward] worst tensors:', worst)
    # Since round 5 the LAST conv of the backward (24 -> 58 channels on the pooled stem output) is fused too: the gradient it
    # accumulates into the pooled stem output then differs between the two runs at the 1e-6 level, and the stem BatchNorm's dbeta / dgamma
    # -- sums of that gradient over 16 M elements of both signs, the end of the chain -- see it amplified: measured 8.7e-5 / 1.1e-5 of
    # their scale between the two float32 paths.  Each path is held to the float64 oracle at north_star's 1e-4 elsewhere (smoke(),
    # test_pinned_decisions_*: measured <= 7e-5 for these two tensors); two such paths may sit up to twice that apart, so the gate
    # BETWEEN them is 2e-4 (a 1e-4 gate with 8.7e-5 measured would flake across boxes) and the measured value is recorded.
    stem_bn = {k: v for k, v in w.items() if k.startswith('trunk/img.stem.bn.')}
    print('[fused conv backward vs two-kernel backward] stem BatchNorm gradients between the two float32 paths:', stem_bn)
    import json, os
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/paths_fused_vs_two_kernel_backward.json', 'w') as f:
        json.dump(dict(worst=worst, stem_bn=stem_bn, gates=dict(stem_bn=2e-4, others=5e-5)), f, indent=1)
    bad = {k: v for k, v in w.items() if v >= (2e-4 if k.startswith('trunk/img.stem.bn.') else 5e-5)}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])
    gmax = max(v.abs().max().item() for k, v in a0.items() if k.startswith('trunk/'))
    for k in a1:        # the zero gradients stay negligible next to the real ones
        if k.startswith('trunk/') and _zero_gradient(k.split('/', 1)[-1]):
            assert a1[k].abs().max().item() < 1e-5 * gmax, k


def test_split_precision_convs_agree_with_float32_mfma_at_full_size(tmp_path):
    """Default engine (stage-1 forward 1x1 convs on the bf16 matrix pipe by exact three-way operand splitting,
    gemm_pw_x3.hip) vs the same engine on float32 MFMA, at the benchmark shape.  The split product is float32-accurate but
    not bit-identical, so among ~5e8 activations a few hundred ReLU6 / max-pool decisions flip -- as between ANY two float32
    implementations -- and the tower gradients then differ at the percent level (measured median 1e-2
    per tensor; the float32 torch oracle sits equally far from the float64 oracle).  What this test pins are the SMOOTH
    quantities: loss 1e-6, trunk output and BatchNorm moving statistics 1e-4, head and trunk-tail gradients 1e-4.  The gate
    on tower gradients (1e-4 with the decisions pinned) is tests/test_gpu_learner.py::test_pinned_decisions_*."""
    a = _run(str(tmp_path / 'x3.pt'))
    a0 = _run(str(tmp_path / 'f32.pt'), CDRL_PW_X3=0)
    assert abs(a['loss'].item() - a0['loss'].item()) <= 1e-6 * max(1.0, abs(a0['loss'].item()))
    w = _worst(a, a0, floor=1e-2, skip_zero_gradients=True)      # floor: bn3 moving means are ~1e-9 (zero-mean inputs, zero bias)
    assert w['dyn'] < 1e-4                                       # measured 1.4e-5
    assert max(v for k, v in w.items() if k.startswith('mv/')) < 1e-4
    w = _worst(a, a0, skip_zero_gradients=True)
    tail = {k: v for k, v in w.items() if not k.startswith('trunk/img.') and not k.startswith('mv/') and k != 'dyn'}
    assert max(tail.values()) < 1e-4, sorted(tail.items(), key=lambda kv: -kv[1])[:4]


def test_depthwise_backward_in_strip_form_agrees_with_the_pixel_mapped_form_at_full_size(tmp_path):
    """Round 5: the depthwise backward in strip form (dws_bwd_kernel / dws2_bwd_kernel: thread = channel pair x row strip, filter
    gradient in scatter form from the window of D, BN1 sums in float32 along a strip; the default) vs the pixel-mapped kernel of rounds
    1-4 (CDRL_DWS=0): same forward, hence the same decisions; every gradient tensor within 5e-5 (other summation orders)."""
    a1 = _run(str(tmp_path / 'strips.pt'))
    a0 = _run(str(tmp_path / 'pixels.pt'), CDRL_DWS=0)
    assert a1['loss'].item() == a0['loss'].item()
    assert torch.equal(a1['dyn'], a0['dyn'])
    w = _worst(a1, a0, skip_zero_gradients=True)
    worst = sorted(w.items(), key=lambda kv: -kv[1])[:4]
    assert worst[0][1] < 5e-5, worst
    assert any(not torch.equal(a1[k], a0[k]) for k in a1 if k.startswith('trunk/img.') and k.endswith('.dw.w'))     # the switch took effect


def test_band_staged_stem_forward_is_bit_identical_to_the_window_form_at_full_size(tmp_path):
    """Round 5: the stem conv + statistics kernel with the image band staged in LDS (stem_fwd_band_kernel, the default) vs the form whose
    threads fetch their pixels' windows from global memory (CDRL_STEM_FWD_BAND=0): the same fmaf chain per output element, and per-thread
    double statistics folded in another order -- the float statistics, hence every decision, the loss and every gradient, are bit-identical."""
    a1 = _run(str(tmp_path / 'band.pt'))
    a0 = _run(str(tmp_path / 'window.pt'), CDRL_STEM_FWD_BAND=0)
    assert a1['loss'].item() == a0['loss'].item()
    for k in a1:
        assert torch.equal(a1[k], a0[k]), k
