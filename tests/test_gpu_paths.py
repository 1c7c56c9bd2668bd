"""Full-size consistency of the fused kernels with the unfused ones (which the small-size oracle parity tests also
cover): the same policy pass at the benchmark shape (B=256, T=4, 90x120x3) through (a) the default engine -- fused
depthwise block, persistent pointwise GEMMs with BatchNorm prologues / epilogues, BN-backward operand prologue, fused stem
block, identity half riding on the BatchNorm ops, side / aux streams -- and (b) the engine with every fusion and the side
stream switched off, in two subprocesses (the switches are per process).  Every reduction on the path accumulates in
double over float data and every GEMM is a k-ordered fmaf chain, so the two paths agree to float rounding -- in practice
bit for bit -- everywhere except on the analytically-zero bias gradients (pure rounding noise, tests/util.py)."""
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


def test_fused_and_unfused_paths_agree_at_full_size(tmp_path):
    a = _run(str(tmp_path / 'fused.pt'))
    b = _run(str(tmp_path / 'plain.pt'), CDRL_FUSED_DW=0, CDRL_FUSED_PW=0, CDRL_FUSED_STEM=0, CDRL_FUSED_PASS=0, CDRL_FUSED_BB=0,
             CDRL_SIDE_STREAM=0)
    assert abs(a['loss'].item() - b['loss'].item()) <= 1e-6 * max(1.0, abs(b['loss'].item()))
    worst = {}
    for k in a:
        if k == 'loss' or is_degenerate_bias(k.split('/', 1)[-1]):
            continue
        x, y = a[k].double(), b[k].double()
        worst[k] = (x - y).abs().max().item() / (y.abs().max().item() + 1e-30)
    bad = {k: v for k, v in worst.items() if v > 1e-5}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
