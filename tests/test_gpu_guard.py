"""Out-of-bounds canaries of the learner workspace (sanitizer row).  The kernels address a
bump-allocated multi-GB workspace through raw buffer descriptors and there is no GPU AddressSanitizer on this pool: with CDRL_GUARD=1
every workspace tensor is followed by a 64 KB band filled with a pattern at bind; after a full-size update-step (both passes, both
optimizer steps, the on-device re-sampling, an inference forward) in float32 AND with bf16 activation storage every band must be
intact -- and a band that is written must be reported with its offset."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(*args, **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, os.path.join(HERE, 'guard_bands.py'), *[str(a) for a in args]], env=e, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('bands ')][-1].split()
    return int(line[1]), int(line[2]), int(line[3]), int(line[4])


@pytest.mark.parametrize('compute', ['f32', 'bf16s'])
def test_every_guard_band_is_intact_after_a_full_size_update_step(compute):
    bad, first, ws, _ = _run('check', compute, CDRL_GUARD=1)
    assert (bad, first) == (0, -1), (bad, first)
    assert ws > 5 * 2 ** 30 * (0.4 if compute == 'bf16s' else 1.0) * 0.5        # the full-size plan, bands included


def test_a_written_band_is_reported_and_the_switch_is_needed():
    for which in (0, 5):        # the first band of the workspace and a later one
        bad, first, _, poked = _run('poke', 'f32', which, CDRL_GUARD=1, GB_B=4, GB_H=48, GB_W=64)
        assert bad == 1 and first <= poked < first + 65536, (bad, first, poked)
    # without the switch the call fails loudly (no silent "0 bad bands")
    e = dict(os.environ)
    e.pop('CDRL_GUARD', None)
    r = subprocess.run([sys.executable, os.path.join(HERE, 'guard_bands.py'), 'check', 'f32'], env=dict(e, GB_B='4', GB_H='48', GB_W='64'),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and 'CDRL_GUARD' in r.stderr
