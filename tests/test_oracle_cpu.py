"""CPU suite: oracle self-checks, structure pinned against the reference's checkpoint indices,
C-ABI library loads and exports every declared symbol (no compute without a GPU)."""
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import model as OM
from oracle.np_tower import tower_forward_np
from oracle.spec import NetConfig, trunk_spec, policy_spec, value_spec, count

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _inventory():
    return json.load(open(os.path.join(HERE, 'golden', 'ref_ckpt_inventory.json')))


def test_param_counts_match_reference_checkpoints():
    cfg = NetConfig()
    inv = _inventory()
    assert count(trunk_spec(cfg)) == inv['full']['dynamics_model']['total'] == 2145014
    assert count(policy_spec(cfg)) == inv['full']['policy_net']['total'] == 272134
    assert count(value_spec(cfg)) == inv['full']['value_net']['total'] == 271492
    for k, v in inv['totals'].items():       # all six shipped stages have the same structure
        assert v in (2145014, 272134, 271492), k


def test_shapes_match_reference_checkpoints():
    """Multiset of tensor shapes == the shapes stored in the reference's TF checkpoints."""
    cfg = NetConfig()
    inv = _inventory()
    for spec, key in ((trunk_spec(cfg), 'dynamics_model'), (policy_spec(cfg), 'policy_net'), (value_spec(cfg), 'value_net')):
        mine = sorted(tuple(s) for _, s, _, _ in spec)
        ref = sorted(tuple(s) for _, s in inv['full'][key]['tensors'])
        assert mine == ref, key


def test_policy_head_order_matches_checkpoint():
    # layer_with_weights-4..7 kernels are (320,2),(320,2),(320,1),(320,1): alpha, beta, similarity, speed
    inv = _inventory()['full']['policy_net']['tensors']
    kern = {int(re.search(r'-(\d+)/', k).group(1)): tuple(s) for k, s in inv if k.endswith('/kernel')}
    assert [kern[i] for i in (4, 5, 6, 7)] == [(320, 2), (320, 2), (320, 1), (320, 1)]
    spec = [s for n, s, _, _ in policy_spec(NetConfig()) if n.endswith('.w')]
    assert spec[-4:] == [(320, 2), (320, 2), (320, 1), (320, 1)]


def test_engine_inventory_equals_oracle_spec():
    from carla_driving_rl_agent_amd.engine import LearnerEngine
    for kw in (dict(), dict(A=3, vehicle=5, navigation=10, W=360)):
        cfg = NetConfig(**kw)
        eng = LearnerEngine(4, device=None, **kw)
        for m, spec in (('trunk', trunk_spec(cfg)), ('policy', policy_spec(cfg)), ('value', value_spec(cfg))):
            assert eng.tables[m].spec() == [(n, tuple(s), t) for n, s, _, t in spec]


def test_library_exports_every_declared_symbol():
    from carla_driving_rl_agent_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, 'include', 'cdrl.h')).read()
    declared = set(re.findall(r'\b(cdrl_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.cdrl_version() == 1


def test_torch_oracle_matches_independent_numpy_tower():
    cfg = NetConfig(H=41, W=58)
    tp = OM.init_params(trunk_spec(cfg), 5)
    img = np.random.default_rng(0).random((2, 4, 41, 58, 3), dtype=np.float32)
    p = OM.to_torch(tp, trunk_spec(cfg), torch.float64)
    with torch.no_grad():
        a = OM.shufflenet_v2(torch.tensor(img, dtype=torch.float64), p, cfg, True).numpy()
    b = tower_forward_np(img, tp, cfg)
    assert np.abs(a - b).max() < 1e-10


def test_channel_shuffle_is_deinterleave():
    x = torch.arange(8.0).reshape(1, 1, 8, 1, 1)
    assert OM.channel_shuffle(x).flatten().tolist() == [0, 2, 4, 6, 1, 3, 5, 7]       # F7


def test_same_padding_rule():
    assert OM.same_pad(22, 3, 2) == (0, 1) and OM.same_pad(11, 3, 2) == (1, 1) and OM.same_pad(30, 3, 1) == (1, 1)


def test_clip_by_norm_and_adam_known_answers():
    g = torch.tensor([3.0, 4.0])
    assert torch.allclose(OM.clip_by_norm(g, 1.0), torch.tensor([0.6, 0.8]))
    assert torch.allclose(OM.clip_by_norm(g, 10.0), g)
    p = {'w': torch.tensor([1.0])}
    opt = OM.Adam(['w'], p)
    opt.step(p, {'w': torch.tensor([0.5])}, lr=0.1)
    # first Adam step moves by ~lr regardless of the gradient scale
    assert abs(p['w'].item() - 0.9) < 1e-5


def test_fp32_oracle_close_to_fp64_oracle():
    """Headroom of the 1e-4 bar: the fp32 oracle itself sits ~1e-5 from an fp64 run."""
    from carla_driving_rl_agent_amd import synthetic
    cfg = NetConfig(H=48, W=64)
    tp = OM.init_params(trunk_spec(cfg), 1)
    pp = OM.init_params(policy_spec(cfg), 2)
    vp = OM.init_params(value_spec(cfg), 3)
    r = synthetic.make_rollout(4, H=48, W=64)
    outs = []
    for dt in (torch.float32, torch.float64):
        L = OM.OracleLearner(cfg, tp, pp, vp, synthetic.DEFAULT_HP, dtype=dt)
        a, b, v, d = L.predict(r['states'])
        outs.append(d.double().numpy())
    assert np.abs(outs[0] - outs[1]).max() / np.abs(outs[1]).max() < 1e-4


def _tiny_problem(seed=4, B=5, H=35, W=37, A=2):
    from carla_driving_rl_agent_amd import synthetic
    cfg = NetConfig(H=H, W=W, A=A)
    tp = OM.init_params(trunk_spec(cfg), seed + 1, dtype=np.float64)
    pp = OM.init_params(policy_spec(cfg), seed + 2, dtype=np.float64)
    vp = OM.init_params(value_spec(cfg), seed + 3, dtype=np.float64)
    r = synthetic.make_rollout(B, H=H, W=W, A=A, seed=seed)
    rng = np.random.default_rng(seed)
    pol = dict(states=r['states'], advantages=rng.standard_normal(B), old_log_prob=r['old_log_prob'].astype(np.float64),
               speed=r['speed'] / 100.0, similarity=r['similarity'].astype(np.float64), u=r['action'].astype(np.float64),
               du_da=rng.uniform(-0.2, 0.2, (B, A)), du_db=rng.uniform(-0.2, 0.2, (B, A)))
    val = dict(states=r['states'], returns=r['value'].astype(np.float64), speed=pol['speed'], similarity=pol['similarity'])
    learner = OM.OracleLearner(cfg, tp, pp, vp, dict(synthetic.DEFAULT_HP), dtype=torch.float64)
    return cfg, learner, pol, val, (tp, pp, vp)


def test_second_numpy_implementation_of_tail_heads_and_losses():
    """oracle/np_model.py (numpy / scipy, no shared code) reproduces the torch restatement's trunk output, alpha / beta /
    log-prob / entropy / ratio, values and both total losses to float64 round-off -- every forward quantity of the hot path
    now has two independent implementations (tower: oracle/np_tower.py)."""
    from oracle import np_model as NM
    cfg, learner, pol, val, (tp, pp, vp) = _tiny_problem()
    st = {k: torch.as_tensor(v, dtype=torch.float64) for k, v in pol['states'].items()}
    taps = {}
    with torch.no_grad():
        d_ref = OM.dynamics_forward(st, learner.trunk, cfg, True, taps)
    d = NM.dynamics_np(taps['img_feat'].numpy(), pol['states'], tp)
    assert np.abs(d - d_ref.numpy()).max() < 1e-10 * max(1.0, np.abs(d).max())
    feat = tower_forward_np(pol['states']['state_image'], tp, cfg)
    assert np.abs(feat - taps['img_feat'].numpy()).max() < 1e-10
    loss, gp, gt, aux = learner.policy_grads(pol)
    out = NM.policy_objective_np(d, pp, pol, learner.hp['clip_ratio'], learner.hp['entropy_coef'])
    assert abs(out['loss'] - float(loss.detach())) < 1e-10 * max(1.0, abs(float(loss.detach())))
    for k in ('alpha', 'beta', 'log_prob', 'ratio'):
        assert np.abs(out[k] - aux[k].detach().numpy()).max() < 1e-9, k
    assert abs(out['entropy'] - float(aux['entropy'])) < 1e-10
    vloss, gv, gt2, vaux = learner.value_grads(val)
    vout = NM.value_objective_np(d, vp, val)
    assert abs(vout['loss'] - float(vloss)) < 1e-10 * max(1.0, abs(float(vloss)))
    assert np.abs(vout['values'] - vaux['values'].detach().numpy()).max() < 1e-10


@pytest.mark.parametrize('which', ['policy', 'value'])
def test_oracle_gradients_match_float64_finite_differences(which):
    """Pins the BACKWARD of the torch restatement (autograd through per-slice BatchNorm, the GRUs, the injected pathwise Beta
    sample, both objectives): central finite differences of the float64 loss along random directions, one direction per
    parameter tensor group, with the ReLU6 / max-pool decisions replayed (oracle.model.Decisions) so that the loss is a
    smooth function along the probe."""
    cfg, learner, pol, val, _ = _tiny_problem(seed=9)
    batch = pol if which == 'policy' else val
    grads_fn = learner.policy_grads if which == 'policy' else learner.value_grads
    head = learner.policy if which == 'policy' else learner.value
    OM.DEC.start('record')
    loss0, gh, gt, _ = grads_fn(batch)
    OM.DEC.start('off')
    recorded = OM.DEC.items

    def loss_at():
        OM.DEC.items = recorded
        OM.DEC.start('replay')
        try:
            with torch.no_grad():
                st = {k: torch.as_tensor(v, dtype=torch.float64) for k, v in batch['states'].items()}
                d = OM.dynamics_forward(st, learner.trunk, cfg, True)
                b = learner._cast(batch)
                if which == 'policy':
                    # the injected sample is a first-order model u(alpha, beta) = u0 + J (theta - theta0): reproduce it
                    al, be, _, _ = OM.policy_heads(d, learner.policy, True)
                    b = dict(b, u=b['u'] + b['du_da'] * (al - alpha0) + b['du_db'] * (be - beta0))
                    return float(OM.policy_objective(d, learner.policy, b, learner.hp)[0])
                return float(OM.value_objective(d, learner.value, b)[0])
        finally:
            OM.DEC.start('off')

    if which == 'policy':
        with torch.no_grad():
            OM.DEC.items = recorded
            OM.DEC.start('replay')
            st = {k: torch.as_tensor(v, dtype=torch.float64) for k, v in batch['states'].items()}
            alpha0, beta0, _, _ = OM.policy_heads(OM.dynamics_forward(st, learner.trunk, cfg, True), learner.policy, True)
            OM.DEC.start('off')
    assert abs(loss_at() - float(loss0)) < 1e-12 * max(1.0, abs(float(loss0)))
    rng = np.random.default_rng(0)
    groups = {'tower-early': [n for n in gt if n.startswith('img.s0') or n.startswith('img.stem')],
              'tower-late': [n for n in gt if n.startswith('img.s2') or n.startswith('img.head')],
              'feature-nets': [n for n in gt if n.split('.')[0] in ('road', 'vehicle', 'navigation')],
              'grus': [n for n in gt if n.startswith('gru_')], 'tail': [n for n in gt if n.startswith('dyn.')], 'head': list(gh)}
    for gname, names in groups.items():
        params, grads = (head, gh) if gname == 'head' else (learner.trunk, gt)
        dirs = {n: torch.as_tensor(rng.standard_normal(tuple(params[n].shape))) for n in names}
        analytic = sum(float((grads[n] * dirs[n]).sum()) for n in names)
        h = 1e-7        # the truncation error falls as h^2 (4e-2 at 1e-6, 4e-4 at 1e-7 on the stem group); round-off ~3e-9
        with torch.no_grad():
            for n in names:
                params[n].add_(h * dirs[n])
            lp = loss_at()
            for n in names:
                params[n].sub_(2 * h * dirs[n])
            lm = loss_at()
            for n in names:
                params[n].add_(h * dirs[n])
        fd = (lp - lm) / (2 * h)
        scale = max(abs(analytic), 1e-3 * sum(float((grads[n] ** 2).sum()) for n in names) ** 0.5)
        assert abs(fd - analytic) < 5e-6 * scale + 1e-8, (gname, fd, analytic)
