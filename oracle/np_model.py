"""Second, independently written forward of everything BEHIND the image tower, in plain numpy / scipy float64
(TEST INFRASTRUCTURE ONLY; parity unpinned at the TF boundary like the rest of oracle/).

Shares no code with oracle/model.py: feature nets, the Keras GRU cell written per gate with explicit loops over time,
trunk tail, control branches, Beta log-density / entropy through scipy.special (gammaln, digamma), both objectives.
Together with oracle/np_tower.py it cross-checks every forward quantity of the torch restatement; the backward of the
torch restatement is pinned separately by a float64 finite-difference check (tests/test_oracle_cpu.py).

Follows: core/architectures.py:9-27 (feature_net), core/networks.py:24-30,37-56 (dynamics_layers), :59-66 (control_branch),
:115-137 (policy heads), :255-275 (value heads), :96-110,139-144 (PolicyNetwork.call, _clip_actions),
core/carla_agent.py:394-428 (policy_objective), :469-486 (value_objective); TF/TFP defaults from SURVEY.md Appendix A.
"""
import numpy as np
from scipy import special

EPS_BN = 1e-3
F32_EPS = float(np.finfo(np.float32).eps)


def _bn_rows(x, p, pre):
    """Training-mode BatchNormalization of one (rows, C) slice: biased variance, epsilon 1e-3."""
    m = x.mean(axis=0)
    v = ((x - m) ** 2).mean(axis=0)
    return (x - m) / np.sqrt(v + EPS_BN) * p[pre + '.gamma'] + p[pre + '.beta']


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def feature_net_np(v, p, name):
    """v (B,T,D) -> list of T arrays (B,16): [Dense(16, relu6) -> BN] x 2, one call per time slice."""
    out = []
    for t in range(v.shape[1]):
        x = v[:, t].astype(np.float64)
        for i in range(2):
            x = np.minimum(np.maximum(x @ p[f'{name}.fc{i}.w'] + p[f'{name}.fc{i}.b'], 0.0), 6.0)
            x = _bn_rows(x, p, f'{name}.bn{i}')
        out.append(x)
    return out


def gru_np(xs, p, name):
    """Keras GRU v2 (reset_after=True): xs = list of T arrays (B,In) -> last hidden state (B,u)."""
    K, R, b = p[name + '.kernel'], p[name + '.recurrent'], p[name + '.bias']
    u = R.shape[0]
    h = np.zeros((xs[0].shape[0], u))
    for x in xs:
        xz, xr, xh = np.split(x @ K + b[0], 3, axis=1)
        hz, hr, hh = np.split(h @ R + b[1], 3, axis=1)
        z = _sigmoid(xz + hz)
        r = _sigmoid(xr + hr)
        cand = np.tanh(xh + r * hh)
        h = z * h + (1.0 - z) * cand
    return h


def dynamics_np(img_feat, states, p):
    """img_feat (T,B,last) from np_tower.tower_forward_np; states dict of (B,T,D) -> (B,512)."""
    hs = [gru_np([img_feat[t] for t in range(img_feat.shape[0])], p, 'gru_image')]
    for name in ('road', 'vehicle', 'navigation'):
        hs.append(gru_np(feature_net_np(states[f'state_{name}'], p, name), p, f'gru_{name}'))
    cat = np.concatenate(hs, axis=1)
    return _bn_rows(cat, p, 'dyn.bn') @ p['dyn.fc.w'] + p['dyn.fc.b']


def _branch(d, p, pre):
    x = d
    for i in range(2):
        x = _bn_rows(x, p, f'{pre}.bn{i}')
        x = x @ p[f'{pre}.fc{i}.w'] + p[f'{pre}.fc{i}.b']
        x = np.minimum(x * _sigmoid(x), 6.0)
    return x


def _softplus101(x):
    return np.logaddexp(0.0, x) + 1.01


def policy_objective_np(d, p, batch, clip_ratio, entropy_coef):
    """-> dict(loss, alpha, beta, log_prob, entropy, ratio)."""
    h = _branch(d, p, 'pi')
    al = _softplus101(h @ p['pi.alpha.w'] + p['pi.alpha.b'])
    be = _softplus101(h @ p['pi.beta.w'] + p['pi.beta.b'])
    sim = np.tanh(h @ p['pi.similarity.w'] + p['pi.similarity.b'])
    spd = 2.0 * _sigmoid(h @ p['pi.speed.w'] + p['pi.speed.b'])
    x = np.clip(batch['u'].astype(np.float64), F32_EPS, 1.0 - F32_EPS)
    lnB = special.gammaln(al) + special.gammaln(be) - special.gammaln(al + be)
    logp = (al - 1.0) * np.log(x) + (be - 1.0) * np.log1p(-x) - lnB
    ent = lnB - (al - 1.0) * special.digamma(al) - (be - 1.0) * special.digamma(be) + (al + be - 2.0) * special.digamma(al + be)
    ratio = np.exp(logp - batch['old_log_prob']).mean(axis=1)
    adv = batch['advantages'].astype(np.float64)
    bound = np.where(adv > 0, (1.0 + clip_ratio) * adv, (1.0 - clip_ratio) * adv)
    l_pi = -np.minimum(ratio * adv, bound).mean()
    l_speed = 0.5 * ((batch['speed'] - spd) ** 2).mean()
    l_sim = 0.5 * ((batch['similarity'] - sim) ** 2).mean()
    return dict(loss=l_pi - entropy_coef * ent.mean() + l_speed + l_sim, alpha=al, beta=be, log_prob=logp, entropy=ent.mean(),
                ratio=ratio)


def value_objective_np(d, p, batch, exp_scale=6.0):
    h = _branch(d, p, 'v')
    base = np.tanh(h @ p['v.base.w'] + p['v.base.b'])[:, 0]
    ex = exp_scale * _sigmoid(h @ p['v.exp.w'] + p['v.exp.b'])[:, 0]
    spd = 2.0 * _sigmoid(h @ p['v.speed.w'] + p['v.speed.b'])
    sim = np.tanh(h @ p['v.similarity.w'] + p['v.similarity.b'])
    ret = batch['returns'].astype(np.float64)
    l_value = 0.25 * ((ret[:, 0] - base) ** 2).mean() + ((ret[:, 1] - ex) ** 2).mean() / exp_scale ** 2
    l_speed = ((batch['speed'] - spd) ** 2).mean()
    l_sim = ((batch['similarity'] - sim) ** 2).mean()
    return dict(loss=0.25 * (l_value + l_speed + l_sim), values=np.stack([base, ex], axis=1))
