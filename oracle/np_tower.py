"""Second, independently written forward of the image tower in plain numpy / NHWC
(TEST INFRASTRUCTURE ONLY).  It shares no code with oracle/model.py (no torch conv,
no NCHW) and exists to cross-check the torch restatement: padding arithmetic, the
de-interleaving channel shuffle (F7), per-time-slice statistics (F6).

Follows core/architectures.py:30-173 slice by slice, literally: a python list of T
arrays goes through every shared layer, like the reference's list comprehensions.
"""
import numpy as np

from .spec import NetConfig, unit_plan

EPS = 1e-3


def _same(n, s):
    out = -(-n // s)
    tot = max((out - 1) * s + 3 - n, 0)
    return out, tot // 2


def _bn_train(xs, p, pre):
    out = []
    for x in xs:                                       # one call per time slice
        ax = tuple(range(x.ndim - 1))
        m = x.mean(axis=ax, dtype=np.float64)
        v = ((x - m) ** 2).mean(axis=ax, dtype=np.float64)
        out.append(((x - m) / np.sqrt(v + EPS)) * p[pre + '.gamma'] + p[pre + '.beta'])
    return out


def _relu6(xs):
    return [np.clip(x, 0.0, 6.0) for x in xs]


def _pw(xs, p, pre):
    w = p[pre + '.w'][0, 0].astype(np.float64)
    return [x @ w + p[pre + '.b'] for x in xs]


def _dw(xs, p, pre, s):
    w = p[pre + '.w'][..., 0].astype(np.float64)        # (3,3,C)
    out = []
    for x in xs:
        B, H, W, C = x.shape
        Ho, pt = _same(H, s)
        Wo, pl = _same(W, s)
        y = np.zeros((B, Ho, Wo, C))
        for i in range(3):
            for j in range(3):
                for oy in range(Ho):
                    iy = oy * s + i - pt
                    if iy < 0 or iy >= H:
                        continue
                    ox = np.arange(Wo)
                    ix = ox * s + j - pl
                    ok = (ix >= 0) & (ix < W)
                    y[:, oy, ox[ok], :] += x[:, iy, ix[ok], :] * w[i, j]
        out.append(y + p[pre + '.b'])
    return out


def _shuffle(x):
    C = x.shape[-1]
    out = np.empty_like(x)
    for b in range(2):
        for a in range(C // 2):
            out[..., b * (C // 2) + a] = x[..., 2 * a + b]      # F7
    return out


def tower_forward_np(image, p, cfg: NetConfig):
    """image (B,T,H,W,3) -> (T,B,last_channels) float64, training-mode BN (batch stats)."""
    p = {k: v.astype(np.float64) for k, v in p.items()}
    xs = [image[:, t].astype(np.float64) for t in range(cfg.T)]
    w = p['img.stem.conv.w']
    ys = []
    for x in xs:                                       # stem: 3x3, stride 2, valid
        B, H, W, _ = x.shape
        Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
        y = np.zeros((B, Ho, Wo, w.shape[-1]))
        for i in range(3):
            for j in range(3):
                y += x[:, i:i + 2 * Ho - 1:2, j:j + 2 * Wo - 1:2, :] @ w[i, j]
        ys.append(y + p['img.stem.conv.b'])
    xs = _relu6(_bn_train(ys, p, 'img.stem.bn'))
    ys = []
    for x in xs:                                       # maxpool 3x3 s2 same (-inf pad)
        B, H, W, C = x.shape
        Ho, pt = _same(H, 2)
        Wo, pl = _same(W, 2)
        xp = np.full((B, H + 3, W + 3, C), -np.inf)
        xp[:, pt:pt + H, pl:pl + W] = x
        y = np.full((B, Ho, Wo, C), -np.inf)
        for i in range(3):
            for j in range(3):
                y = np.maximum(y, xp[:, i:i + 2 * Ho - 1:2, j:j + 2 * Wo - 1:2])
        ys.append(y)
    xs = ys
    for u in unit_plan(cfg):
        pre = f"img.s{u['stage']}.u{u['unit']}"
        if u['stride'] == 1:
            sc = [x[..., :u['shortcut_c']] for x in xs]
            m = [x[..., u['shortcut_c']:] for x in xs]
        else:
            sc, m = xs, xs
        m = _relu6(_bn_train(_pw(m, p, pre + '.pw1'), p, pre + '.bn1'))
        m = _bn_train(_dw(m, p, pre + '.dw', u['stride']), p, pre + '.bn2')
        m = _relu6(_bn_train(_pw(m, p, pre + '.pw2'), p, pre + '.bn3'))
        if u['stride'] == 2:
            sc = _bn_train(_dw(sc, p, pre + '.sc_dw', 2), p, pre + '.sc_bn1')
            sc = _relu6(_bn_train(_pw(sc, p, pre + '.sc_pw'), p, pre + '.sc_bn2'))
        xs = [_shuffle(np.concatenate([a, b], axis=-1)) for a, b in zip(sc, m)]
    xs = _relu6(_bn_train(_pw(xs, p, 'img.head.conv'), p, 'img.head.bn'))
    return np.stack([x.mean(axis=(1, 2)) for x in xs], axis=0)
