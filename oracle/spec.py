"""Oracle-side architecture + parameter inventory (TEST INFRASTRUCTURE ONLY).

Independent restatement of the variable inventory created by the reference's
Keras graph builders:
  * core/architectures.py:30-173   (shufflenet_v2, time-distributed, shared weights)
  * core/architectures.py:9-27     (feature_net)
  * core/networks.py:37-56         (dynamics_layers: 4 GRUs, concat, BN, Dense 512)
  * core/networks.py:59-66,115-137 (control_branch, policy heads)
  * core/networks.py:255-275       (value branch / heads)
Shapes use the Keras layouts: Conv2D (kh,kw,Cin,Cout), DepthwiseConv2D (3,3,C,1),
Dense (in,out), GRU kernel (in,3u) / recurrent (u,3u) / bias (2,3u), BN 4x(C,).
"""
from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass
class NetConfig:
    T: int = 4                  # env.time_horizon (core/carla_env.py:26-27)
    H: int = 90
    W: int = 120                # env image width (3*W' for the three-camera env)
    road: int = 9
    vehicle: int = 4
    navigation: int = 5
    A: int = 2                  # num_actions
    stage_channels: Tuple[int, int, int] = (116, 232, 464)   # g=1.0, core/architectures.py:33-34
    stage_blocks: Tuple[int, int, int] = (4, 8, 4)           # core/architectures.py:165-167
    stem_channels: int = 24
    last_channels: int = 768    # core/carla_agent.py:66
    feat_units: int = 16
    rnn_image: int = 256
    rnn_small: int = 32
    dyn_units: int = 512
    head_units: int = 320


# (name, shape, init, trainable)
Spec = Tuple[str, Tuple[int, ...], str, bool]


def _bn(prefix: str, c: int) -> List[Spec]:
    return [(f'{prefix}.gamma', (c,), 'ones', True), (f'{prefix}.beta', (c,), 'zeros', True),
            (f'{prefix}.moving_mean', (c,), 'zeros', False), (f'{prefix}.moving_var', (c,), 'ones', False)]


def _conv(prefix: str, k: int, cin: int, cout: int) -> List[Spec]:
    return [(f'{prefix}.w', (k, k, cin, cout), 'glorot', True), (f'{prefix}.b', (cout,), 'zeros', True)]


def _dw(prefix: str, c: int) -> List[Spec]:
    return [(f'{prefix}.w', (3, 3, c, 1), 'glorot', True), (f'{prefix}.b', (c,), 'zeros', True)]


def _dense(prefix: str, cin: int, cout: int, bias='glorot') -> List[Spec]:
    return [(f'{prefix}.w', (cin, cout), 'glorot', True), (f'{prefix}.b', (cout,), bias, True)]


def unit_plan(cfg: NetConfig):
    """List of (stage, unit, stride, cin, mid, main_out, shortcut_c) following
    shufflenet_v2_unit (core/architectures.py:120-145)."""
    plan = []
    cin = cfg.stem_channels
    for s, (c, nb) in enumerate(zip(cfg.stage_channels, cfg.stage_blocks)):
        for u in range(nb):
            stride = 2 if u == 0 else 1
            if stride == 2:
                shortcut_c = cin
                main_in = cin
            else:
                shortcut_c = cin // 2
                main_in = cin - cin // 2
            mid = c // 2
            main_out = c - shortcut_c
            plan.append(dict(stage=s, unit=u, stride=stride, cin=cin, main_in=main_in, mid=mid,
                             main_out=main_out, shortcut_c=shortcut_c, cout=c))
            cin = c
    return plan


def tower_spec(cfg: NetConfig) -> List[Spec]:
    sp: List[Spec] = []
    sp += _conv('img.stem.conv', 3, 3, cfg.stem_channels)
    sp += _bn('img.stem.bn', cfg.stem_channels)
    for p in unit_plan(cfg):
        pre = f"img.s{p['stage']}.u{p['unit']}"
        sp += _conv(f'{pre}.pw1', 1, p['main_in'], p['mid'])
        sp += _bn(f'{pre}.bn1', p['mid'])
        sp += _dw(f'{pre}.dw', p['mid'])
        sp += _bn(f'{pre}.bn2', p['mid'])
        sp += _conv(f'{pre}.pw2', 1, p['mid'], p['main_out'])
        sp += _bn(f'{pre}.bn3', p['main_out'])
        if p['stride'] == 2:
            sp += _dw(f'{pre}.sc_dw', p['shortcut_c'])
            sp += _bn(f'{pre}.sc_bn1', p['shortcut_c'])
            sp += _conv(f'{pre}.sc_pw', 1, p['shortcut_c'], p['shortcut_c'])
            sp += _bn(f'{pre}.sc_bn2', p['shortcut_c'])
    sp += _conv('img.head.conv', 1, cfg.stage_channels[-1], cfg.last_channels)
    sp += _bn('img.head.bn', cfg.last_channels)
    return sp


def trunk_spec(cfg: NetConfig) -> List[Spec]:
    sp = tower_spec(cfg)
    for name, dim in (('road', cfg.road), ('vehicle', cfg.vehicle), ('navigation', cfg.navigation)):
        sp += _dense(f'{name}.fc0', dim, cfg.feat_units)
        sp += _bn(f'{name}.bn0', cfg.feat_units)
        sp += _dense(f'{name}.fc1', cfg.feat_units, cfg.feat_units)
        sp += _bn(f'{name}.bn1', cfg.feat_units)
    for name, cin, u in (('image', cfg.last_channels, cfg.rnn_image), ('road', cfg.feat_units, cfg.rnn_small),
                         ('vehicle', cfg.feat_units, cfg.rnn_small), ('navigation', cfg.feat_units, cfg.rnn_small)):
        sp += [(f'gru_{name}.kernel', (cin, 3 * u), 'glorot', True),
               (f'gru_{name}.recurrent', (u, 3 * u), 'orthogonal', True),
               (f'gru_{name}.bias', (2, 3 * u), 'glorot', True)]
    cat = cfg.rnn_image + 3 * cfg.rnn_small
    sp += _bn('dyn.bn', cat)
    sp += _dense('dyn.fc', cat, cfg.dyn_units)
    return sp


def branch_spec(prefix: str, cfg: NetConfig) -> List[Spec]:
    sp: List[Spec] = []
    sp += _bn(f'{prefix}.bn0', cfg.dyn_units)
    sp += _dense(f'{prefix}.fc0', cfg.dyn_units, cfg.head_units)
    sp += _bn(f'{prefix}.bn1', cfg.head_units)
    sp += _dense(f'{prefix}.fc1', cfg.head_units, cfg.head_units)
    return sp


def policy_spec(cfg: NetConfig) -> List[Spec]:
    sp = branch_spec('pi', cfg)
    # core/networks.py:133-134: alpha/beta Dense have default (zeros) bias init
    sp += _dense('pi.alpha', cfg.head_units, cfg.A, bias='zeros')
    sp += _dense('pi.beta', cfg.head_units, cfg.A, bias='zeros')
    sp += _dense('pi.similarity', cfg.head_units, 1)
    sp += _dense('pi.speed', cfg.head_units, 1)
    return sp


def value_spec(cfg: NetConfig) -> List[Spec]:
    sp = branch_spec('v', cfg)
    sp += _dense('v.base', cfg.head_units, 1)
    sp += _dense('v.exp', cfg.head_units, 1)
    sp += _dense('v.speed', cfg.head_units, 1)
    sp += _dense('v.similarity', cfg.head_units, 1)
    return sp


def count(spec: List[Spec], trainable=None) -> int:
    n = 0
    for _, shape, _, tr in spec:
        if trainable is None or tr == trainable:
            k = 1
            for d in shape:
                k *= d
            n += k
    return n
