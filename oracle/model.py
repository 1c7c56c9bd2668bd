"""PyTorch-CPU restatement of the reference learner (TEST INFRASTRUCTURE ONLY).

PARITY STATUS: parity unpinned at the TF boundary (see oracle/__init__.py).

Every function cites the reference lines it follows; TF/TFP defaults are the
ones written out in SURVEY.md Appendix A.  The whole time axis is carried as a
leading dimension (T, B, ...) but every BatchNorm computes statistics **per
time slice**, exactly as the reference's python list-comprehension over the T
tensors does (core/architectures.py:44-85).
"""
import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .spec import NetConfig, trunk_spec, policy_spec, value_spec, unit_plan

BN_EPS = 1e-3          # Keras BatchNormalization default epsilon
BN_MOMENTUM = 0.99     # Keras BatchNormalization default momentum
EPSILON = float(np.finfo(np.float32).eps)   # rl/utils.py:24-25


# ------------------------------------------------------------------------------------------------
# parameters
# ------------------------------------------------------------------------------------------------

def init_params(spec, seed: int, randomize_bn: bool = True, dtype=np.float32) -> Dict[str, np.ndarray]:
    """Deterministic initial weights (weights are an explicit *input* of the parity
    contract).  glorot-uniform kernels and (where the
    reference says bias_initializer='glorot_uniform') biases, orthogonal GRU
    recurrent kernels.  ``randomize_bn`` perturbs gamma/beta/moving stats so that
    tests exercise them (Keras defaults are 1/0/0/1)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape, init, _ in spec:
        if init == 'glorot':
            if len(shape) == 4:          # conv (kh,kw,cin,cout); depthwise (3,3,C,1)
                rf = shape[0] * shape[1]
                fan_in, fan_out = shape[2] * rf, shape[3] * rf
            elif len(shape) == 2:
                fan_in, fan_out = shape
            else:
                fan_in = fan_out = shape[0]
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            w = rng.uniform(-lim, lim, size=shape)
        elif init == 'orthogonal':
            a = rng.standard_normal(size=(max(shape), max(shape)))
            q, r = np.linalg.qr(a)
            q = q * np.sign(np.diag(r))
            w = q[:shape[0], :shape[1]]
        elif init == 'zeros':
            w = np.zeros(shape)
            if randomize_bn and (name.endswith('.beta') or name.endswith('.moving_mean')):
                w = rng.uniform(-0.2, 0.2, size=shape)
            elif randomize_bn and name.endswith('.b'):
                w = rng.uniform(-0.05, 0.05, size=shape)
        elif init == 'ones':
            w = np.ones(shape)
            if randomize_bn:
                w = rng.uniform(0.7, 1.3, size=shape)
        else:
            raise ValueError(init)
        out[name] = np.ascontiguousarray(w, dtype=dtype)
    return out


def to_torch(params: Dict[str, np.ndarray], spec, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    out = {}
    for name, _, _, trainable in spec:
        t = torch.tensor(params[name], dtype=dtype)
        t.requires_grad_(trainable)
        out[name] = t
    return out


# ------------------------------------------------------------------------------------------------
# primitive layers
# ------------------------------------------------------------------------------------------------

def same_pad(n: int, k: int, s: int):
    """TF 'SAME' padding: pad_before = floor(total/2), pad_after = rest."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


class Decisions:
    """Record / replay of the DISCRETE decisions of a forward pass (ReLU6 regions, max-pool argmax).

    Between two float32 implementations (or float32 and float64) an element within rounding distance of a
    ReLU6 kink or of a max-pool tie takes a different branch, and the gradient changes by an O(1) amount per
    flipped element.  A gradient comparison is only well defined when both sides took the same decisions:
    `record` stores them during one forward, `replay` forces them onto another forward (e.g. the float64
    oracle evaluated with the float32 decisions), which then is a smooth function of its inputs."""

    def __init__(self):
        self.mode = 'off'
        self.items = []
        self.cursor = 0

    def start(self, mode):
        assert mode in ('off', 'record', 'replay')
        self.mode = mode
        self.cursor = 0
        if mode == 'record':
            self.items = []
        return self

    def put(self, item):
        self.items.append(item)

    def get(self):
        item = self.items[self.cursor]
        self.cursor += 1
        return item


DEC = Decisions()


def relu6(x):
    """ReLU(max_value=6), core/architectures.py:47 (gradient 1 strictly inside (0, 6))."""
    if DEC.mode == 'replay':
        mid, hi = DEC.get()
        return x * mid.to(x.dtype) + 6.0 * hi.to(x.dtype)
    if DEC.mode == 'record':
        DEC.put(((x > 0.0) & (x < 6.0), x >= 6.0))
    return torch.clamp(x, 0.0, 6.0)


def max_pool_3x3_s2(xx):
    """MaxPool2D(3, 2, 'same') on an input already padded with -inf (core/architectures.py:161)."""
    if DEC.mode == 'replay':
        idx = DEC.get()
        flat = xx.reshape(xx.shape[0], xx.shape[1], -1)
        return torch.gather(flat, 2, idx.reshape(idx.shape[0], idx.shape[1], -1)).reshape(idx.shape)
    if DEC.mode == 'record':
        y, idx = F.max_pool2d(xx, 3, 2, return_indices=True)
        DEC.put(idx)
        return y
    return F.max_pool2d(xx, 3, 2)


def swish6(x):
    return torch.minimum(x * torch.sigmoid(x), torch.tensor(6.0, dtype=x.dtype))   # rl/utils.py:420-421


def bn_slices(x, p, prefix, training: bool, bessel: bool):
    """BatchNormalization shared over T time slices, applied slice by slice
    (core/architectures.py:44-57).  x: (T, B, C, ...) channels at dim 2."""
    gamma, beta = p[f'{prefix}.gamma'], p[f'{prefix}.beta']
    mm, mv = p[f'{prefix}.moving_mean'], p[f'{prefix}.moving_var']
    red = [1] + list(range(3, x.dim()))
    shape = [1, 1, -1] + [1] * (x.dim() - 3)
    if training:
        mean = x.mean(dim=red, keepdim=True)
        var = ((x - mean) ** 2).mean(dim=red, keepdim=True)
        y = (x - mean) * torch.rsqrt(var + BN_EPS) * gamma.view(shape) + beta.view(shape)
        with torch.no_grad():
            n = x.numel() // (x.shape[0] * x.shape[2])
            corr = n / (n - 1.0) if (bessel and n > 1) else 1.0
            for t in range(x.shape[0]):        # T sequential EMA updates (F6)
                mm.mul_(BN_MOMENTUM).add_((1.0 - BN_MOMENTUM) * mean[t].reshape(-1))
                mv.mul_(BN_MOMENTUM).add_((1.0 - BN_MOMENTUM) * corr * var[t].reshape(-1))
        return y
    return (x - mm.view(shape)) * torch.rsqrt(mv.view(shape) + BN_EPS) * gamma.view(shape) + beta.view(shape)


def _fold(x):          # (T,B,...) -> (T*B,...)
    return x.reshape((x.shape[0] * x.shape[1],) + tuple(x.shape[2:]))


def _unfold(x, T):
    return x.reshape((T, x.shape[0] // T) + tuple(x.shape[1:]))


def _bf16_round(t):
    """Round-to-nearest-even to bf16, kept in t's dtype (float64 inputs pass through float32 first, as the engine's operands
    are float32 numbers)."""
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


class _PwBf16Operands(torch.autograd.Function):
    """1x1 convolution in the engine's bf16-OPERAND compute mode (include/cdrl.h CDRL_COMPUTE_BF16_OPERANDS; BASELINE.json
    configs[2]): forward y = bf(x) bf(W) + b, backward-data dx = bf(dy) bf(W)^T, filter gradient dW = bf(x)^T bf(dy), all
    accumulated in the working dtype; the bias gradient is the plain sum of dy (a float32 reduction in the engine)."""

    @staticmethod
    def forward(ctx, x, w, b):           # x (N,Cin,H,W), w (Cin,Cout), b (Cout)
        ctx.save_for_backward(x, w)
        return torch.einsum('nchw,cd->ndhw', _bf16_round(x), _bf16_round(w)) + b.view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = torch.einsum('ndhw,cd->nchw', _bf16_round(dy), _bf16_round(w))
        dw = torch.einsum('nchw,ndhw->cd', _bf16_round(x), _bf16_round(dy))
        return dx, dw, dy.sum(dim=(0, 2, 3))


PW_BF16_OPERANDS = False        # set by the bf16-mode parity tests (tests/test_gpu_bf16.py); module state like DEC
# bf16 ACTIVATION STORAGE (include/cdrl.h CDRL_COMPUTE_BF16_STORAGE, BASELINE.json configs[2]): every activation tensor of the image
# tower that the engine keeps in HBM is rounded to bf16 where it is stored, and so is every activation GRADIENT it stores; what is
# recomputed on load (BatchNorm apply / backward apply, ReLU6) stays in the working precision.  Set together with PW_BF16_OPERANDS.
BF16_STORAGE = False
# ablation switches of the storage rule (tests/test_oracle_bf16_ablation.py): round only the stored ACTIVATIONS, or only the stored
# activation GRADIENTS
BF16_STORE_FWD = True
BF16_STORE_BWD = True


class _Store(torch.autograd.Function):
    """A storage point: forward value rounded to bf16 when `fwd`, incoming gradient rounded to bf16 when `bwd`."""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return _bf16_round(x) if fwd else x

    @staticmethod
    def backward(ctx, g):
        return (_bf16_round(g) if ctx.bwd else g), None, None


def _st(x, fwd, bwd):
    """Storage point of the bf16-storage contract (no-op outside that mode).  Which tensors are stored, and which of their
    gradients, follows the engine's workspace plan (csrc/engine.hip::build_trunk):
      fwd: raw conv / depthwise outputs (the BatchNorm inputs), the max-pool output, every unit output, the shortcut branch's BN1 output
      bwd: gradients w.r.t. unit outputs / the pool output, w.r.t. the BN2 output (a2.g), the ReLU6-masked gradient at the BN1
           output (dz1), the shortcut's BN1 output, and -- on the unfused paths (shortcut conv, head conv) -- w.r.t. the conv output."""
    return _Store.apply(x, fwd and BF16_STORE_FWD, bwd and BF16_STORE_BWD) if BF16_STORAGE else x


def conv_pw(x, p, prefix):
    """Conv2D(k=1) (core/architectures.py:130,134,140,170); kernel (1,1,Cin,Cout)."""
    T = x.shape[0]
    if PW_BF16_OPERANDS:
        w2 = p[f'{prefix}.w']
        return _unfold(_PwBf16Operands.apply(_fold(x), w2.reshape(w2.shape[2], w2.shape[3]), p[f'{prefix}.b']), T)
    w = p[f'{prefix}.w'].permute(3, 2, 0, 1)
    return _unfold(F.conv2d(_fold(x), w, p[f'{prefix}.b']), T)


def conv_dw(x, p, prefix, stride):
    """DepthwiseConv2D(3, strides, 'same') (core/architectures.py:132,138); kernel (3,3,C,1)."""
    T = x.shape[0]
    c = x.shape[2]
    w = p[f'{prefix}.w'].permute(2, 3, 0, 1)           # (C,1,3,3)
    ph = same_pad(x.shape[3], 3, stride)
    pw_ = same_pad(x.shape[4], 3, stride)
    xx = F.pad(_fold(x), (pw_[0], pw_[1], ph[0], ph[1]))
    return _unfold(F.conv2d(xx, w, p[f'{prefix}.b'], stride=stride, groups=c), T)


def channel_shuffle(x):
    """core/architectures.py:109-118: reshape (..,C/2,2) -> transpose -> (..,2,C/2);
    out[b*C/2 + a] = in[2a + b].  x: (T,B,C,H,W)."""
    T, B, C, H, W = x.shape
    return x.reshape(T, B, C // 2, 2, H, W).transpose(2, 3).reshape(T, B, C, H, W)


# ------------------------------------------------------------------------------------------------
# tower / trunk
# ------------------------------------------------------------------------------------------------

def shufflenet_unit(x, p, u, training: bool):
    """shufflenet_v2_unit, core/architectures.py:120-145 (u: one entry of oracle.spec.unit_plan).  x: (T,B,C,H,W)."""
    pre = f"img.s{u['stage']}.u{u['unit']}"
    if u['stride'] == 1:
        sc, m = x[:, :, :u['shortcut_c']], x[:, :, u['shortcut_c']:]               # tf.split, :87-97
    else:
        sc, m = _st(x, False, True), x          # (engine: the shortcut's input gradient is stored, the main branch accumulates onto it)
    m = _st(conv_pw(m, p, f'{pre}.pw1'), True, False)
    m = _st(bn_slices(m, p, f'{pre}.bn1', training, True), False, True)
    m = relu6(m)
    m = _st(conv_dw(m, p, f'{pre}.dw', u['stride']), True, False)
    m = _st(bn_slices(m, p, f'{pre}.bn2', training, True), False, True)
    m = _st(conv_pw(m, p, f'{pre}.pw2'), True, False)
    m = relu6(bn_slices(m, p, f'{pre}.bn3', training, True))
    if u['stride'] == 2:
        sc = _st(conv_dw(sc, p, f'{pre}.sc_dw', 2), True, False)
        sc = _st(bn_slices(sc, p, f'{pre}.sc_bn1', training, True), True, True)
        sc = _st(conv_pw(sc, p, f'{pre}.sc_pw'), True, True)
        sc = relu6(bn_slices(sc, p, f'{pre}.sc_bn2', training, True))
    return _st(channel_shuffle(torch.cat([sc, m], dim=2)), True, True)               # :144-145


def shufflenet_v2(image, p, cfg: NetConfig, training: bool, taps: Optional[dict] = None):
    """core/architectures.py:30-173.  image: (B,T,H,W,3) -> (T,B,last_channels)."""
    T = cfg.T
    x = image.permute(1, 0, 4, 2, 3)                   # (T,B,3,H,W)
    w = p['img.stem.conv.w'].permute(3, 2, 0, 1)
    x = _unfold(F.conv2d(_fold(x), w, p['img.stem.conv.b'], stride=2), T)       # valid, :159
    x = _st(x, True, False)
    if taps is not None:
        taps['stem.y'] = x
    x = relu6(bn_slices(x, p, 'img.stem.bn', training, True))                     # :160
    ph = same_pad(x.shape[3], 3, 2)
    pw_ = same_pad(x.shape[4], 3, 2)
    xx = F.pad(_fold(x), (pw_[0], pw_[1], ph[0], ph[1]), value=float('-inf'))
    x = _st(_unfold(max_pool_3x3_s2(xx), T), True, True)                       # :161
    if taps is not None:
        taps['pool'] = x
    for u in unit_plan(cfg):                                                      # :164-167
        x = shufflenet_unit(x, p, u, training)
        if taps is not None:
            taps[f"img.s{u['stage']}.u{u['unit']}"] = x
    x = _st(conv_pw(x, p, 'img.head.conv'), True, True)                          # :170
    x = relu6(bn_slices(x, p, 'img.head.bn', training, True))
    return x.mean(dim=(3, 4))                                                     # GAP :172


def feature_net(v, p, name, training):
    """core/architectures.py:9-27 with DEFAULT_DYNAMICS (units 16, 2 layers, relu6, no input BN).
    v: (B,T,D) -> (T,B,16)."""
    x = v.permute(1, 0, 2)
    for i in range(2):
        x = relu6(x @ p[f'{name}.fc{i}.w'] + p[f'{name}.fc{i}.b'])
        x = bn_slices(x, p, f'{name}.bn{i}', training, False)
    return x


def gru_last(x, p, name):
    """Keras GRU v2 (reset_after=True, tanh/sigmoid, zero init state, last output);
    core/networks.py:47-50, SURVEY.md A.5.  x: (T,B,In) -> (B,u)."""
    K, R, b = p[f'{name}.kernel'], p[f'{name}.recurrent'], p[f'{name}.bias']
    u = R.shape[0]
    h = torch.zeros(x.shape[1], u, dtype=x.dtype)
    for t in range(x.shape[0]):
        xp = x[t] @ K + b[0]
        hp = h @ R + b[1]
        z = torch.sigmoid(xp[:, :u] + hp[:, :u])
        r = torch.sigmoid(xp[:, u:2 * u] + hp[:, u:2 * u])
        hh = torch.tanh(xp[:, 2 * u:] + r * hp[:, 2 * u:])
        h = z * h + (1.0 - z) * hh
    return h


def dynamics_forward(states: Dict[str, torch.Tensor], p, cfg: NetConfig, training: bool, taps=None):
    """dynamics_layers, core/networks.py:37-56 -> (B, 512)."""
    img = shufflenet_v2(states['state_image'], p, cfg, training, taps)
    road = feature_net(states['state_road'], p, 'road', training)
    veh = feature_net(states['state_vehicle'], p, 'vehicle', training)
    nav = feature_net(states['state_navigation'], p, 'navigation', training)
    if taps is not None:
        taps['img_feat'] = img
    hi = gru_last(img, p, 'gru_image')
    hr = gru_last(road, p, 'gru_road')
    hv = gru_last(veh, p, 'gru_vehicle')
    hn = gru_last(nav, p, 'gru_navigation')
    cat = torch.cat([hi, hr, hv, hn], dim=1)                                      # :53-54
    if taps is not None:
        taps['dyn_in'] = cat
    x = bn_slices(cat[None], p, 'dyn.bn', training, False)[0]                     # linear_combination :24-30
    return x @ p['dyn.fc.w'] + p['dyn.fc.b']


def control_branch(d, p, prefix, training):
    """core/networks.py:59-66: [BN -> Dense(320, swish6)] x 2."""
    x = d
    for i in range(2):
        x = bn_slices(x[None], p, f'{prefix}.bn{i}', training, False)[0]
        x = swish6(x @ p[f'{prefix}.fc{i}.w'] + p[f'{prefix}.fc{i}.b'])
    return x


def softplus101(x):
    return F.softplus(x) + 1.01            # utils.softplus(1.0 + 1e-2), rl/utils.py:411-416


def policy_heads(d, p, training):
    """PolicyNetwork.policy_branch, core/networks.py:115-137."""
    h = control_branch(d, p, 'pi', training)
    alpha = softplus101(h @ p['pi.alpha.w'] + p['pi.alpha.b'])
    beta = softplus101(h @ p['pi.beta.w'] + p['pi.beta.b'])
    similarity = torch.tanh(h @ p['pi.similarity.w'] + p['pi.similarity.b'])
    speed = 2.0 * torch.sigmoid(h @ p['pi.speed.w'] + p['pi.speed.b'])
    return alpha, beta, speed, similarity


def value_heads(d, p, training, exp_scale=6.0):
    """CARLANetwork.value_branch / value_head, core/networks.py:255-275."""
    h = control_branch(d, p, 'v', training)
    base = torch.tanh(h @ p['v.base.w'] + p['v.base.b'])
    exp = exp_scale * torch.sigmoid(h @ p['v.exp.w'] + p['v.exp.b'])
    speed = 2.0 * torch.sigmoid(h @ p['v.speed.w'] + p['v.speed.b'])
    similarity = torch.tanh(h @ p['v.similarity.w'] + p['v.similarity.b'])
    return torch.cat([base, exp], dim=1), speed, similarity


# ------------------------------------------------------------------------------------------------
# Beta distribution (TFP Beta(concentration1=alpha, concentration0=beta))
# ------------------------------------------------------------------------------------------------

def beta_log_prob(x, a, b):
    return (a - 1.0) * torch.log(x) + (b - 1.0) * torch.log1p(-x) \
        - (torch.lgamma(a) + torch.lgamma(b) - torch.lgamma(a + b))


def beta_entropy(a, b):
    lnB = torch.lgamma(a) + torch.lgamma(b) - torch.lgamma(a + b)
    return lnB - (a - 1.0) * torch.digamma(a) - (b - 1.0) * torch.digamma(b) \
        + (a + b - 2.0) * torch.digamma(a + b)


class _InjectedSample(torch.autograd.Function):
    """u = Beta sample with pathwise Jacobians du/dalpha, du/dbeta supplied by the harness
    (F8 / SURVEY.md Appendix C-1: the sample and its reparameterisation gradient are
    explicit inputs of the parity contract)."""

    @staticmethod
    def forward(ctx, a, b, u, du_da, du_db):
        ctx.save_for_backward(du_da, du_db)
        return u.clone()

    @staticmethod
    def backward(ctx, g):
        du_da, du_db = ctx.saved_tensors
        return g * du_da, g * du_db, None, None, None


def policy_objective(d, p, batch, hp, training=True):
    """CARLAgent.policy_objective, core/carla_agent.py:394-428 (+ PolicyNetwork.call,
    core/networks.py:96-110).  batch: advantages (B,), old_log_prob (B,A), speed (B,1),
    similarity (B,1), u (B,A), du_da (B,A), du_db (B,A).  In 'stored action' mode the caller
    passes u = stored actions and zero Jacobians (rl/agents/ppo.py:322-325 semantics)."""
    alpha, beta, speed, similarity = policy_heads(d, p, training)
    u = _InjectedSample.apply(alpha, beta, batch['u'], batch['du_da'], batch['du_db'])
    x = torch.clamp(u, EPSILON, 1.0 - EPSILON)                                    # _clip_actions :139-144
    log_prob = beta_log_prob(x, alpha, beta)
    entropy = beta_entropy(alpha, beta).mean()
    ratio = torch.exp(log_prob - batch['old_log_prob']).mean(dim=1)               # :408-409
    adv = batch['advantages']
    c = hp['clip_ratio']
    min_adv = torch.where(adv > 0.0, (1.0 + c) * adv, (1.0 - c) * adv)
    speed_loss = 0.5 * ((batch['speed'] - speed) ** 2).mean(dim=-1).mean()
    sim_loss = 0.5 * ((batch['similarity'] - similarity) ** 2).mean(dim=-1).mean()
    policy_loss = -torch.minimum(ratio * adv, min_adv).mean()
    total = policy_loss - hp['entropy_coef'] * entropy + speed_loss + sim_loss
    aux = dict(alpha=alpha, beta=beta, log_prob=log_prob, entropy=entropy, ratio=ratio, speed=speed,
               similarity=similarity, policy_loss=policy_loss)
    return total, aux


def value_objective(d, p, batch, training=True, exp_scale=6.0):
    """CARLAgent.value_objective, core/carla_agent.py:469-486."""
    values, speed, similarity = value_heads(d, p, training, exp_scale)
    ret = batch['returns']
    base_loss = ((ret[:, 0] - values[:, 0]) ** 2).mean()
    exp_loss = ((ret[:, 1] - values[:, 1]) ** 2).mean()
    value_loss = 0.25 * base_loss + exp_loss / (exp_scale ** 2)
    speed_loss = ((batch['speed'] - speed) ** 2).mean(dim=-1).mean()
    sim_loss = ((batch['similarity'] - similarity) ** 2).mean(dim=-1).mean()
    total = (value_loss + speed_loss + sim_loss) * 0.25
    return total, dict(values=values, speed=speed, similarity=similarity)


# ------------------------------------------------------------------------------------------------
# optimisation
# ------------------------------------------------------------------------------------------------

def clip_by_norm(g: torch.Tensor, c: float) -> torch.Tensor:
    """tf.clip_by_norm per tensor (rl/utils.py:120-121): g * c / max(||g||, c)."""
    l2 = (g * g).sum()
    norm = torch.sqrt(l2) if l2 > 0 else l2
    return g * c / torch.maximum(norm, torch.tensor(c, dtype=g.dtype))


class Adam:
    """Keras Adam (beta1 .9, beta2 .999, eps 1e-7, no amsgrad); one instance per optimizer,
    `t` counts apply_gradients calls (rl/utils.py:29-46).

    TF keeps lr / beta1 / beta2 / epsilon as float32 tensors and evaluates `1 - beta`,
    `beta ** t` and the bias-corrected step size in float32 (Keras `_prepare_local` +
    `ResourceApplyAdam`): 1 - float32(0.999) = 0.000999987..., not 0.001.  The float32 oracle
    reproduces that float32 scalar arithmetic; the float64 oracle evaluates the same formulas
    exactly on the float32-rounded constants."""

    def __init__(self, names: List[str], params: Dict[str, torch.Tensor], beta1=0.9, beta2=0.999, eps=1e-7):
        self.names = names
        self.b1, self.b2, self.eps = np.float32(beta1), np.float32(beta2), np.float32(eps)
        self.t = 0
        self.m = {n: torch.zeros_like(params[n]) for n in names}
        self.v = {n: torch.zeros_like(params[n]) for n in names}
        self.f32 = params[names[0]].dtype == torch.float32

    def step(self, params, grads: Dict[str, torch.Tensor], lr: float):
        self.t += 1
        if self.f32:
            one = np.float32(1.0)
            t = np.float32(self.t)
            alpha = float(np.float32(lr) * np.sqrt(one - np.power(self.b2, t)) / (one - np.power(self.b1, t)))
            omb1, omb2, eps = float(one - self.b1), float(one - self.b2), float(self.eps)
        else:
            b1, b2, lr64 = float(self.b1), float(self.b2), float(np.float32(lr))
            alpha = lr64 * math.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
            omb1, omb2, eps = 1.0 - b1, 1.0 - b2, float(self.eps)
        with torch.no_grad():
            for n in self.names:
                g = grads[n]
                self.m[n].add_((g - self.m[n]) * omb1)
                self.v[n].add_((g * g - self.v[n]) * omb2)
                params[n].sub_(self.m[n] * alpha / (torch.sqrt(self.v[n]) + eps))


class OracleLearner:
    """One reference learner: trunk + policy + old_policy + value, three Adam optimizers.
    step order follows core/carla_agent.py:351-388,430-463 and rl/agents/ppo.py:238-275."""

    def __init__(self, cfg: NetConfig, trunk_np, policy_np, value_np, hp: dict, dtype=torch.float32):
        self.cfg, self.hp, self.dtype = cfg, dict(hp), dtype
        self.tspec, self.pspec, self.vspec = trunk_spec(cfg), policy_spec(cfg), value_spec(cfg)
        self.trunk = to_torch(trunk_np, self.tspec, dtype)
        self.policy = to_torch(policy_np, self.pspec, dtype)
        self.value = to_torch(value_np, self.vspec, dtype)
        self.old_policy = {k: v.detach().clone() for k, v in self.policy.items()}   # update_old_policy in ctor
        tn = lambda spec: [n for n, _, _, tr in spec if tr]
        self.opt_trunk = Adam(tn(self.tspec), self.trunk)
        self.opt_policy = Adam(tn(self.pspec), self.policy)
        self.opt_value = Adam(tn(self.vspec), self.value)
        self.last_grads = {}

    def _cast(self, batch):
        return {k: (torch.as_tensor(v).to(self.dtype) if not isinstance(v, dict) else
                    {kk: torch.as_tensor(vv).to(self.dtype) for kk, vv in v.items()}) for k, v in batch.items()}

    def _grads(self, loss, params, spec):
        names = [n for n, _, _, tr in spec if tr]
        gs = torch.autograd.grad(loss, [params[n] for n in names], retain_graph=True, allow_unused=True)
        return {n: (g if g is not None else torch.zeros_like(params[n])) for n, g in zip(names, gs)}

    def policy_grads(self, batch):
        batch = self._cast(batch)
        d = dynamics_forward(batch['states'], self.trunk, self.cfg, training=True)
        loss, aux = policy_objective(d, self.policy, batch, self.hp, training=True)
        gp = self._grads(loss, self.policy, self.pspec)
        gt = self._grads(loss, self.trunk, self.tspec)
        aux['dynamics'] = d
        return loss, gp, gt, aux

    def value_grads(self, batch):
        batch = self._cast(batch)
        d = dynamics_forward(batch['states'], self.trunk, self.cfg, training=True)
        loss, aux = value_objective(d, self.value, batch, training=True)
        gv = self._grads(loss, self.value, self.vspec)
        gt = self._grads(loss, self.trunk, self.tspec)
        aux['dynamics'] = d
        return loss, gv, gt, aux

    def policy_forward(self, batch):
        """Forward quantities of policy_grads only (loss, aux: alpha / beta / log_prob ...), no autograd graph."""
        with torch.no_grad():
            batch = self._cast(batch)
            d = dynamics_forward(batch['states'], self.trunk, self.cfg, training=True)
            return policy_objective(d, self.policy, batch, self.hp, training=True)

    def value_forward(self, batch):
        with torch.no_grad():
            batch = self._cast(batch)
            d = dynamics_forward(batch['states'], self.trunk, self.cfg, training=True)
            return value_objective(d, self.value, batch, training=True)

    def policy_step(self, batch, grads=None):
        """`grads` lets the N-shard data-parallel emulation inject averaged gradients."""
        if grads is None:
            loss, gp, gt, aux = self.policy_grads(batch)
        else:
            loss, gp, gt, aux = grads
        self.last_grads = dict(policy=gp, trunk=gt)
        self.opt_trunk.step(self.trunk, gt, self.hp['dynamics_lr'])                # unclipped (F9)
        c = self.hp.get('clip_norm_policy', 1.0)
        if c is not None:
            gp = {n: clip_by_norm(g, c) for n, g in gp.items()}
        self.old_policy = {k: v.detach().clone() for k, v in self.policy.items()}  # rl/agents/ppo.py:249
        self.opt_policy.step(self.policy, gp, self.hp['policy_lr'])
        return loss, aux

    def value_step(self, batch, grads=None):
        if grads is None:
            loss, gv, gt, aux = self.value_grads(batch)
        else:
            loss, gv, gt, aux = grads
        self.last_grads = dict(value=gv, trunk=gt)
        self.opt_trunk.step(self.trunk, gt, self.hp['dynamics_lr'])
        c = self.hp.get('clip_norm_value', 1.0)
        if c is not None:
            gv = {n: clip_by_norm(g, c) for n, g in gv.items()}
        self.opt_value.step(self.value, gv, self.hp['value_lr'])
        return loss, aux

    def predict(self, states):
        """CARLANetwork.predict deterministic part (core/networks.py:181-193): old_policy +
        value heads on inference-mode trunk.  Returns alpha, beta, value(base,exp)."""
        with torch.no_grad():
            st = {k: torch.as_tensor(v).to(self.dtype) for k, v in states.items()}
            d = dynamics_forward(st, self.trunk, self.cfg, training=False)
            alpha, beta, speed, sim = policy_heads(d, self.old_policy, False)
            value, _, _ = value_heads(d, self.value, False)
        return alpha, beta, value, d
