"""LearnerEngine: thin Python owner of a `cdrl_learner` handle.

PyTorch is used for device memory (parameter arenas, Adam state, workspace), streams and
`torch.distributed` only; every arithmetic step of the learner runs in libcdrl_hip.so.
"""
import ctypes as C
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import TRUNK, POLICY, VALUE, OLD_POLICY

MODEL_IDS = dict(trunk=TRUNK, policy=POLICY, value=VALUE, old_policy=OLD_POLICY)


def make_config(B, T=4, H=90, W=120, road=9, vehicle=4, navigation=5, A=2, **kw) -> _lib.Config:
    lib = _lib.load()
    cfg = _lib.Config()
    lib.cdrl_config_default(C.byref(cfg))
    cfg.B, cfg.T, cfg.H, cfg.W = B, T, H, W
    cfg.road, cfg.vehicle, cfg.navigation, cfg.A = road, vehicle, navigation, A
    for k, v in kw.items():
        if k == 'compute' and isinstance(v, str):
            # 'f32' | 'bf16' (bf16 MFMA operands in the tower's 1x1 convs, float32 tensors) | 'bf16s' (+ bf16 activation STORAGE in
            # the tower: configuration 3 in full)
            v = {'f32': _lib.COMPUTE_F32, 'bf16': _lib.COMPUTE_BF16_OPERANDS, 'bf16s': _lib.COMPUTE_BF16_STORAGE}[v]
        if k in ('stage_c', 'stage_n'):
            for i in range(3):
                getattr(cfg, k)[i] = v[i]
        else:
            setattr(cfg, k, v)
    return cfg


class ParamTable:
    """Variable inventory of one model (trunk / policy / value) as reported by the engine."""

    def __init__(self, lib, handle, model):
        self.entries = []
        n = lib.cdrl_learner_param_count(handle, model)
        for i in range(n):
            pi = _lib.ParamInfo()
            _lib.check(lib.cdrl_learner_param_info(handle, model, i, C.byref(pi)), 'param_info')
            self.entries.append(dict(name=pi.name.decode(), shape=tuple(pi.shape[:pi.ndim]), numel=int(pi.numel),
                                     trainable=bool(pi.trainable), offset=int(pi.offset)))
        self.by_name = {e['name']: e for e in self.entries}

    def spec(self):
        return [(e['name'], e['shape'], e['trainable']) for e in self.entries]


class LearnerEngine:
    def __init__(self, B, device: Optional[str] = 'cuda:0', share_with: 'LearnerEngine' = None, **cfg):
        """device=None -> host-only inspection (parameter tables, workspace size; no HIP calls).
        share_with -> reuse another engine's parameter arenas (e.g. a B=1 rollout engine)."""
        self.lib = _lib.load()
        self.cfg = make_config(B, **cfg)
        h = C.c_void_p()
        _lib.check(self.lib.cdrl_learner_create(C.byref(self.cfg), C.byref(h)), 'cdrl_learner_create')
        self.h = h
        self.tables = {m: ParamTable(self.lib, h, mid) for m, mid in (('trunk', TRUNK), ('policy', POLICY), ('value', VALUE))}
        self.params_total = int(self.lib.cdrl_learner_params_total(h))
        self.grads_total = int(self.lib.cdrl_learner_grads_total(h))
        self.workspace_bytes = int(self.lib.cdrl_learner_workspace_bytes(h))
        self.device = device
        self.hp = dict(policy_lr=3e-4, value_lr=3e-4, dynamics_lr=3e-4, clip_ratio=0.2, entropy_coef=1.0,
                       clip_norm_policy=1.0, clip_norm_value=1.0, beta1=0.9, beta2=0.999, eps=1e-7)
        self._keep = []
        self._keep_stage = {}
        self._pred_out = None
        if device is None:
            return
        dev = torch.device(device)
        if dev.type == 'cuda' and dev.index is not None and dev.index != torch.cuda.current_device():
            # one process per GPU: the engine creates its HIP streams on the CURRENT device and every entry point runs there
            raise ValueError(f'LearnerEngine(device={device!r}): call torch.cuda.set_device({dev.index}) first '
                             f'(current device is {torch.cuda.current_device()})')
        if share_with is not None:
            self.params, self.grads = share_with.params, share_with.grads
            self.adam_m, self.adam_v = share_with.adam_m, share_with.adam_v
        else:
            self.params = torch.zeros(self.params_total, dtype=torch.float32, device=dev)
            self.grads = torch.zeros(self.grads_total, dtype=torch.float32, device=dev)
            self.adam_m = torch.zeros(self.grads_total, dtype=torch.float32, device=dev)
            self.adam_v = torch.zeros(self.grads_total, dtype=torch.float32, device=dev)
        self.workspace = torch.zeros(self.workspace_bytes, dtype=torch.uint8, device=dev)
        _lib.check(self.lib.cdrl_learner_bind(h, _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.adam_m),
                                              _lib.ptr(self.adam_v), _lib.ptr(self.workspace), self.workspace_bytes),
                   'cdrl_learner_bind')
        if share_with is not None and share_with.device is not None:
            # one optimizer: learning rates, clip and the Adam step counters live in the owner's device block
            _lib.check(self.lib.cdrl_learner_share_hparams(h, share_with.h), 'cdrl_learner_share_hparams')
            self.hp = share_with.hp
            self._hp_owner = share_with
        else:
            self._hp_owner = None
            self.set_hparams()

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                self.lib.cdrl_learner_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def region(self, model: str, trainable: bool):
        mid = MODEL_IDS[model]
        off = int(self.lib.cdrl_learner_region_offset(self.h, mid, 1 if trainable else 0))
        n = int(self.lib.cdrl_learner_region_elems(self.h, mid, 1 if trainable else 0))
        return off, n

    def _view(self, flat, model, e, grad=False):
        off, _ = self.region(model, e['trainable'])
        if grad and model == 'old_policy':
            raise KeyError('old_policy has no gradients')
        return flat[off + e['offset']: off + e['offset'] + e['numel']].view(e['shape'])

    def param_views(self, model: str) -> Dict[str, torch.Tensor]:
        table = self.tables['policy' if model == 'old_policy' else model]
        return {e['name']: self._view(self.params, model, e) for e in table.entries}

    def grad_views(self, model: str) -> Dict[str, torch.Tensor]:
        return {e['name']: self._view(self.grads, model, e, True) for e in self.tables[model].entries if e['trainable']}

    def adam_views(self, model: str):
        t = self.tables[model]
        return ({e['name']: self._view(self.adam_m, model, e) for e in t.entries if e['trainable']},
                {e['name']: self._view(self.adam_v, model, e) for e in t.entries if e['trainable']})

    def load_params(self, model: str, values: Dict[str, np.ndarray]):
        views = self.param_views(model)
        for name, v in views.items():
            v.copy_(torch.as_tensor(np.asarray(values[name], dtype=np.float32)).reshape(v.shape))

    def export_params(self, model: str) -> Dict[str, np.ndarray]:
        return {k: v.detach().cpu().numpy().copy() for k, v in self.param_views(model).items()}

    def set_hparams(self, **kw):
        if getattr(self, '_hp_owner', None) is not None:
            return self._hp_owner.set_hparams(**kw)
        self.hp.update(kw)
        hp = _lib.HParams()
        for k in ('policy_lr', 'value_lr', 'dynamics_lr', 'clip_ratio', 'entropy_coef', 'beta1', 'beta2', 'eps'):
            setattr(hp, k, float(self.hp[k]))
        hp.clip_norm_policy = float(self.hp['clip_norm_policy'] or 0.0)
        hp.clip_norm_value = float(self.hp['clip_norm_value'] or 0.0)
        _lib.check(self.lib.cdrl_learner_set_hparams(self.h, C.byref(hp), self._stream()), 'set_hparams')

    def set_comm_stream(self, stream: Optional['torch.cuda.Stream']):
        """See cdrl_learner_set_comm_stream (data-parallel overlap); None switches it off."""
        self._comm_stream = stream      # keep it alive: the engine holds the raw handle
        _lib.check(self.lib.cdrl_learner_set_comm_stream(self.h, C.c_void_p(stream.cuda_stream) if stream is not None else None),
                   'set_comm_stream')

    def tail_offset(self) -> int:
        """See cdrl_learner_tail_offset: first trunk-gradient element that is final when the communication stream is released."""
        off = int(self.lib.cdrl_learner_tail_offset(self.h))
        if off < 0:
            raise _lib.CdrlError('cdrl_learner_tail_offset failed')
        return off

    def reset_optimizer(self):
        self.adam_m.zero_()
        self.adam_v.zero_()
        _lib.check(self.lib.cdrl_learner_reset_optimizer_steps(self.h, self._stream()), 'reset_optimizer_steps')

    def buffer(self, which: int, shape=None) -> torch.Tensor:
        """Zero-copy torch view of an engine-owned workspace buffer."""
        p = C.c_void_p()
        n = C.c_int64()
        _lib.check(self.lib.cdrl_learner_get_buffer(self.h, which, C.byref(p), C.byref(n)), 'get_buffer')
        base = self.workspace.data_ptr()
        off = p.value - base
        t = self.workspace[off: off + 4 * n.value].view(torch.float32)
        return t.view(shape) if shape is not None else t

    def check_guards(self):
        """(bands that lost their pattern, workspace byte offset of the first one or -1): the out-of-bounds canaries behind every
        workspace tensor of a learner created with CDRL_GUARD=1 in the environment (cdrl_learner_check_guards)."""
        bad, first = C.c_int64(0), C.c_int64(-1)
        _lib.check(self.lib.cdrl_learner_check_guards(self.h, self._stream(), C.byref(bad), C.byref(first)), 'check_guards')
        return int(bad.value), int(first.value)

    def named_buffer(self, name: str, dtype=torch.float32, shape=None) -> torch.Tensor:
        """Zero-copy view of a named internal tensor (cdrl_learner_named_buffer; parity tests)."""
        p = C.c_void_p()
        n = C.c_int64()
        _lib.check(self.lib.cdrl_learner_named_buffer(self.h, name.encode(), C.byref(p), C.byref(n)), f'named_buffer({name})')
        off = p.value - self.workspace.data_ptr()
        t = self.workspace[off: off + n.value].view(dtype)
        return t.view(shape) if shape is not None else t

    # ------------------------------------------------------------------ staging
    def stage(self, batch: dict, slot: str) -> dict:
        """Copies a (nested) batch into persistent device buffers owned by the engine and returns those.
        The step entry points are replayed from captured hipGraphs keyed on their pointer arguments, so
        minibatches that live in fresh tensors every time (tf.data-style gathers) go through fixed
        staging buffers (one D2D copy, ~0.05 ms for a 256x4x90x120x3 minibatch)."""
        store = self._keep_stage.setdefault(slot, {})
        # hipGraph replay is opt-in (CDRL_GRAPH=1; eager launches are faster on this stack): without it the
        # entry points take any contiguous device tensor, and the staging copy (a 133 MB D2D copy per B = 256 pass) is skipped
        direct = os.environ.get('CDRL_GRAPH', '0') in ('', '0')

        def put(key, t):
            if t is None:
                return None
            if direct and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous():
                return t
            buf = store.get(key)
            if buf is None or buf.shape != t.shape:
                buf = torch.empty_like(t, memory_format=torch.contiguous_format)
                store[key] = buf
            buf.copy_(t)
            return buf
        out = {}
        for k, v in batch.items():
            out[k] = {kk: put(f'{k}/{kk}', vv) for kk, vv in v.items()} if isinstance(v, dict) else put(k, v)
        return out

    # ------------------------------------------------------------------ steps
    def _states(self, batch):
        st = batch['states'] if 'states' in batch else batch
        return st['state_image'], st['state_road'], st['state_vehicle'], st['state_navigation']

    @staticmethod
    def _chk(t, shape, name):
        if t is None:
            return
        if tuple(t.shape) != tuple(shape) or t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise ValueError(f'{name}: expected contiguous float32 cuda tensor of shape {tuple(shape)}, '
                             f'got {tuple(t.shape)} {t.dtype} contiguous={t.is_contiguous()} cuda={t.is_cuda}')

    def _check_states(self, img, road, veh, nav):
        c = self.cfg
        self._chk(img, (c.B, c.T, c.H, c.W, 3), 'state_image')
        self._chk(road, (c.B, c.T, c.road), 'state_road')
        self._chk(veh, (c.B, c.T, c.vehicle), 'state_vehicle')
        self._chk(nav, (c.B, c.T, c.navigation), 'state_navigation')

    def policy_forward_backward(self, batch, grad_scale=1.0):
        c = self.cfg
        img, road, veh, nav = self._states(batch)
        self._check_states(img, road, veh, nav)
        self._chk(batch['advantages'], (c.B,), 'advantages')
        self._chk(batch['old_log_prob'], (c.B, c.A), 'old_log_prob')
        self._chk(batch['u'], (c.B, c.A), 'u')
        for k in ('speed', 'similarity'):
            if batch[k].numel() != c.B:
                raise ValueError(f'{k}: expected {c.B} elements')
        pb = _lib.PolicyBatch(image=img.data_ptr(), road=road.data_ptr(), vehicle=veh.data_ptr(),
                              navigation=nav.data_ptr(), advantages=batch['advantages'].data_ptr(),
                              old_log_prob=batch['old_log_prob'].data_ptr(), speed=batch['speed'].data_ptr(),
                              similarity=batch['similarity'].data_ptr(), u=batch['u'].data_ptr(),
                              du_dalpha=batch['du_da'].data_ptr() if batch.get('du_da') is not None else None,
                              du_dbeta=batch['du_db'].data_ptr() if batch.get('du_db') is not None else None)
        _lib.check(self.lib.cdrl_learner_policy_forward_backward(self.h, C.byref(pb), float(grad_scale), self._stream()),
                   'policy_forward_backward')

    def _policy_batch(self, batch, with_states=True):
        c = self.cfg
        img, road, veh, nav = self._states(batch)
        if with_states:
            self._check_states(img, road, veh, nav)
        self._chk(batch['advantages'], (c.B,), 'advantages')
        self._chk(batch['old_log_prob'], (c.B, c.A), 'old_log_prob')
        self._chk(batch['u'], (c.B, c.A), 'u')
        for k in ('du_da', 'du_db'):
            if batch.get(k) is not None:
                self._chk(batch[k], (c.B, c.A), k)
        return _lib.PolicyBatch(image=img.data_ptr(), road=road.data_ptr(), vehicle=veh.data_ptr(),
                                navigation=nav.data_ptr(), advantages=batch['advantages'].data_ptr(),
                                old_log_prob=batch['old_log_prob'].data_ptr(), speed=batch['speed'].data_ptr(),
                                similarity=batch['similarity'].data_ptr(), u=batch['u'].data_ptr(),
                                du_dalpha=batch['du_da'].data_ptr() if batch.get('du_da') is not None else None,
                                du_dbeta=batch['du_db'].data_ptr() if batch.get('du_db') is not None else None)

    def policy_forward(self, states):
        """train-mode forward of trunk + policy head; returns (alpha, beta) views (B, A) of the current policy."""
        img, road, veh, nav = self._states(states)
        self._check_states(img, road, veh, nav)
        _lib.check(self.lib.cdrl_learner_policy_forward(self.h, _lib.ptr(img), _lib.ptr(road), _lib.ptr(veh),
                                                        _lib.ptr(nav), self._stream()), 'policy_forward')
        aux = self.buffer(_lib.BUF_AUX_P, (self.cfg.B, 4, self.cfg.A))
        return aux[:, 0], aux[:, 1]

    def policy_backward(self, batch, grad_scale=1.0):
        pb = self._policy_batch(batch)
        _lib.check(self.lib.cdrl_learner_policy_backward(self.h, C.byref(pb), float(grad_scale), self._stream()),
                   'policy_backward')

    def policy_forward_backward_resample(self, batch, seed: int, offset: int, grad_scale=1.0):
        """F8-faithful step with the Beta re-sampling done on the device (no torch.distributions)."""
        b = dict(batch)
        b.setdefault('u', batch['old_log_prob'])       # placeholder; ignored by the entry point
        pb = self._policy_batch(b)
        _lib.check(self.lib.cdrl_learner_policy_forward_backward_resample(self.h, C.byref(pb), int(seed), int(offset),
                                                                         float(grad_scale), self._stream()),
                   'policy_forward_backward_resample')

    def policy_apply(self):
        _lib.check(self.lib.cdrl_learner_policy_apply(self.h, self._stream()), 'policy_apply')

    def value_forward_backward(self, batch, grad_scale=1.0):
        c = self.cfg
        img, road, veh, nav = self._states(batch)
        self._check_states(img, road, veh, nav)
        self._chk(batch['returns'], (c.B, 2), 'returns')
        vb = _lib.ValueBatch(image=img.data_ptr(), road=road.data_ptr(), vehicle=veh.data_ptr(),
                             navigation=nav.data_ptr(), returns=batch['returns'].data_ptr(),
                             speed=batch['speed'].data_ptr(), similarity=batch['similarity'].data_ptr())
        _lib.check(self.lib.cdrl_learner_value_forward_backward(self.h, C.byref(vb), float(grad_scale), self._stream()),
                   'value_forward_backward')

    def value_apply(self):
        _lib.check(self.lib.cdrl_learner_value_apply(self.h, self._stream()), 'value_apply')

    def sequence(self):
        """Context manager around a run of learner calls on the current stream with nothing of the caller's own between them (one
        minibatch on one GPU: policy pass, apply, value pass, apply): the hand-overs between the caller's stream and the engine's
        happen once, around the whole run (`cdrl_learner_sequence_begin / _end`).  Not for runs with collectives in between."""
        eng = self

        class _Seq:
            def __enter__(self_inner):          # (re-entrant on the Python side: only the outermost level brackets)
                depth = getattr(eng, '_seq_depth', 0)
                if depth == 0:
                    _lib.check(eng.lib.cdrl_learner_sequence_begin(eng.h, eng._stream()), 'sequence_begin')
                eng._seq_depth = depth + 1
                return eng

            def __exit__(self_inner, *exc):
                eng._seq_depth -= 1
                if eng._seq_depth == 0:
                    _lib.check(eng.lib.cdrl_learner_sequence_end(eng.h, eng._stream()), 'sequence_end')
                return False

        return _Seq()

    def policy_step(self, batch):
        with self.sequence():
            self.policy_forward_backward(batch)
            self.policy_apply()

    def value_step(self, batch):
        with self.sequence():
            self.value_forward_backward(batch)
            self.value_apply()

    def update_old_policy(self):
        _lib.check(self.lib.cdrl_learner_update_old_policy(self.h, self._stream()), 'update_old_policy')

    def trunk_forward_train(self, states):
        img, road, veh, nav = self._states(states)
        self._check_states(img, road, veh, nav)
        _lib.check(self.lib.cdrl_learner_trunk_forward_train(self.h, _lib.ptr(img), _lib.ptr(road), _lib.ptr(veh),
                                                             _lib.ptr(nav), self._stream()), 'trunk_forward_train')
        return self.buffer(_lib.BUF_DYNAMICS, (self.cfg.B, self.cfg.dyn))

    def predict(self, states):
        c = self.cfg
        st = states['states'] if 'states' in states else states
        staged = self.stage({k: st[k] for k in ('state_image', 'state_road', 'state_vehicle', 'state_navigation')}, 'predict')
        img, road, veh, nav = self._states(staged)
        self._check_states(img, road, veh, nav)
        if self._pred_out is None:       # persistent outputs: the captured graph writes to fixed addresses
            self._pred_out = (torch.empty((c.B, 4, c.A), dtype=torch.float32, device=img.device),
                              torch.empty((c.B, 4), dtype=torch.float32, device=img.device),
                              torch.empty((c.B, c.dyn), dtype=torch.float32, device=img.device))
        dist, value, dyn = self._pred_out
        _lib.check(self.lib.cdrl_learner_predict(self.h, _lib.ptr(img), _lib.ptr(road), _lib.ptr(veh), _lib.ptr(nav),
                                                 _lib.ptr(dist), _lib.ptr(value), _lib.ptr(dyn), self._stream()), 'predict')
        return dict(alpha=dist[:, 0], beta=dist[:, 1], mean=dist[:, 2], std=dist[:, 3], value=value[:, :2],
                    speed=value[:, 2], similarity=value[:, 3], dynamics=dyn)

    def metrics(self, which='policy'):
        m = self.buffer(_lib.BUF_METRICS_P if which == 'policy' else _lib.BUF_METRICS_V).cpu().numpy()
        if which == 'policy':
            keys = ('loss', 'policy_loss', 'entropy', 'speed_loss', 'similarity_loss', 'ratio', 'log_prob')
        else:
            keys = ('loss', 'value_loss', 'speed_loss', 'similarity_loss')
        return {k: float(m[i]) for i, k in enumerate(keys)}


def gae_returns(rewards: torch.Tensor, values_be: torch.Tensor, gamma: float, lambda_: float, scale: float = 2.0):
    """Device GAE / returns for one env shard (cdrl_gae_returns).  rewards (N+1,), values_be (N+1,2)."""
    lib = _lib.load()
    n = rewards.numel() - 1
    dev = rewards.device
    returns = torch.empty(n, dtype=torch.float32, device=dev)
    returns_be = torch.empty((n, 2), dtype=torch.float32, device=dev)
    adv_raw = torch.empty(n, dtype=torch.float32, device=dev)
    adv = torch.empty(n, dtype=torch.float32, device=dev)
    scratch = torch.empty(2 * (n + 1) + 2, dtype=torch.float64, device=dev)
    _lib.check(lib.cdrl_gae_returns(_lib.ptr(rewards), _lib.ptr(values_be), n, float(gamma), float(lambda_), float(scale),
                                    _lib.ptr(returns), _lib.ptr(returns_be), _lib.ptr(adv_raw), _lib.ptr(adv),
                                    _lib.ptr(scratch), C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'gae_returns')
    return returns, returns_be, adv_raw, adv
