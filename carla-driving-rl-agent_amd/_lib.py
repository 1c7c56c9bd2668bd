"""ctypes binding of libcdrl_hip.so (C ABI in include/cdrl.h).

The product path has NO CPU fallback: if the HIP library is missing the import fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libcdrl_hip.so')


class CdrlError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [('B', C.c_int32), ('T', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('road', C.c_int32), ('vehicle', C.c_int32), ('navigation', C.c_int32), ('A', C.c_int32),
                ('stem', C.c_int32), ('stage_c', C.c_int32 * 3), ('stage_n', C.c_int32 * 3), ('last', C.c_int32),
                ('feat', C.c_int32), ('rnn_image', C.c_int32), ('rnn_small', C.c_int32), ('dyn', C.c_int32),
                ('head', C.c_int32), ('exp_scale', C.c_float), ('compute', C.c_int32)]


COMPUTE_F32, COMPUTE_BF16_OPERANDS, COMPUTE_BF16_STORAGE = 0, 1, 2


class ParamInfo(C.Structure):
    _fields_ = [('name', C.c_char * 64), ('shape', C.c_int32 * 4), ('ndim', C.c_int32), ('trainable', C.c_int32),
                ('numel', C.c_int64), ('offset', C.c_int64)]


class HParams(C.Structure):
    _fields_ = [('policy_lr', C.c_float), ('value_lr', C.c_float), ('dynamics_lr', C.c_float),
                ('clip_ratio', C.c_float), ('entropy_coef', C.c_float), ('clip_norm_policy', C.c_float),
                ('clip_norm_value', C.c_float), ('beta1', C.c_float), ('beta2', C.c_float), ('eps', C.c_float)]


_fp = C.c_void_p   # device pointers travel as plain addresses


class PolicyBatch(C.Structure):
    _fields_ = [(n, _fp) for n in ('image', 'road', 'vehicle', 'navigation', 'advantages', 'old_log_prob', 'speed',
                                   'similarity', 'u', 'du_dalpha', 'du_dbeta')]


class ValueBatch(C.Structure):
    _fields_ = [(n, _fp) for n in ('image', 'road', 'vehicle', 'navigation', 'returns', 'speed', 'similarity')]


TRUNK, POLICY, VALUE, OLD_POLICY = 0, 1, 2, 3
BUF_DYNAMICS, BUF_IMG_FEAT, BUF_METRICS_P, BUF_METRICS_V, BUF_AUX_P, BUF_AUX_V, BUF_LIN_P, BUF_LIN_V, BUF_SAMPLE = range(9)

_i, _i64, _f, _d, _sz = C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t
_L = C.c_void_p

# name -> (restype, argtypes); must list every symbol declared in include/cdrl.h
PROTOTYPES = {
    'cdrl_last_error': (C.c_char_p, []),
    'cdrl_version': (_i, []),
    'cdrl_env_overrides': (_i, [C.c_char_p, _i]),
    'cdrl_diag_active': (_i, []),
    'cdrl_crc32c': (C.c_uint32, [C.c_uint32, C.c_void_p, C.c_size_t]),
    'cdrl_config_default': (None, [C.POINTER(Config)]),
    'cdrl_learner_create': (_i, [C.POINTER(Config), C.POINTER(_L)]),
    'cdrl_learner_destroy': (None, [_L]),
    'cdrl_learner_param_count': (_i, [_L, _i]),
    'cdrl_learner_param_info': (_i, [_L, _i, _i, C.POINTER(ParamInfo)]),
    'cdrl_learner_region_offset': (_i64, [_L, _i, _i]),
    'cdrl_learner_region_elems': (_i64, [_L, _i, _i]),
    'cdrl_learner_params_total': (_i64, [_L]),
    'cdrl_learner_grads_total': (_i64, [_L]),
    'cdrl_learner_workspace_bytes': (_sz, [_L]),
    'cdrl_learner_bind': (_i, [_L, _fp, _fp, _fp, _fp, _fp, _sz]),
    'cdrl_learner_set_hparams': (_i, [_L, C.POINTER(HParams), _fp]),
    'cdrl_learner_share_hparams': (_i, [_L, _L]),
    'cdrl_learner_set_comm_stream': (_i, [_L, _fp]),
    'cdrl_learner_tail_offset': (_i64, [_L]),
    'cdrl_learner_reset_optimizer_steps': (_i, [_L, _fp]),
    'cdrl_learner_policy_forward_backward': (_i, [_L, C.POINTER(PolicyBatch), _f, _fp]),
    'cdrl_learner_policy_forward': (_i, [_L, _fp, _fp, _fp, _fp, _fp]),
    'cdrl_learner_policy_backward': (_i, [_L, C.POINTER(PolicyBatch), _f, _fp]),
    'cdrl_learner_policy_forward_backward_resample': (_i, [_L, C.POINTER(PolicyBatch), C.c_uint64, C.c_uint64, _f, _fp]),
    'cdrl_pwconv_x3_packed_bytes': (_i64, [_i]),
    'cdrl_pwconv_x3_packed_bytes_n': (_i64, [_i, _i]),
    'cdrl_pwconv_x3_partial_rows': (_i, [_i, _i, _i, _i]),
    'cdrl_pwconv_x3_pack': (_i, [_fp, _i, _i, _i, _i, _fp, _fp]),
    'cdrl_pwconv_x3': (_i, [_fp, _i, _i, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp, _fp]),
    'cdrl_pwconv_x3_wide_bwd_rows': (_i, [_i]),
    'cdrl_pwconv_x3_wide_bwd': (_i, [_fp, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _fp]),
    'cdrl_pwconv_bwd_fused_workspace': (_i64, [_i, _i, _i, _i, _i]),
    'cdrl_pwconv_bwd_fused': (_i, [_fp, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _i, _i, _fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_void_p, _fp, _i,
                                   _i, _i, _fp, _fp, _fp, C.c_void_p, _i, _i, _i, _i, C.c_void_p]),
    'cdrl_gemm_x3_packed_bytes': (_i64, [_i, _i]),
    'cdrl_gemm_x3_pack': (_i, [_fp, _i, _i, _i, _i, _fp, _fp]),
    'cdrl_gemm_x3': (_i, [_fp, _i, _i, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp]),
    'cdrl_f32_to_bf16': (_i, [_fp, _fp, _i64, _fp]),
    'cdrl_bf16_to_f32': (_i, [_fp, _fp, _i64, _fp]),
    'cdrl_pwconv_bf16_partial_rows': (_i, [_i, _i, _i, _i]),
    'cdrl_pwconv_bf16_packed_elems': (_i64, [_i]),
    'cdrl_pwconv_bf16_pack': (_i, [_fp, _i, _i, _fp, _fp]),
    'cdrl_pwconv_bf16': (_i, [_fp, _i, _i, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp, _fp]),
    'cdrl_gru_step_fwd': (_i, [_fp] * 9 + [_i, _i, _fp]),
    'cdrl_gru_step_bwd': (_i, [_fp, _i] + [_fp] * 9 + [_i, _i, _fp]),
    'cdrl_beta_sample_logp': (_i, [_fp, _fp, _i, _i, _i, C.c_uint64, C.c_uint64, _fp, _fp, _fp]),
    'cdrl_beta_sample': (_i, [_fp, _fp, _i, _i, _i, C.c_uint64, C.c_uint64, _fp, _fp, _fp, _fp]),
    'cdrl_gamma_implicit_grad': (_i, [_fp, _fp, _i, _fp, _fp]),
    'cdrl_beta_sample_gammas': (_i, [_fp, _fp, _i, _i, _i, C.c_uint64, C.c_uint64, _fp, _fp]),
    'cdrl_philox_words': (_i, [C.c_uint64, C.c_uint64, C.c_uint64, _i, _i, _fp, _fp]),
    'cdrl_learner_policy_apply': (_i, [_L, _fp]),
    'cdrl_learner_sequence_begin': (_i, [_L, _fp]),
    'cdrl_learner_sequence_end': (_i, [_L, _fp]),
    'cdrl_learner_value_forward_backward': (_i, [_L, C.POINTER(ValueBatch), _f, _fp]),
    'cdrl_learner_value_apply': (_i, [_L, _fp]),
    'cdrl_learner_update_old_policy': (_i, [_L, _fp]),
    'cdrl_learner_predict': (_i, [_L, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    'cdrl_learner_trunk_forward_train': (_i, [_L, _fp, _fp, _fp, _fp, _fp]),
    'cdrl_learner_get_buffer': (_i, [_L, _i, C.POINTER(_fp), C.POINTER(_i64)]),
    'cdrl_learner_named_buffer': (_i, [_L, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(_i64)]),
    'cdrl_learner_check_guards': (_i, [_L, _fp, C.POINTER(_i64), C.POINTER(_i64)]),
    'cdrl_gae_returns': (_i, [_fp, _fp, _i, _d, _d, _f, _fp, _fp, _fp, _fp, _fp, _fp]),
    'cdrl_gather_rows': (_i, [_fp, _fp, _fp, _i, _i64, _fp]),
    'cdrl_gemm_nn': (_i, [_fp, _i, _i, _fp, _i, _i, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp]),
    'cdrl_gemm_tn_workspace_elems': (_i64, [_i, _i, _i]),
    'cdrl_gemm_tn': (_i, [_fp, _i, _i, _fp, _i, _i, _fp, _i, _i, _i, _fp, _i, _fp]),
    'cdrl_bn_small_fwd': (_i, [_fp, _i, _i, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    'cdrl_bn_small_bwd': (_i, [_fp, _fp, _i, _i, _fp, _fp, _fp, _fp, _fp, _fp]),
    'cdrl_linear_heads_fwd': (_i, [_fp, _i, _fp, _fp, _fp, _fp, _i, _i, _fp]),
    'cdrl_linear_heads_bwd': (_i, [_fp, _i, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _fp]),
    'cdrl_stem_fwd': (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp]),
    'cdrl_stem_fwd_stats_rows': (_i, [_i] * 5),
    'cdrl_stem_fwd_stats': (_i, [_fp] * 5 + [_i] * 5 + [_fp]),
    'cdrl_stem_bwd_workspace_doubles': (_i64, [_i, _i, _i, _i, _i]),
    'cdrl_stem_bwd_filter': (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _fp]),
    'cdrl_dwconv_fwd': (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp]),
    'cdrl_dwconv_bwd_data': (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _fp]),
    'cdrl_dwconv_bwd_workspace_doubles': (_i64, [_i, _i, _i, _i, _i]),
    'cdrl_dwconv_bwd_filter': (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _fp]),
    'cdrl_augment_workspace_floats': (_i64, [_i, _i, _i]),
    'cdrl_augment_images': (_i, [_fp, _fp, _i, _i, _i, _fp, _fp, _fp]),
    'cdrl_stem_block_bwd_workspace_doubles': (_i64, [_i] * 5),
    'cdrl_stem_block_bwd': (_i, [_fp] * 5 + [_i] * 5 + [_fp] * 7),
    'cdrl_stem_block_bwd_pooled': (_i, [_fp] * 6 + [_i] * 5 + [_fp] * 7),
    'cdrl_pwconv_fused_partial_rows': (_i, [_i, _i, _i, _i]),
    'cdrl_pwconv_fused': (_i, [_fp, _i, _i, _fp, _fp, _i, _i, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _fp, _fp, _fp, _fp]),
    'cdrl_pwconv_pack_elems': (_i64, [_i, _i]),
    'cdrl_pwconv_pack': (_i, [_fp, _i, _i, _i, _i, _fp, _i, _fp]),
    'cdrl_pwconv_fused_packed': (_i, [_fp, _i, _i, _fp, _fp, _i, _i, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _i, _fp]),
    'cdrl_pwconv_bn_bwd_packed': (_i, [_fp, _i, _i, _i, _i, _fp, _fp, _fp, _i, _i, _fp, _fp, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _i, _i, _i, _fp, _fp, _fp, _fp, _i, _fp]),
    'cdrl_pwconv_bn_bwd_workspace_bytes': (_i64, [_i, _i, _i, _i]),
    'cdrl_pwconv_bn_bwd': (_i, [_fp, _i, _i, _i, _i, _fp, _fp, _fp, _i, _i, _fp, _fp, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _i, _i, _i, _fp, _fp, _fp, _fp]),
    'cdrl_dwconv_bn_workspace_doubles': (_i64, [_i] * 6),
    'cdrl_dwconv_bn_fwd': (_i, [_fp] * 5 + [_i] * 6 + [_fp] * 4 + [_i, _fp, _fp, _fp]),
    'cdrl_dwconv_bn_bwd': (_i, [_fp] * 6 + [_i] * 6 + [_fp] * 11),
    'cdrl_maxpool_fwd': (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _fp]),
    'cdrl_maxpool_bwd': (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _fp]),
    'cdrl_bn_train_fwd': (_i, [_fp, _i, _i, _i, _fp, _fp, _fp, _fp, _i, _i, _fp, _i, _i, _i, _fp, _fp, _fp]),
    'cdrl_bn_train_bwd': (_i, [_fp, _i, _i, _i, _fp, _i, _i, _i, _fp, _i, _fp, _fp, _fp, _fp, _fp, _fp]),
    'cdrl_maxpool_bn_fwd': (_i, [_fp, _fp, _i, _i, _fp, _fp, _i, _i, _i, _i, _fp]),
    'cdrl_bn_train_bwd_pooled': (_i, [_fp, _fp, _i, _i, _fp, _i, _i, _i, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    'cdrl_beta_ppo_loss': (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _f, _f, _i, _i, _f, _fp, _fp, _fp, _fp, _fp]),
    'cdrl_value_loss': (_i, [_fp, _fp, _fp, _fp, _i, _f, _f, _fp, _fp, _fp, _fp]),
}

_lib = None


# Op-level entry points whose ACTIVATION tensors may be bf16 (configuration 3's storage): the C ABI takes the element type as an explicit
# `int act_type` argument (0 float32, 1 bf16) in front of the stream -- the library keeps no mode of its own (round 6; it had a
# thread-local switch).  The prototypes above list the arguments WITHOUT it; it is spliced in here, and the binding object below offers
# the tests a mode of its own (`lib.cdrl_set_op_activation_type(at)`, a Python attribute of the binding) that fills it in.
ACT_TYPE_BEFORE_STREAM = ('cdrl_pwconv_bwd_fused', 'cdrl_gemm_tn', 'cdrl_gemm_x3', 'cdrl_stem_fwd_stats', 'cdrl_stem_block_bwd',
                          'cdrl_stem_block_bwd_pooled', 'cdrl_pwconv_fused_packed', 'cdrl_pwconv_bn_bwd', 'cdrl_pwconv_bn_bwd_packed',
                          'cdrl_dwconv_bn_fwd', 'cdrl_dwconv_bn_bwd', 'cdrl_bn_train_fwd', 'cdrl_bn_train_bwd', 'cdrl_maxpool_bn_fwd')
ACT_TYPE_LAST = ('cdrl_pwconv_bwd_fused_workspace',)
for _n in ACT_TYPE_BEFORE_STREAM:
    _r, _a = PROTOTYPES[_n]
    PROTOTYPES[_n] = (_r, list(_a[:-1]) + [_i, _a[-1]])
for _n in ACT_TYPE_LAST:
    _r, _a = PROTOTYPES[_n]
    PROTOTYPES[_n] = (_r, list(_a) + [_i])


class Binding:
    """libcdrl_hip.so through ctypes.  Attribute access returns the C function; for the op-level entry points with an `act_type`
    argument it returns a wrapper that fills the argument from `self.act_type` (0 unless a test sets it), so that callers spell the
    float32 and the bf16-storage call alike.  `raw` is the ctypes library itself (every argument explicit)."""

    def __init__(self, cdll):
        object.__setattr__(self, 'raw', cdll)
        object.__setattr__(self, 'act_type', 0)

    def cdrl_set_op_activation_type(self, at: int) -> int:
        if at not in (0, 1):
            raise ValueError('activation type: 0 (float32) or 1 (bf16)')
        object.__setattr__(self, 'act_type', int(at))
        return 0

    def __getattr__(self, name):
        fn = getattr(self.raw, name)
        if name in ACT_TYPE_BEFORE_STREAM:
            return lambda *a: fn(*a[:-1], self.act_type, a[-1])
        if name in ACT_TYPE_LAST:
            return lambda *a: fn(*a, self.act_type)
        return fn


def load():
    """Loads libcdrl_hip.so; raises ImportError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950). There is no CPU fallback for the learner hot path.')
    # torch first: it brings its own HIP runtime (libamdhip64 under torch/lib).  libcdrl_hip.so must resolve its HIP symbols to
    # THAT runtime -- loaded the other way round the process ends up with two runtimes, and device memory / streams handed over
    # from torch are unknown to the one this library would have bound to.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = Binding(lib)
    return _lib


def env_overrides():
    """Every CDRL_* variable of the process environment as the LIBRARY sees it (cdrl_env_overrides), e.g. for a benchmark line."""
    buf = C.create_string_buffer(8192)
    n = load().cdrl_env_overrides(buf, len(buf))
    return buf.value.decode().split(' ') if n else []


def diag_active() -> int:
    """Number of wrong-result diagnostic switches (CDRL_DIAG_* with the master CDRL_DIAG=1) in effect."""
    return int(load().cdrl_diag_active())


def check(rc, what=''):
    if rc != 0:
        msg = load().cdrl_last_error()
        raise CdrlError(f'{what} failed (rc={rc}): {msg.decode() if msg else ""}')


def ptr(t):
    """Device (or host) address of a torch tensor / None."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())
