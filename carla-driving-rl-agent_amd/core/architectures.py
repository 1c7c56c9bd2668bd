"""Architecture descriptions of the time-distributed sub-networks.

Same entry points as the reference's core/architectures.py (`shufflenet_v2`, `feature_net`), but
instead of building Keras layers they return plain *spec dicts* that CARLANetwork hands to the
native engine (cdrl_config).  Unsupported options fail loudly instead of silently changing the
network."""

SHUFFLENET_CHANNELS = {0.5: [48, 96, 192], 1.0: [116, 232, 464], 1.5: [176, 352, 704], 2.0: [244, 488, 976]}


def shufflenet_v2(inputs, time_horizon: int, g=1.0, leak=0.0, last_channels=1024) -> dict:
    """inputs: image shape (H, W, 3) per time slice.  Stages of 4 / 8 / 4 units, stem of 24 channels."""
    if g not in SHUFFLENET_CHANNELS:
        raise ValueError(f'g must be one of {sorted(SHUFFLENET_CHANNELS)}')
    if leak != 0.0:
        raise NotImplementedError('leaky ReLU6 (leak != 0) is not implemented in the HIP kernels')
    h, w, c = tuple(inputs)
    if c != 3:
        raise ValueError('the stem kernel expects RGB images (3 channels)')
    return dict(kind='shufflenet_v2', time_horizon=time_horizon, image=(h, w, c), stem=24,
                stage_c=list(SHUFFLENET_CHANNELS[g]), stage_n=[4, 8, 4], last=int(last_channels))


def feature_net(inputs, time_horizon: int, units=32, num_layers=2, activation='relu', normalization=None) -> dict:
    """inputs: feature-vector shape (D,) per time slice; [Dense(units, activation) -> BatchNorm] x num_layers."""
    name = getattr(activation, '__name__', activation)
    if num_layers != 2:
        raise NotImplementedError('the engine builds feature nets with exactly 2 layers (reference default)')
    if name not in ('relu6',):
        raise NotImplementedError(f'feature-net activation {name!r}: only relu6 (CARLAgent.DEFAULT_DYNAMICS) is implemented')
    if normalization is not None:
        raise NotImplementedError('input normalization of feature nets is unused by CARLAgent and not implemented')
    return dict(kind='feature_net', time_horizon=time_horizon, dim=int(tuple(inputs)[0]), units=int(units))
