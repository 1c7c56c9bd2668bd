"""Reader for the reference's TensorFlow checkpoint-V2 files (`policy_net`, `value_net`,
`dynamics_model`: `<name>.index` SSTable + `<name>.data-0000N-of-0000M` shards) without TensorFlow,
and the mapping of Keras' `layer_with_weights-N/<var>` keys onto the engine's parameter names.

This is what lets the checkpoints the reference ships under `weights/stage-*/` (written by
`CARLANetwork.save_weights`, reference core/networks.py:297-300) load into the native learner.
Format notes: SURVEY.md Appendix F.  Keras numbers the weighted layers of a functional model in
topological (depth) order, which interleaves the two branches of the stride-2 ShuffleNet units and
the per-modality feature nets; `keras_layer_order` reproduces that order (pinned against the shipped
checkpoint indices in tests/test_tf_checkpoint.py).
"""
import os
import struct
from typing import Dict, List, Tuple

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_SUFFIX = '/.ATTRIBUTES/VARIABLE_VALUE'


def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7f) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _block(data, off, size):
    blk = data[off:off + size]
    nrestart = struct.unpack('<I', blk[-4:])[0]
    end = len(blk) - 4 - 4 * nrestart
    pos, key, out = 0, b'', []
    while pos < end:
        shared, pos = _varint(blk, pos)
        non_shared, pos = _varint(blk, pos)
        vlen, pos = _varint(blk, pos)
        key = key[:shared] + blk[pos:pos + non_shared]
        pos += non_shared
        out.append((key, blk[pos:pos + vlen]))
        pos += vlen
    return out


def _proto(buf):
    pos, out = 0, {}
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            v, pos = buf[pos:pos + 4], pos + 4
        elif wt == 1:
            v, pos = buf[pos:pos + 8], pos + 8
        else:
            raise ValueError(f'unsupported protobuf wire type {wt}')
        out.setdefault(field, []).append(v)
    return out


def read_index(path: str) -> List[dict]:
    """[{key, shape, dtype, shard, offset, size}] for every float tensor of `<prefix>.index`."""
    data = open(path, 'rb').read()
    footer = data[-48:]
    if struct.unpack('<Q', footer[-8:])[0] != _MAGIC:
        raise ValueError(f'{path}: not a TF checkpoint index (bad magic)')
    pos = 0
    _, pos = _varint(footer, pos)
    _, pos = _varint(footer, pos)
    ioff, pos = _varint(footer, pos)
    isize, pos = _varint(footer, pos)
    entries = []
    for _, handle in _block(data, ioff, isize):
        boff, p = _varint(handle, 0)
        bsize, p = _varint(handle, p)
        entries += _block(data, boff, bsize)
    out = []
    for key, val in entries:
        if not key or key == b'_CHECKPOINTABLE_OBJECT_GRAPH':
            continue
        e = _proto(val)
        if e.get(1, [0])[0] != 1:           # DT_FLOAT only
            continue
        shape = [_proto(d).get(1, [0])[0] for d in _proto(e[2][0]).get(2, [])] if 2 in e else []
        out.append(dict(key=key.decode().replace(_SUFFIX, ''), shape=tuple(shape), shard=e.get(3, [0])[0],
                        offset=e.get(4, [0])[0], size=e.get(5, [0])[0]))
    return out


def load_checkpoint(prefix: str) -> Dict[str, np.ndarray]:
    """key -> float32 array for every tensor whose data shard is present."""
    entries = read_index(prefix + '.index')
    nshards = 1 + max(e['shard'] for e in entries)
    shards = {}
    for s in range(nshards):
        for total in range(nshards, nshards + 8):
            p = f'{prefix}.data-{s:05d}-of-{total:05d}'
            if os.path.exists(p):
                shards[s] = open(p, 'rb').read()
                break
    out = {}
    for e in entries:
        raw = shards.get(e['shard'])
        if raw is None or e['offset'] + e['size'] > len(raw):
            continue
        out[e['key']] = np.frombuffer(raw[e['offset']:e['offset'] + e['size']], dtype='<f4').reshape(e['shape']).copy()
    return out


# ------------------------------------------------------------------------------------------------
# Keras variable naming -> engine parameter names
# ------------------------------------------------------------------------------------------------
_BN = (('gamma', 'gamma'), ('beta', 'beta'), ('moving_mean', 'moving_mean'), ('moving_variance', 'moving_var'))
_CONV = (('kernel', 'w'), ('bias', 'b'))
_DW = (('depthwise_kernel', 'w'), ('bias', 'b'))
_GRU = (('cell/kernel', 'kernel'), ('cell/recurrent_kernel', 'recurrent'), ('cell/bias', 'bias'))


def keras_layer_order(model: str, stage_n=(4, 8, 4)) -> List[Tuple[str, tuple]]:
    """[(engine layer prefix, ((keras var, engine var), ...))] in `layer_with_weights-N` order."""
    if model in ('policy', 'value'):
        p = 'pi' if model == 'policy' else 'v'
        heads = ('alpha', 'beta', 'similarity', 'speed') if model == 'policy' else ('base', 'exp', 'speed', 'similarity')
        return [(f'{p}.bn0', _BN), (f'{p}.fc0', _CONV), (f'{p}.bn1', _BN), (f'{p}.fc1', _CONV)] + \
               [(f'{p}.{h}', _CONV) for h in heads]
    if model != 'trunk':
        raise ValueError(model)
    order = [('img.stem.conv', _CONV), ('img.stem.bn', _BN)]
    for s, n in enumerate(stage_n):
        for u in range(n):
            pre = f'img.s{s}.u{u}'
            if u == 0:      # stride 2: the 4-layer shortcut chain is depth-aligned with the last 4 main layers
                order += [(f'{pre}.pw1', _CONV), (f'{pre}.bn1', _BN), (f'{pre}.sc_dw', _DW), (f'{pre}.dw', _DW),
                          (f'{pre}.sc_bn1', _BN), (f'{pre}.bn2', _BN), (f'{pre}.sc_pw', _CONV), (f'{pre}.pw2', _CONV),
                          (f'{pre}.sc_bn2', _BN), (f'{pre}.bn3', _BN)]
            else:
                order += [(f'{pre}.pw1', _CONV), (f'{pre}.bn1', _BN), (f'{pre}.dw', _DW), (f'{pre}.bn2', _BN),
                          (f'{pre}.pw2', _CONV), (f'{pre}.bn3', _BN)]
    mods = ('road', 'vehicle', 'navigation')
    order += [('img.head.conv', _CONV)] + [(f'{m}.fc0', _CONV) for m in mods]
    order += [('img.head.bn', _BN)] + [(f'{m}.bn0', _BN) for m in mods]
    order += [(f'{m}.fc1', _CONV) for m in mods] + [(f'{m}.bn1', _BN) for m in mods]
    order += [('gru_image', _GRU)] + [(f'gru_{m}', _GRU) for m in mods]
    order += [('dyn.bn', _BN), ('dyn.fc', _CONV)]
    return order


def key_map(model: str, stage_n=(4, 8, 4)) -> Dict[str, str]:
    """Keras checkpoint key -> engine parameter name."""
    out = {}
    for i, (prefix, variables) in enumerate(keras_layer_order(model, stage_n)):
        for kv, ev in variables:
            out[f'layer_with_weights-{i}/{kv}'] = f'{prefix}.{ev}'
    return out


def load_into_engine(engine, model: str, prefix: str, strict=True) -> List[str]:
    """Loads `<prefix>.index/.data-*` into the engine's `model` ('policy' | 'value' | 'trunk') arena.
    Returns the names that were loaded; with strict=True every engine parameter must be present
    (the reference repository ships the trunk's index but not its data shard)."""
    tensors = load_checkpoint(prefix)
    mapping = key_map(model, tuple(engine.cfg.stage_n))
    values, loaded = {}, []
    expect = {e['name']: e['shape'] for e in engine.tables[model].entries}
    for k, name in mapping.items():
        if k in tensors:
            if tuple(tensors[k].shape) != tuple(expect[name]):
                raise ValueError(f'{k} -> {name}: shape {tensors[k].shape} != {expect[name]}')
            values[name] = tensors[k]
            loaded.append(name)
    missing = sorted(set(expect) - set(values))
    if missing and strict:
        raise KeyError(f'{prefix}: {len(missing)} tensors missing (e.g. {missing[:3]})')
    views = engine.param_views(model)
    import torch
    for name, v in values.items():
        views[name].copy_(torch.as_tensor(v).reshape(views[name].shape))
    if model == 'policy':
        engine.update_old_policy()
    return loaded
