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


def _raw_entries(path: str):
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
    return entries


def _shard_path(prefix: str, shard: int):
    for total in range(1, 64):
        p = f'{prefix}.data-{shard:05d}-of-{total:05d}'
        if os.path.exists(p):
            return p
    return None


def read_object_graph(prefix: str) -> Dict[str, str]:
    """{checkpoint key (without the /.ATTRIBUTES suffix): variable full_name} from the `_CHECKPOINTABLE_OBJECT_GRAPH`
    string tensor (a serialized TrackableObjectGraph: nodes[] -> attributes[] = {name, full_name, checkpoint_key}).
    The full names are Keras' creation-order layer names (`conv2d_17/kernel`, `v-speed-0/kernel`, ...): they identify
    WHICH layer a `layer_with_weights-N` slot holds independently of tensor shapes."""
    for key, val in _raw_entries(prefix + '.index'):
        if key != b'_CHECKPOINTABLE_OBJECT_GRAPH':
            continue
        e = _proto(val)
        path = _shard_path(prefix, e.get(3, [0])[0])
        if path is None:
            return {}
        off, size = e.get(4, [0])[0], e.get(5, [0])[0]
        raw = open(path, 'rb').read()[off:off + size]
        n, pos = _varint(raw, 0)                    # string tensor: varint length, 4-byte masked crc32c of the lengths, bytes
        graph = _proto(raw[pos + 4:pos + 4 + n])
        out = {}
        for node in graph.get(1, []):
            for attr in _proto(node).get(2, []):
                a = _proto(attr)
                ck = a.get(3, [b''])[0].decode()
                if ck.endswith(_SUFFIX):
                    out[ck[:-len(_SUFFIX)]] = a.get(2, [b''])[0].decode()
        return out
    return {}


def read_index(path: str) -> List[dict]:
    """[{key, shape, dtype, shard, offset, size}] for every float tensor of `<prefix>.index`."""
    entries = _raw_entries(path)
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
# writer (what Keras `save_weights(..., save_format='tf')` leaves on disk; reference core/networks.py:297-300)
# ------------------------------------------------------------------------------------------------
def _crc32c(data: bytes, crc: int = 0) -> int:
    from . import _lib
    import ctypes as C
    return int(_lib.load().cdrl_crc32c(crc, C.c_char_p(data) if isinstance(data, bytes) else data, len(data)))


def _mask(crc: int) -> int:
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


def _put_varint(n: int) -> bytes:
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7f) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def _field(num: int, wt: int, payload) -> bytes:
    tag = _put_varint((num << 3) | wt)
    if wt == 0:
        return tag + _put_varint(payload)
    if wt == 2:
        return tag + _put_varint(len(payload)) + payload
    if wt == 5:
        return tag + struct.pack('<I', payload)
    raise ValueError(wt)


def _bundle_entry(dtype: int, shape, shard: int, offset: int, size: int, crc: int) -> bytes:
    dims = b''.join(_field(2, 2, _field(1, 0, int(d))) for d in shape)
    out = _field(1, 0, dtype) + _field(2, 2, dims)
    if shard:
        out += _field(3, 0, shard)
    if offset:
        out += _field(4, 0, offset)
    return out + _field(5, 0, size) + _field(6, 5, crc)


def _table_block(entries, restart_interval=16) -> bytes:
    """LevelDB-format block: prefix-compressed entries, uint32 restart offsets, uint32 restart count."""
    out, restarts, last = bytearray(), [], b''
    for i, (key, val) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(key), len(last)) and key[shared] == last[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(val)) + key[shared:] + val
        last = key
    if not restarts:
        restarts = [0]
    return bytes(out) + b''.join(struct.pack('<I', r) for r in restarts) + struct.pack('<I', len(restarts))


def _short_successor(key: bytes) -> bytes:
    for i, b in enumerate(key):
        if b != 0xff:
            return key[:i] + bytes([b + 1])
    return key


def build_object_graph(keys_to_full_names: Dict[str, str]) -> bytes:
    """Minimal TrackableObjectGraph for a Keras functional model: root -> `layer_with_weights-N` -> (`cell` ->) variable
    nodes whose single attribute is {VARIABLE_VALUE, full_name, checkpoint_key} (the restore matches checkpoint and model
    nodes by these child paths)."""
    nodes = [dict(children=[], attrs=[])]

    def child(parent, name):
        for cid, cname in nodes[parent]['children']:
            if cname == name:
                return cid
        nodes.append(dict(children=[], attrs=[]))
        nodes[parent]['children'].append((len(nodes) - 1, name))
        return len(nodes) - 1
    order = sorted(keys_to_full_names, key=lambda k: (int(k.split('/')[0].rsplit('-', 1)[1]), k))
    for key in order:
        node = 0
        for part in key.split('/'):
            node = child(node, part)
        nodes[node]['attrs'].append(('VARIABLE_VALUE', keys_to_full_names[key], key + _SUFFIX))
    out = b''
    for nd in nodes:
        body = b''.join(_field(1, 2, _field(1, 0, cid) + _field(2, 2, name.encode())) if cid else
                        _field(1, 2, _field(2, 2, name.encode())) for cid, name in nd['children'])
        body += b''.join(_field(2, 2, _field(1, 2, a.encode()) + _field(2, 2, f.encode()) + _field(3, 2, c.encode()))
                         for a, f, c in nd['attrs'])
        out += _field(1, 2, body)
    return out


def save_checkpoint(prefix: str, tensors: Dict[str, np.ndarray], full_names: Dict[str, str] = None,
                    object_graph: bytes = None, order: List[str] = None):
    """Writes `<prefix>.index`, `<prefix>.data-00000-of-00002` (object graph) and `<prefix>.data-00001-of-00002` (all
    float32 tensors in key order) -- the shard layout of the reference's shipped checkpoints.  `tensors` maps checkpoint keys
    (`layer_with_weights-N/<var>`) to arrays.  Re-encoding the reference's own policy_net / value_net reproduces its three
    files byte for byte (tests/test_tf_checkpoint.py); TensorFlow itself is not installable here, so a restore by Keras
    is not exercised.  `order`: keys in the order their bytes are laid out (default: iteration order of `tensors`)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    if object_graph is None:
        names = full_names or {k: k for k in tensors}
        object_graph = build_object_graph({k: names.get(k, k) for k in tensors})
    # shard 0: the object graph as a scalar DT_STRING tensor = varint length, masked crc32c of the length bytes, payload
    lens = _put_varint(len(object_graph))
    crc = _crc32c(struct.pack('<I', len(object_graph)))          # lengths are checksummed as uint32 values
    len_ck = struct.pack('<I', _mask(crc))
    crc = _crc32c(object_graph, _crc32c(len_ck, crc))
    shard0 = lens + len_ck + object_graph
    entries = [(b'', _field(1, 0, 2) + _field(3, 2, _field(1, 0, 1))),           # BundleHeaderProto: 2 shards, version.producer 1
               (b'_CHECKPOINTABLE_OBJECT_GRAPH', _bundle_entry(7, (), 0, 0, len(shard0), _mask(crc)))]
    shard1 = bytearray()
    # tensor bytes are laid out in the order the saver visits the variables (object-graph order: layer by layer, variables in
    # Keras' attribute order), NOT in key order; `tensors` / `order` carry that order
    for key in (order if order is not None else list(tensors)):
        a = np.ascontiguousarray(tensors[key], dtype='<f4')
        raw = a.tobytes()
        entries.append(((key + _SUFFIX).encode(), _bundle_entry(1, a.shape, 1, len(shard1), len(raw), _mask(_crc32c(raw)))))
        shard1 += raw
    entries.sort(key=lambda e: e[0])

    def with_trailer(block: bytes) -> bytes:
        return block + b'\x00' + struct.pack('<I', _mask(_crc32c(b'\x00', _crc32c(block))))
    data_block = _table_block(entries)
    meta_block = _table_block([])
    index = bytearray(with_trailer(data_block))
    meta_off = len(index)
    index += with_trailer(meta_block)
    index_block = _table_block([(_short_successor(entries[-1][0]), _put_varint(0) + _put_varint(len(data_block)))], 1)
    index_off = len(index)
    index += with_trailer(index_block)
    footer = _put_varint(meta_off) + _put_varint(len(meta_block)) + _put_varint(index_off) + _put_varint(len(index_block))
    index += footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', _MAGIC)
    # data shards first, the index last, each through a temporary file + rename: a reader (or a second writer) never sees
    # a torn file, and an index never points at shards that are not there yet
    for suffix, blob in (('.data-00000-of-00002', shard0), ('.data-00001-of-00002', bytes(shard1)), ('.index', bytes(index))):
        tmp = f'{prefix}{suffix}.tmp{os.getpid()}'
        try:
            with open(tmp, 'wb') as f:
                f.write(blob)
            os.replace(tmp, prefix + suffix)
        except BaseException:
            try:                            # no temporary file left behind by a failed write (disk full, permissions)
                os.unlink(tmp)
            except OSError:
                pass
            raise


def keras_full_names(model: str, stage_n=(4, 8, 4)) -> Dict[str, str]:
    """checkpoint key -> Keras variable full_name for a freshly built reference model (creation-order layer names:
    main branch of a unit before its shortcut, feature nets road / vehicle / navigation, named heads)."""
    count = {}

    def new(kind):
        i = count.get(kind, 0)
        count[kind] = i + 1
        return kind if i == 0 else f'{kind}_{i}'
    layer = {}
    kinds = {_CONV: 'conv2d', _DW: 'depthwise_conv2d', _BN: 'batch_normalization', _GRU: 'gru'}
    if model == 'trunk':
        created = [('img.stem.conv', _CONV), ('img.stem.bn', _BN)]
        for s, n in enumerate(stage_n):
            for u in range(n):
                pre = f'img.s{s}.u{u}'
                created += [(f'{pre}.pw1', _CONV), (f'{pre}.bn1', _BN), (f'{pre}.dw', _DW), (f'{pre}.bn2', _BN), (f'{pre}.pw2', _CONV),
                            (f'{pre}.bn3', _BN)]
                if u == 0:
                    created += [(f'{pre}.sc_dw', _DW), (f'{pre}.sc_bn1', _BN), (f'{pre}.sc_pw', _CONV), (f'{pre}.sc_bn2', _BN)]
        created += [('img.head.conv', _CONV), ('img.head.bn', _BN)]
        for prefix, var in created:
            layer[prefix] = new(kinds[var])
        for m in ('road', 'vehicle', 'navigation'):
            layer[f'{m}.fc0'], layer[f'{m}.fc1'] = new('dense'), new('dense')
            layer[f'{m}.bn0'], layer[f'{m}.bn1'] = new('batch_normalization'), new('batch_normalization')
        for m in ('image', 'road', 'vehicle', 'navigation'):
            layer[f'gru_{m}'] = new('gru')
        layer['dyn.bn'] = new('batch_normalization')
        layer['dyn.fc'] = 'dynamics-linear'
    else:
        p = 'pi' if model == 'policy' else 'v'
        layer = {f'{p}.bn0': 'batch_normalization', f'{p}.fc0': 'dense', f'{p}.bn1': 'batch_normalization_1', f'{p}.fc1': 'dense_1'}
        heads = dict(alpha='alpha-0', beta='beta-0', similarity='pi-similarity-0', speed='pi-speed-0') if model == 'policy' else \
            dict(base='v-base-0', exp='v-exp-0', similarity='v-similarity-0', speed='v-speed-0')
        layer.update({f'{p}.{h}': n for h, n in heads.items()})
    out = {}
    for i, (prefix, variables) in enumerate(keras_layer_order(model, stage_n)):
        for kv, _ in variables:
            lname = layer[prefix]
            if variables is _GRU:
                idx = lname[3:]                        # 'gru_2' -> cell name 'gru_cell_2'
                out[f'layer_with_weights-{i}/{kv}'] = f'{lname}/gru_cell{idx}/{kv.split("/")[1]}'
            else:
                out[f'layer_with_weights-{i}/{kv}'] = f'{lname}/{kv}'
    return out


def save_from_engine(engine, model: str, prefix: str):
    """Engine arena of `model` -> TF checkpoint-V2 files under `prefix` (inverse of load_into_engine)."""
    values = engine.export_params(model)
    mapping = key_map(model, tuple(engine.cfg.stage_n))
    save_checkpoint(prefix, {k: values[name] for k, name in mapping.items()}, keras_full_names(model, tuple(engine.cfg.stage_n)))


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
        # Keras flattens a dict of outputs in sorted-key order: ..., similarity, speed (both models; pinned against the variable
        # names in the shipped checkpoints' object graphs: v-similarity-0 = layer_with_weights-6, v-speed-0 = -7)
        heads = ('alpha', 'beta', 'similarity', 'speed') if model == 'policy' else ('base', 'exp', 'similarity', 'speed')
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
