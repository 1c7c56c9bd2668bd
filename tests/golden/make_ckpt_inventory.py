#!/usr/bin/env python3
"""Generates tests/golden/ref_ckpt_inventory.json from the reference's own checkpoints.

Run in the build container only (it reads /root/reference/weights, which does not
exist on the GPU box).  Output = DATA only: per checkpoint the (key, shape) list of
every tensor in the TF-checkpoint-V2 `.index` SSTable, plus summary statistics of the
policy/value tensors whose data shards are present.  No reference source is copied.

Format notes: footer = last 48 bytes (metaindex handle, index
handle, magic); blocks hold prefix-compressed entries + restart array; values are
BundleEntryProto (1 dtype, 2 shape, 3 shard, 4 offset, 5 size).
"""
import json
import os
import struct
import sys

import numpy as np

REF = '/root/reference/weights'


def varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7f) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def read_block(data, off, size):
    blk = data[off:off + size]
    nrestart = struct.unpack('<I', blk[-4:])[0]
    end = len(blk) - 4 - 4 * nrestart
    pos, key, out = 0, b'', []
    while pos < end:
        shared, pos = varint(blk, pos)
        non_shared, pos = varint(blk, pos)
        vlen, pos = varint(blk, pos)
        key = key[:shared] + blk[pos:pos + non_shared]
        pos += non_shared
        out.append((key, blk[pos:pos + vlen]))
        pos += vlen
    return out


def parse_proto(buf):
    """Minimal protobuf wire parser -> {field: [values]} (varint + length-delimited)."""
    pos, out = 0, {}
    while pos < len(buf):
        tag, pos = varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = varint(buf, pos)
        elif wt == 2:
            ln, pos = varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError(wt)
        out.setdefault(field, []).append(v)
    return out


def read_index(path):
    data = open(path, 'rb').read()
    footer = data[-48:]
    assert struct.unpack('<Q', footer[-8:])[0] == 0xdb4775248b80fb57
    pos = 0
    _, pos = varint(footer, pos)
    _, pos = varint(footer, pos)
    ioff, pos = varint(footer, pos)
    isize, pos = varint(footer, pos)
    entries = []
    for _, handle in read_block(data, ioff, isize):
        boff, p = varint(handle, 0)
        bsize, p = varint(handle, p)
        entries += read_block(data, boff, bsize)
    out = []
    for key, val in entries:
        if not key or key == b'_CHECKPOINTABLE_OBJECT_GRAPH':
            continue
        e = parse_proto(val)
        dtype = e.get(1, [0])[0]
        shape = []
        if 2 in e:
            for dim in parse_proto(e[2][0]).get(2, []):
                shape.append(parse_proto(dim).get(1, [0])[0])
        out.append(dict(key=key.decode(), dtype=dtype, shape=shape, shard=e.get(3, [0])[0],
                        offset=e.get(4, [0])[0], size=e.get(5, [0])[0]))
    return out


def read_object_graph(prefix):
    """{checkpoint key: variable full_name} from the _CHECKPOINTABLE_OBJECT_GRAPH string tensor (data-00000 shard)."""
    data = open(prefix + '.index', 'rb').read()
    footer = data[-48:]
    pos = 0
    _, pos = varint(footer, pos)
    _, pos = varint(footer, pos)
    ioff, pos = varint(footer, pos)
    isize, pos = varint(footer, pos)
    for _, handle in read_block(data, ioff, isize):
        boff, p = varint(handle, 0)
        bsize, p = varint(handle, p)
        for key, val in read_block(data, boff, bsize):
            if key != b'_CHECKPOINTABLE_OBJECT_GRAPH':
                continue
            e = parse_proto(val)
            raw = open(prefix + '.data-%05d-of-00002' % e.get(3, [0])[0], 'rb').read()
            raw = raw[e.get(4, [0])[0]:e.get(4, [0])[0] + e.get(5, [0])[0]]
            n, q = varint(raw, 0)
            out = {}
            for node in parse_proto(raw[q + 4:q + 4 + n]).get(1, []):
                for attr in parse_proto(node).get(2, []):
                    a = parse_proto(attr)
                    ck = a.get(3, [b''])[0].decode().replace('/.ATTRIBUTES/VARIABLE_VALUE', '')
                    out[ck] = a.get(2, [b''])[0].decode()
            return out
    return {}


def main():
    inv = {}
    for stage in sorted(os.listdir(REF)):
        for model in ('dynamics_model', 'policy_net', 'value_net'):
            idx = os.path.join(REF, stage, model + '.index')
            if not os.path.exists(idx):
                continue
            ents = [e for e in read_index(idx) if e['dtype'] == 1]
            rec = dict(tensors=[[e['key'].replace('/.ATTRIBUTES/VARIABLE_VALUE', ''), e['shape']] for e in ents],
                       total=int(sum(int(np.prod(e['shape'])) if e['shape'] else 1 for e in ents)))
            names = read_object_graph(os.path.join(REF, stage, model))
            rec['names'] = [names.get(k, '') for k, _ in rec['tensors']]       # variable full names, same order as `tensors`
            shard = os.path.join(REF, stage, model + '.data-00001-of-00002')
            if os.path.exists(shard) and os.path.getsize(shard) > 1024:
                raw = open(shard, 'rb').read()
                stats = []
                for e in ents:
                    a = np.frombuffer(raw[e['offset']:e['offset'] + e['size']], dtype='<f4')
                    stats.append([float(a.mean()), float(a.std()), float(a.min()), float(a.max())])
                rec['stats'] = stats
            inv[f'{stage}/{model}'] = rec
    # structure is identical across stages: keep one full listing + per-stage totals
    out = dict(source='/root/reference/weights/*/{dynamics_model,policy_net,value_net}.index',
               full={k.split('/')[1]: v for k, v in inv.items() if k.startswith('stage-s5-curriculum/')},
               totals={k: v['total'] for k, v in inv.items()})
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ref_ckpt_inventory.json')
    json.dump(out, open(dst, 'w'), indent=0)
    print('wrote', dst, {k: v['total'] for k, v in out['full'].items()})


if __name__ == '__main__':
    sys.exit(main())
