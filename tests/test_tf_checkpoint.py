"""TF-checkpoint-V2 reader + Keras-order key mapping.  The mapping is pinned against the reference's
own checkpoint indices (committed inventory fixture: every `layer_with_weights-N/<var>` shape must equal
the shape of the engine parameter it maps to); reading real shards is exercised when the reference
checkout is present (build container only)."""
import json
import os

import numpy as np
import pytest

from carla_driving_rl_agent_amd import tf_checkpoint as tfc
from carla_driving_rl_agent_amd.engine import LearnerEngine

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/weights/stage-s5-curriculum'


@pytest.mark.parametrize('model,ckpt', [('trunk', 'dynamics_model'), ('policy', 'policy_net'), ('value', 'value_net')])
def test_keras_key_mapping_matches_reference_checkpoint_shapes(model, ckpt):
    inv = json.load(open(os.path.join(HERE, 'golden', 'ref_ckpt_inventory.json')))['full'][ckpt]['tensors']
    eng = LearnerEngine(1, device=None)
    shapes = {e['name']: tuple(e['shape']) for e in eng.tables[model].entries}
    mapping = tfc.key_map(model)
    assert len(mapping) == len(inv) == len(shapes)
    assert sorted(mapping.values()) == sorted(shapes)           # bijection onto the engine's parameters
    for key, shape in inv:
        assert key in mapping, key
        assert shapes[mapping[key]] == tuple(shape), (key, mapping[key])


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not present on this machine')
def test_reads_reference_policy_and_value_shards():
    stats = json.load(open(os.path.join(HERE, 'golden', 'ref_ckpt_inventory.json')))['full']
    for ckpt in ('policy_net', 'value_net'):
        t = tfc.load_checkpoint(os.path.join(REF, ckpt))
        keys = [k for k, _ in stats[ckpt]['tensors']]
        assert sorted(t) == sorted(keys)
        for (k, shape), (mean, std, mn, mx) in zip(stats[ckpt]['tensors'], stats[ckpt]['stats']):
            assert t[k].shape == tuple(shape)
            assert abs(t[k].mean() - mean) < 1e-6 + 1e-5 * abs(mean) and abs(t[k].max() - mx) < 1e-6
    # the trunk's data shard is not shipped (.MISSING_LARGE_BLOBS): the reader must not invent tensors
    assert tfc.load_checkpoint(os.path.join(REF, 'dynamics_model')) == {}
