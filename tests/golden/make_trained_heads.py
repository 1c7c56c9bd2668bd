#!/usr/bin/env python3
"""Generates tests/golden/ref_trained_heads.npz from the reference's SHIPPED trained checkpoint.

Run in the build container only (it reads /root/reference/weights/stage-s5-curriculum/{policy_net,value_net}.*, the
checkpoint the reference's README evaluates; the trunk's data shard is not shipped).  Output = DATA only: the 20 + 20
tensors of the policy and value branches (core/networks.py:59-66,115-137,255-275) under this repository's parameter names,
float32 as stored.  They exercise the control branches on REALISTIC BatchNorm state (moving variances with mean ~36, trained
gammas), which random initialisation never produces.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from carla_driving_rl_agent_amd import tf_checkpoint as tfc   # noqa: E402

REF = '/root/reference/weights/stage-s5-curriculum'


def main():
    out = {}
    for model, stem in (('policy', 'policy_net'), ('value', 'value_net')):
        tensors = tfc.load_checkpoint(os.path.join(REF, stem))
        mapping = tfc.key_map(model)
        assert len(mapping) == 20
        for key, name in mapping.items():
            out[f'{model}/{name}'] = np.ascontiguousarray(tensors[key], dtype=np.float32)
    path = os.path.join(HERE, 'ref_trained_heads.npz')
    np.savez_compressed(path, **out)
    mv = [v for k, v in out.items() if k.endswith('moving_var')]
    print('wrote', path, os.path.getsize(path), 'bytes,', len(out), 'tensors; mean moving_var', float(np.mean([m.mean() for m in mv])))


if __name__ == '__main__':
    main()
