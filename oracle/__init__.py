"""CPU oracle for the PPO learner hot path (TEST INFRASTRUCTURE ONLY).

This package is a PyTorch-CPU / numpy restatement of the reference's learner
algorithm (CARLANetwork fwd/bwd, losses, per-tensor clip + Keras-Adam, GAE).
It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product path
(``carla-driving-rl-agent_amd``) never imports it and fails loudly when the HIP
library is missing.

PARITY STATUS: **parity unpinned at the TensorFlow boundary.**  The reference
ships no tests / golden vectors, and tensorflow==2.3.1, tensorflow-probability
==0.11.1, gym and carla cannot be imported in the build container, so the
TF/TFP default semantics restated here could not be
executed.  What *is* pinned:
  * structure  - parameter names/shapes/counts against the reference's own
    checkpoint indices (tests/golden/ref_ckpt_inventory.json, generated from
    /root/reference/weights by tests/golden/make_ckpt_inventory.py);
  * GAE / returns - against scipy.signal.lfilter itself (the routine the
    reference calls at rl/utils.py:59), which is importable here;
  * the tower forward - against a second, independently written numpy forward
    (oracle/np_tower.py).
"""
