"""MI355X-native PPO learner hot path of carla-driving-rl-agent (HIP kernels behind a C ABI).

Sub-modules
  _lib      ctypes binding of libcdrl_hip.so (fails loudly when the library is missing)
  engine    LearnerEngine: parameter arenas + workspace as torch tensors, step functions
  synthetic deterministic synthetic rollout buffers
  core, rl  host-side mirror of the reference's CARLAgent / CARLANetwork / PPOMemory API
"""
__version__ = '0.1.0'
