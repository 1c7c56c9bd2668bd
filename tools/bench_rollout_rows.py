#!/usr/bin/env python3
"""Measurements of the section-8(f) rows next to the learner hot path (the callers / data formats either side of it), each with
its CPU stand-in timed on the host beside it (the oracle = the reference's algorithm in PyTorch / numpy on the CPU):

  (f)1  rollout inference  CARLANetwork.predict for E environments (batched inference forward + device Beta sampling)
  (f)2  pathwise Beta sampling on the device (cdrl_beta_sample: sample + du/dalpha, du/dbeta)
  (f)3  rollout-time augmentation (cdrl_augment_images, the 'all' plan of tests/test_gpu_augment.py)
  A13   returns / GAE advantages of one rollout buffer (cdrl gae kernel vs the scipy.lfilter form of the oracle)
  (f)4  TensorFlow checkpoint-V2 writer + reader round trip of the three models (host code)

Prints one JSON object; tools/final_round.sh stores it as profiles/<round>_rollout_rows.json.
The CPU stand-ins are the oracle and therefore live in bench.py's baseline leg (`cpu_rollout_rows`, the only place besides tests/
and smoke() that may execute oracle/): `python3 bench.py --rollout-rows` runs this file with them, `python3
tools/bench_rollout_rows.py` alone reports the device side only."""
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carla_driving_rl_agent_amd import _lib, synthetic  # noqa: E402
from carla_driving_rl_agent_amd.core.carla_agent import CARLAgent, FakeCARLAEnvironment  # noqa: E402

DEV = 'cuda:0'


def dev_time(fn, iters, warm=3, repeats=3):
    """(device seconds, wall seconds) per call: best of `repeats` timed loops (the first loop of a process runs on cold clocks /
    a cold allocator: 2.7 vs 1.05 ms for predict at E = 1)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(repeats):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        r = (e0.elapsed_time(e1) / iters * 1e-3, (time.perf_counter() - t0) / iters)
        if best is None or r[1] < best[1]:
            best = r
    return best


def cpu_time(fn, iters, warm=1):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    return (time.perf_counter() - t0) / iters


def main(cpu=None):
    """cpu: optional object with predict(states) / beta(alpha, beta) / augment(stack, plan) / gae(rewards, values_be, gamma, lambda_)
    callables that run the CPU stand-in once (bench.py::cpu_rollout_rows)."""
    out = {}
    H, W, T = 90, 120, 4
    env = FakeCARLAEnvironment(image_shape=(H, W, 3), time_horizon=T, num_waypoints=5, vehicle_features=4, num_actions=2, seed=1)
    agent = CARLAgent(env, batch_size=32, log_mode=None, seed=1, aug_intensity=0.0)
    net = agent.network
    # ---- (f)1 rollout inference
    rows = []
    for E in (1, 8, 32, 128):
        r = synthetic.make_rollout(E, T=T, H=H, W=W, seed=E)
        st = {k: torch.as_tensor(v).to(DEV) for k, v in r['states'].items()}
        t_dev, t_wall = dev_time(lambda: net.predict(st), 30 if E <= 32 else 10)
        entry = dict(envs=E, device_ms=round(t_dev * 1e3, 3), wall_ms=round(t_wall * 1e3, 3), env_steps_per_s=round(E / t_wall, 1))
        if E <= 32 and cpu is not None:
            t_cpu = cpu_time(lambda: cpu.predict(r['states']), 3 if E <= 8 else 1)
            entry.update(cpu_oracle_ms=round(t_cpu * 1e3, 1), cpu_env_steps_per_s=round(E / t_cpu, 2))
        rows.append(entry)
    out['f1_rollout_inference'] = dict(what='CARLANetwork.predict: inference forward (moving statistics, old_policy + value heads) + '
                                            'cdrl_beta_sample_logp, inputs resident in HBM; wall = host time per call (launch-bound at small E)',
                                       image=[T, H, W, 3], rows=rows, cpu='oracle.predict, float32, 16 threads')
    # ---- (f)2 pathwise Beta sampling
    lib = _lib.load()
    S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    rows = []
    for n in (256, 65536, 4194304):
        a = torch.rand(n, 2, device=DEV) * 4 + 0.5
        b = torch.rand(n, 2, device=DEV) * 4 + 0.5
        u, da, db = (torch.empty(n, 2, device=DEV) for _ in range(3))
        t_dev, _ = dev_time(lambda: _lib.check(lib.cdrl_beta_sample(P(a), P(b), n, 2, 2, 7, 11, P(u), P(da), P(db), S())), 20)
        entry = dict(samples=2 * n, device_us=round(t_dev * 1e6, 1), samples_per_s=round(2 * n / t_dev))
        if n <= 65536 and cpu is not None:
            an, bn = a.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)
            t_cpu = cpu_time(lambda: cpu.beta(an, bn), 1, warm=0)
            entry.update(cpu_numpy_scipy_ms=round(t_cpu * 1e3, 1), cpu_samples_per_s=round(2 * n / t_cpu))
        rows.append(entry)
    out['f2_beta_sampling'] = dict(what='cdrl_beta_sample: Beta(alpha, beta) sample with implicit-reparameterisation gradients, Philox counter stream',
                                   rows=rows, cpu='synthetic.beta_sample_with_jacobian (numpy + scipy.special), 1 thread')
    # ---- (f)3 augmentation
    from carla_driving_rl_agent_amd.rl.augmentations import Augmenter, empty_plan
    w3 = list(np.random.default_rng(9).normal(1.0, 0.25, 27).astype(np.float32)) + [0.0] * 48
    plan = empty_plan(seed=0x1234, offset=5)
    plan.update(jitter=1, brightness=0.05, contrast=1.2, saturation=1.3, hue=0.07, blur_size=3, blur_kernel=w3, salt_pepper=1, gauss_noise=1,
                normalize=1, cutout_size=6, cutout_cell=3, dropout_size=81)
    rows = []
    for (h, w) in ((90, 120), (135, 180), (90, 360)):
        x = torch.rand(T, h, w, 3, device=DEV)
        aug = Augmenter(DEV)
        t_dev, t_wall = dev_time(lambda: aug(x, plan), 30)
        entry = dict(stack=[T, h, w, 3], device_us=round(t_dev * 1e6, 1), wall_us=round(t_wall * 1e6, 1), stacks_per_s=round(1.0 / t_wall, 1))
        if cpu is not None:
            xn = x.cpu().numpy()
            entry['cpu_numpy_ms'] = round(cpu_time(lambda: cpu.augment(xn, plan), 2, warm=0) * 1e3, 1)
        rows.append(entry)
    out['f3_augmentation'] = dict(what="cdrl_augment_images, every stage of the reference's pipeline switched on (colour jitter, blur, "
                                       'salt-and-pepper, Gaussian noise, normalisation, cutout, coarse dropout) on one observation stack',
                                  rows=rows, cpu='oracle/augment.py (numpy), 1 thread')
    # ---- A13 returns + GAE
    from carla_driving_rl_agent_amd.engine import gae_returns
    rows = []
    for n in (256, 4096, 65536):
        r = torch.randn(n + 1, device=DEV)
        v = torch.rand(n + 1, 2, device=DEV)
        t_dev, t_wall = dev_time(lambda: gae_returns(r, v, 0.9999, 0.999, 2.0), 30)
        entry = dict(timesteps=n, device_us=round(t_dev * 1e6, 1), wall_us=round(t_wall * 1e6, 1))
        if cpu is not None:
            rn32, vn32 = r.cpu().numpy().astype(np.float32), v.cpu().numpy().astype(np.float32)
            entry['cpu_us'] = round(cpu_time(lambda: cpu.gae(rn32, vn32, 0.9999, 0.999), 3) * 1e6, 1)
        rows.append(entry)
    out['a13_returns_gae'] = dict(what='returns + GAE(lambda) advantages + base/exponent decomposition of one rollout buffer (float64 scan, one launch)',
                                  rows=rows, cpu='oracle/gae.py (scipy.signal.lfilter form), 1 thread')
    # ---- (f)4 checkpoint round trip
    with tempfile.TemporaryDirectory() as d:
        agent2 = CARLAgent(FakeCARLAEnvironment(image_shape=(H, W, 3), time_horizon=T, num_waypoints=5, vehicle_features=4, num_actions=2, seed=1), batch_size=32, log_mode=None, seed=1,
                           aug_intensity=0.0, weights_dir=d, name='rows')
        t0 = time.perf_counter()
        agent2.network.save_weights()
        t_save = time.perf_counter() - t0
        size = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(d) for f in fs)
        t0 = time.perf_counter()
        agent2.network.load_weights()
        torch.cuda.synchronize()
        t_load = time.perf_counter() - t0
    out['f4_tf_checkpoint'] = dict(what='TensorFlow checkpoint-V2 (index SSTable + data shard, masked CRC-32C) of dynamics / policy / value '
                                        'written from and read back into the device arenas', bytes=size, save_ms=round(t_save * 1e3, 1),
                                   load_ms=round(t_load * 1e3, 1))
    # ---- the whole collect / update cycle through the reference's entry point (configs[0]: FakeCARLAEnvironment defaults)
    import contextlib
    import io
    env1 = FakeCARLAEnvironment(time_horizon=4, seed=3)
    ag = CARLAgent(env1, batch_size=32, log_mode=None, seed=3, skip_data=1, aug_intensity=0.0, optimization_steps=(1, 1))
    times = []
    for rep in range(3):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            t0 = time.perf_counter()
            ag.learn(episodes=1, timesteps=256, close=False)
            torch.cuda.synchronize()
            t_cycle = time.perf_counter() - t0
        upd = [float(l.split()[2][:-1]) for l in buf.getvalue().splitlines() if l.startswith('Update took')]
        times.append((t_cycle, upd[0] if upd else None))
    t_cycle, t_upd = min(times)
    out['c1_agent_cycle'] = dict(what='CARLAgent.learn(episodes=1, timesteps=256) on FakeCARLAEnvironment defaults (90x360x3, A=3, minibatch 32): '
                                      '256 x (observation H2D, predict, action D2H, env.step, memory append) + GAE + update() = 8 policy + 8 value '
                                      'minibatch passes (explicit index lists, device gathers); best of 3',
                                 cycle_s=round(t_cycle, 3), update_s=t_upd, rollout_ms_per_env_step=round((t_cycle - (t_upd or 0.0)) / 256 * 1e3, 3),
                                 update_ms_per_minibatch_pass=round((t_upd or 0.0) / 16 * 1e3, 3))
    print(json.dumps(out))


if __name__ == '__main__':
    main()
