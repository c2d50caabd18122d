#!/usr/bin/env python3
"""GPU box: the PCIe-inclusive rate of the NumPy convenience path (actions host -> device, every
output device -> host, per step) next to the resident rate.  python tools/host_io.py [num_envs]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gym_copter_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(0)
acts = [rng.uniform(-1, 1, (n, 4)).astype(np.float32) for _ in range(8)]
K = 300
for copy in (True, False):
    env = gym_copter_amd.make("Lander-v0", num_envs=n, seed=1, autoreset_mode="next_step", copy=copy)
    env.reset()
    for j in range(20):
        env.step(acts[j % 8])
    t0 = time.perf_counter()
    for j in range(K):
        obs, r, term, trunc, _ = env.step(acts[j % 8])      # NumPy in -> NumPy out (synchronous)
    dt = (time.perf_counter() - t0) / K
    assert isinstance(obs, np.ndarray)
    print("NumPy in/out (PCIe-inclusive), copy=%s: %d envs  %.1f us/step  %.3f G env-steps/s" % (copy, n, dt * 1e6, n / dt / 1e9))
dev = [torch.from_numpy(a).to(env.device) for a in acts]
torch.cuda.synchronize()
t0 = time.perf_counter()
for j in range(K):
    env.step(dev[j % 8])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("device tensors, eager launches: %d envs  %.1f us/step  %.3f G env-steps/s" % (n, dt * 1e6, n / dt / 1e9))
