#!/usr/bin/env python3
"""Diagnostic (GPU box): host cost of the eager Python step path (no hipGraph) vs the kernel's own pace.
  python tools/eager_rate.py [num_envs]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import gym_copter_amd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = gym_copter_amd.CopterVecEnv("lander3d", N, seed=1, autoreset_mode="next_step")
env.reset()
ring = [torch.rand((N, 4), device=env.device) * 2 - 1 for _ in range(64)]
def rate(label):
    for j in range(2000):
        env.step(ring[j & 63])
    torch.cuda.synchronize()
    K = 20000
    t0 = time.perf_counter()
    for j in range(K):
        env.step(ring[j & 63])
    t1 = time.perf_counter()           # host has enqueued everything
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("eager env.step (%s): host enqueue %.2f us per step, wall %.2f us per step (%d envs)"
          % (label, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6, N))


fast = env._fast
for rep in range(2):
    env._fast = fast
    rate("_cs_call" if fast else "ctypes (module not built)")
    env._fast = None
    rate("ctypes")
env._fast = fast
import cProfile
import pstats
pr = cProfile.Profile()
pr.enable()
for j in range(5000):
    env.step(ring[j & 63])
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
