#!/usr/bin/env python3
"""Diagnostic (GPU box): host cost of one CopterVecEnv.step() call, through the _cs_call module and through
ctypes, on a 64-env context (the kernel is far shorter than the call, so the loop is host-bound).
  python tools/enqueue_cost.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gym_copter_amd  # noqa: E402

env = gym_copter_amd.CopterVecEnv("lander3d", 64, seed=1, autoreset_mode="next_step")
env.reset()
a = torch.rand((64, 4), device=env.device) * 2 - 1
for label, fast in (("_cs_call", env._fast), ("ctypes", None)):
    env._fast = fast
    for j in range(2000): env.step(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(2000): env.step(a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("python 64 envs (%s): host enqueue %.2f us per step" % (label, (t1 - t0) / 2000 * 1e6))
