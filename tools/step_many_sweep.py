#!/usr/bin/env python3
"""GPU box: throughput of cs_step_many (K steps per launch) over batch size and K."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gym_copter_amd  # noqa: E402

dev = torch.device("cuda", 0)
for law in ("uniform", "near_hover"):
    for N in (65536, 262144, 1048576):
        for K in (10, 100):
            env = gym_copter_amd.CopterVecEnv("lander3d", N, seed=1, autoreset_mode="next_step")
            env.reset()
            a = (torch.rand((K, N, 4), device=dev) * 2 - 1) if law == "uniform" \
                else 0.01656 * (1 + 0.01 * torch.randn((K, N, 4), device=dev))
            for _ in range(3):
                env.step_many(a)
            torch.cuda.synchronize()
            R = max(2, 2000 // K) if N <= 262144 else max(2, 300 // K)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(R):
                env.step_many(a)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e-3
            print("lander3d %-10s N %8d K %4d  us/step %7.3f  G env-steps/s %6.2f"
                  % (law, N, K, t / (R * K) * 1e6, N * R * K / t / 1e9), flush=True)
            env.close()
