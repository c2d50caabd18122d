#!/usr/bin/env python3
"""Diagnostic (GPU box): what each per-step output costs at one batch size -- the same step kernel with
output pointers withheld (cs_step accepts NULL for each), hipGraph replay of 100-step chunks.
  python tools/store_cost.py [num_envs]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import gym_copter_amd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = gym_copter_amd.CopterVecEnv("lander3d", N, seed=1, autoreset_mode="next_step")
env.reset()
dev = env.device
ring = torch.rand((64, N, 4), device=dev) * 2 - 1
lib = env._lib
P = lambda t: C.c_void_p(t.data_ptr())
# four separate arrays (the wrapper's default outputs are the columns of one packed array, which cannot be withheld
# one by one), plus the default packed rows as their own variant
sep = (torch.empty((N, env.obs_dim), device=dev), torch.empty(N, device=dev),
       torch.empty(N, dtype=torch.uint8, device=dev), torch.empty(N, dtype=torch.uint8, device=dev))
full = dict(obs=P(sep[0]), reward=P(sep[1]), term=P(sep[2]), trunc=P(sep[3]))
variants = {"packed rows (default)": dict(obs=P(env._obs), reward=P(env._reward), term=P(env._term), trunc=P(env._trunc)),
            "all outputs": full,
            "no truncated": dict(full, trunc=None),
            "no terminated, no truncated": dict(full, trunc=None, term=None),
            "no reward": dict(full, reward=None),
            "no observation rows": dict(full, obs=None),
            "no outputs at all": dict(obs=None, reward=None, term=None, trunc=None)}
res = {}
for rep in range(3):
    for name, v in variants.items():
        def one(j):
            lib.cs_step(env._ctx, P(ring[j % 64]), v["obs"], v["reward"], v["term"], v["trunc"], env._stream())
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for j in range(3):
                one(j)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for j in range(100):
                one(j)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(name, []).append(e0.elapsed_time(e1) / 20000 * 1e3)
for name, t in res.items():
    print("%-30s us per step: %s" % (name, " ".join("%.3f" % x for x in t)))
