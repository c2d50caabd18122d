#!/usr/bin/env python3
"""GPU box: the env inside a learner-style loop -- a small MLP policy (obs -> 64 -> 64 -> action, tanh)
evaluated with PyTorch and CopterVecEnv.step() in ONE captured hipGraph, nothing on the host per step.
The policy is a stand-in (random weights); what is measured is how the stepper composes with
framework kernels on the same stream.   python tools/policy_loop.py [num_envs] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gym_copter_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
env = gym_copter_amd.make("Lander-v0", num_envs=n, seed=3, autoreset_mode="next_step")
dev = env.device
torch.manual_seed(0)
policy = torch.nn.Sequential(torch.nn.Linear(10, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(),
                             torch.nn.Linear(64, 4), torch.nn.Tanh()).to(dev).half()
obs, _ = env.reset()
ret = torch.zeros(n, device=dev)


def one_step():
    with torch.no_grad():
        a = policy(obs.half()).float()          # obs is the env's persistent output buffer
    o, r, term, trunc, _ = env.step(a)
    ret.add_(r)


chunk = 50
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        one_step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(chunk):
        one_step()
g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps // chunk):
    g.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / (steps // chunk * chunk)
print("policy MLP (fp16, 10-64-64-4) + env.step in one hipGraph: %d envs  %.2f us per step  %.2f G env-steps/s"
      % (n, dt * 1e6, n / dt / 1e9))
