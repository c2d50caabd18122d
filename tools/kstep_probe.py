#!/usr/bin/env python3
"""tools/kstep_probe.py -- a minimal target for profilers (rocprofv3 --att / --pmc / --kernel-trace): L launches of ONE
K-step leg and nothing else.

    python3 tools/kstep_probe.py <pid|many|random|step> [envs=65536] [K=100] [launches=5] [substeps=1]

Prints one JSON line with the HIP-event time per env step (un-profiled figure when run bare).
    rocprofv3 --att -d gpurun_out/att_pid -- python3 tools/kstep_probe.py pid 65536 4 2
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    leg = sys.argv[1] if len(sys.argv) > 1 else "pid"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    L = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    nsub = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    import torch
    import gym_copter_amd as gca
    env = gca.CopterVecEnv(task="lander3d", num_envs=n, device=0, seed=1234, autoreset_mode="next_step", substeps=nsub)
    env.reset()
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    if leg == "many":
        acts = torch.rand((K, n, 4), generator=g, device="cuda") * 2 - 1
        call = lambda: env.step_many(acts)
    elif leg == "pid":
        env.configure_pid()
        env.reset()
        call = lambda: env.rollout_pid(K)
    elif leg == "random":
        call = lambda: env.rollout_random(K)
    else:
        hover = 0.016560178185018043
        acts = hover * (1 + 0.01 * torch.randn((8, n, 4), generator=g, device="cuda"))
        state = {"j": 0}

        def call():
            for _ in range(K):
                env.step(acts[state["j"] % 8])
                state["j"] += 1
    call()                       # warm-up (allocates the output buffers)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(L):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (L * K)
    print(json.dumps({"leg": leg, "envs": n, "steps_per_launch": K, "launches": L, "substeps": nsub,
                      "us_per_env_step_batch": us, "env_steps_per_s": n / us * 1e6}))
    env.close()


if __name__ == "__main__":
    main()
