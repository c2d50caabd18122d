#!/usr/bin/env python3
"""tools/kstep_outputs_probe.py -- what bounds a K-step kernel at 65 536 envs: instruction issue or its output stream?

    python3 tools/kstep_outputs_probe.py [envs=65536] [launches=20]

cs_step_many / cs_rollout_random / cs_rollout_pid through the C ABI with output pointers withheld one group at a time
(NULL is legal for every output), and with K = 16 (output buffers of 48 MB: they stay in the 256 MiB Infinity Cache)
against K = 100 (300 MB: streamed to HBM).  Interleaved, three passes; HIP-event time per env step.  If a kernel is
bound by instruction issue, withholding an output saves only that output's store instructions (a few per cent) and K
does not matter; if it is bound by its write stream, withholding the observation rows saves much more and K = 16 is
faster than K = 100.  -> profiles/r06_kstep_outputs_probe.txt"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    import torch
    import gym_copter_amd as gca
    from gym_copter_amd import _lib
    env = gca.CopterVecEnv(task="lander3d", num_envs=n, device=0, seed=1234, autoreset_mode="next_step")
    env.configure_pid()
    lib, ctx = env._lib, env._ctx
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    KMAX = 100
    acts = torch.rand((KMAX, n, 4), generator=g, device="cuda") * 2 - 1
    obs = torch.empty((KMAX, n, env.obs_dim), device="cuda")
    rew = torch.empty((KMAX, n), device="cuda")
    flags = torch.empty((KMAX, n, 2), dtype=torch.uint8, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    null = C.c_void_p(0)
    forms = {"all outputs": (p(obs), p(rew), p(flags[:, :, 0]), p(flags[:, :, 1])),
             "no obs rows": (null, p(rew), p(flags[:, :, 0]), p(flags[:, :, 1])),
             "obs rows only": (p(obs), null, null, null),
             "no outputs": (null, null, null, null)}

    def call(leg, K, out):
        s = env._stream()
        if leg == "step_many":
            _lib.check(lib.cs_step_many(ctx, K, p(acts), *out, s))
        elif leg == "rollout_random":
            _lib.check(lib.cs_rollout_random(ctx, K, null, *out, s))
        else:
            _lib.check(lib.cs_rollout_pid(ctx, K, null, *out, s))

    def time_one(leg, K, out):
        reps = max(1, L * 100 // K)
        call(leg, K, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call(leg, K, out)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (reps * K)

    res = {}
    for _ in range(3):
        for leg in ("step_many", "rollout_random", "rollout_pid"):
            env.reset()
            for K in (100, 16):
                for name, out in forms.items():
                    res.setdefault((leg, K, name), []).append(round(time_one(leg, K, out), 4))
    print("# %d envs, us per env step (batch), three interleaved passes; K = steps per launch" % n)
    for (leg, K, name), v in res.items():
        print("%-15s K=%-4d %-14s %s" % (leg, K, name, v))
    print(json.dumps({"%s|%d|%s" % k: sorted(v)[1] for k, v in res.items()}))
    env.close()


if __name__ == "__main__":
    main()
