#!/usr/bin/env python3
"""GPU box: does the product's double-buffered half-batch schedule hide the dependent-launch gap?  The 65 536 envs of
the headline as ONE context stepped on one stream (bench.py's headline) against H contexts of 65 536 / H envs, each a
chain of one-launch steps on its OWN stream (hipGraphs of 100 launches, replayed concurrently) -- what
gym_copter_amd.sharded.HalfBatchPipeline gives a learner that works on one half while the other half steps.
  python3 tools/half_batches_probe.py [total_envs=65536] [replays=200]
Prints one JSON line: us per step of the whole batch for H = 1, 2, 4."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import gym_copter_amd as gca  # noqa: E402

total = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
replays = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
out = {"total_envs": total, "chunk": 100, "replays": replays, "us_per_step": {}}
for H in (1, 2, 4, 1, 2, 4):
    n = total // H
    envs, steppers, streams = [], [], []
    for h in range(H):
        env = gca.CopterVecEnv(task="lander3d", num_envs=n, device=0, seed=1234, autoreset_mode="next_step",
                               env_id_base=h * n)
        env.reset()
        acts = bench.make_actions(torch, "uniform", 64, n, dev, 1234 + h)
        envs.append(env)
        steppers.append(bench.Stepper(torch, env, acts, True, 100))
        streams.append(torch.cuda.Stream(device=dev))
    torch.cuda.synchronize()

    def region(r):
        for _ in range(r):
            for h in range(H):
                with torch.cuda.stream(streams[h]):
                    steppers[h].graph.replay()
    region(20)
    torch.cuda.synchronize()
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        region(replays)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out["us_per_step"].setdefault(str(H), []).append(round(best / (replays * 100) * 1e6, 4))
    for e in envs:
        e.close()
    del steppers, envs
    torch.cuda.empty_cache()
print(json.dumps(out))
