#!/usr/bin/env python3
"""Diagnostic (GPU box): served stepping at a given size, eager and with graph-captured feeders, with short
timeouts and the session status printed after every phase.  python tools/serve_probe.py [envs] [K] [ring]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import gym_copter_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ring = int(sys.argv[3]) if len(sys.argv) > 3 else 2
env = gym_copter_amd.CopterVecEnv("lander3d", n, seed=1, autoreset_mode="next_step")
env.configure_pid()
env.reset()
acts = torch.rand((8, n, 4), device=env.device) * 2 - 1
print("max envs", env.serve_max_envs(), flush=True)
# many streams: HIP multiplexes them onto a few hardware queues, and a feeder stream that landed on the env
# kernel's queue would deadlock the session (until its timeouts)
streams = [torch.cuda.Stream(device=env.device) for _ in range(24)]
for j, st in enumerate(streams):
    with torch.cuda.stream(st):
        env.reset()
        env.serve_begin(8, ring=2, timeout=0.2)
        for s in range(8):
            env.serve_policy_pid(s)
        env.serve_end(wait=False)
        st.synchronize()
        print("feeder stream %2d: status %r" % (j, env.serve_status()), flush=True)


def run(name, body, graph=None):
    env.reset()
    cur = torch.cuda.current_stream(env.device)
    t0 = time.perf_counter()
    env.serve_begin(K, ring=ring, timeout=0.2)
    if graph is not None:
        graph.replay()
    else:
        for s in range(K):
            body(s)
    env.serve_end(wait=False)
    cur.synchronize()
    dt = time.perf_counter() - t0
    print("%-34s %8.1f us/step  status %r" % (name, dt / K * 1e6, env.serve_status()), flush=True)


for name, body in (("submit+collect", lambda s: (env.serve_submit(s, acts[s % 8]), env.serve_collect(s))),
                   ("policy_pid", lambda s: env.serve_policy_pid(s)),
                   ("submit only", lambda s: env.serve_submit(s, acts[s % 8]))):
    run(name + " eager", body)
    run(name + " eager again", body)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for s in range(K):
            body(s)
    run(name + " graph", body, g)
    run(name + " graph again", body, g)
    t0 = time.perf_counter()
    for _ in range(20):
        env.serve_begin(K, ring=ring, timeout=0.2)
        g.replay()
        env.serve_end(wait=False)
    torch.cuda.current_stream(env.device).synchronize()
    print("%-34s %8.3f us/step  status %r" % (name + " 20 sessions", (time.perf_counter() - t0) / (20 * K) * 1e6,
                                              env.serve_status()), flush=True)

# ---- what in bench.py's flow makes graph-replayed feeders slow?  one experiment per line ----------------------
body = lambda s: env.serve_policy_pid(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    for s in range(K):
        body(s)


def sessions(label, count=10, timeout=0.2, events=False, sync_each=False):
    env.reset()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if events:
        e0.record()
    for _ in range(count):
        env.serve_begin(K, ring=ring, timeout=timeout)
        g.replay()
        env.serve_end(wait=False)
        if sync_each:
            torch.cuda.current_stream(env.device).synchronize()
    if events:
        e1.record()
    torch.cuda.current_stream(env.device).synchronize()
    print("%-44s %8.3f us/step  status %r" % (label, (time.perf_counter() - t0) / (count * K) * 1e6, env.serve_status()),
          flush=True)


sessions("A plain")
sessions("B timing events around")
sessions("B timing events around", events=True)
sessions("C timeout 1.0", timeout=1.0)
other = torch.zeros(65536, device=env.device)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    for _ in range(100):
        other.add_(1.0)
for _ in range(50):
    g2.replay()
torch.cuda.synchronize()
sessions("D after another torch graph was replayed")
side = torch.cuda.Stream(device=env.device)
with torch.cuda.stream(side):
    sessions("E on a side stream")
sessions("F again plain")

# G: as bench.py does it: the SAME env first stepped from a captured hipGraph of cs_step launches
acts64 = torch.rand((64, n, 4), device=env.device) * 2 - 1
s2 = torch.cuda.Stream(device=env.device)
s2.wait_stream(torch.cuda.current_stream(env.device))
with torch.cuda.stream(s2):
    for j in range(3):
        env.step(acts64[j])
torch.cuda.current_stream(env.device).wait_stream(s2)
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3):
    for j in range(100):
        env.step(acts64[j % 64])
for _ in range(30):
    g3.replay()
torch.cuda.synchronize()
sessions("G after a cs_step graph on the same env")
g4 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g4, capture_error_mode="thread_local"):
    for s in range(K):
        body(s)
g = g4
sessions("H feeders re-captured after that")
