#!/usr/bin/env python3
"""GPU box: stress the END of a served session -- cs_serve_end raises the stop word right behind the last submitted
action row; the env kernel must still take that row.  Many short sessions (K steps each, all rows submitted at once,
closed without waiting), each followed by cs_serve_status: a session whose tiles did not all complete K steps is a
failure (round 5: one flaky run of tests/test_gpu_served.py::test_a_session_closed_without_waiting_is_drained_...
led here -- the env kernel looked at the stop word AFTER a stale look at the row and gave up on a row that had arrived
in between).

    python3 tools/serve_drain_repro.py [sessions=2000] [K=1] [envs=65536] [ring=2]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch            # noqa: E402
import gym_copter_amd   # noqa: E402

sessions = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
ring = int(sys.argv[4]) if len(sys.argv) > 4 else 2
env = gym_copter_amd.CopterVecEnv(task="lander3d", num_envs=n, seed=8, autoreset_mode="next_step")
env.reset()
g = torch.Generator(device=env.device)
g.manual_seed(4)
acts = torch.rand((K, n, 4), generator=g, device=env.device) * 2 - 1
torch.cuda.synchronize()
short = []
for s in range(sessions):
    env.serve_begin(K, ring=ring, timeout=5.0)
    for k in range(K):
        env.serve_submit(k, acts[k])
    env.serve_end(wait=False)
    st = env.serve_status()                       # synchronises
    if st != (K, K, 0):
        short.append((s, st))
        if len(short) <= 10:
            print("session %d: serve_status %r (every tile should have completed %d steps)" % (s, st, K), flush=True)
print("serve_drain_repro: %d of %d sessions (K=%d, %d envs, ring %d) ended short %s"
      % (len(short), sessions, K, n, ring, short[:5]))
env.close()
