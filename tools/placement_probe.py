#!/usr/bin/env python3
"""GPU box: does the 4 M-env (HBM-resident) step time depend on where the buffers happen to land?
  python tools/placement_probe.py [envs] [pad_mb]
One process = one placement: prints the device addresses of the state tiles, the action ring and the packed output rows
(modulo a few powers of two) and the step time (hipGraph replay, HIP events).  Run it several times (different
processes get different placements) and compare.  pad_mb: allocate and keep that many MB first (shifts everything)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import gym_copter_amd  # noqa: E402
from gym_copter_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
pad = int(sys.argv[2]) if len(sys.argv) > 2 else 0
keep = torch.empty(pad << 20, dtype=torch.uint8, device="cuda") if pad else None
env = gym_copter_amd.CopterVecEnv("lander3d", N, seed=1234, autoreset_mode="next_step")
env.reset()
v = _lib.LaunchView()
v.struct_size = C.sizeof(_lib.LaunchView)
_lib.check(env._lib.cs_get_launch_view(env._ctx, C.byref(v)))
tiles = C.cast(v.state, C.POINTER(C.c_uint64))[0]
ring = 4
acts = torch.rand((ring, N, 4), device=env.device) * 2 - 1
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for j in range(3):
        env.step(acts[j % ring])
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for j in range(20):
        env.step(acts[j % ring])
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 100 * 1e3)
fmt = lambda a: "0x%012x (mod 2M %7d K, mod 1G %4d M)" % (a, (a % (2 << 20)) >> 10, (a % (1 << 30)) >> 20)
print("envs %d pad %d MB: %.2f us/step (min %.2f)  tiles %s  actions %s  rows %s" %
      (N, pad, sorted(ts)[2], min(ts), fmt(tiles), fmt(acts.data_ptr()), fmt(env._rows.data_ptr())), flush=True)
