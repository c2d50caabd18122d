#!/usr/bin/env python3
"""Diagnostic (GPU box): per-wavefront phase timing of step_kernel from the stamp build.

  make -C gym_copter_amd/csrc stamps          # -> gym_copter_amd/csrc/build/libcopterstep_stamps.so
  python tools/stamps.py [num_envs] [uniform|near_hover] [substeps]
  STAMPS_THRASH_MB=512 python tools/stamps.py ...   # a 512 MB read-modify-write before every step (cold caches)

The stamp build (-DCS_STAMPS) records s_memtime at phase boundaries of every wavefront into a
side buffer; it serialises the phases, so read the SHARES, not the total.  Never the product."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["COPTERSTEP_LIB"] = os.path.join(ROOT, "gym_copter_amd", "csrc", "build",
                                            "libcopterstep_stamps.so")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import gym_copter_amd  # noqa: E402
from gym_copter_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
law = sys.argv[2] if len(sys.argv) > 2 else "uniform"
nsub = int(sys.argv[3]) if len(sys.argv) > 3 else 1
env = gym_copter_amd.CopterVecEnv("lander3d", N, seed=1, autoreset_mode="next_step", substeps=nsub)
env.reset()
dev = env.device
acts = [torch.rand((N, 4), device=dev) * 2 - 1 if law == "uniform"
        else 0.01656 * (1 + 0.01 * torch.randn((N, 4), device=dev)) for _ in range(8)]
for j in range(30):
    env.step(acts[j % 8])
torch.cuda.synchronize()
lib = _lib.load()
nt = (N + 255) // 256 * 4
buf = np.zeros((nt, 8), dtype=np.uint64)
lib.cs_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
res = []
thrash = torch.empty(int(os.environ.get("STAMPS_THRASH_MB", "0")) << 20, dtype=torch.uint8, device=dev)
for rep in range(10):
    if thrash.numel():
        thrash.add_(1)          # STAMPS_THRASH_MB=512: push everything out of the L2s and the Infinity Cache first
    env.step(acts[rep % 8])
    torch.cuda.synchronize()
    lib.cs_debug_read_stamps(env._ctx, buf.ctypes.data_as(C.c_void_p), None)
    b = buf[:max(1, N // 64)].astype(np.int64)
    # slots in use: 0 kernel entry, 1 loads landed, 5 step body done, 6 stores issued, 7 stores acknowledged
    res.append(np.stack([b[:, 1] - b[:, 0], b[:, 5] - b[:, 1], b[:, 6] - b[:, 5], b[:, 7] - b[:, 6],
                         b[:, 7] - b[:, 0]], axis=1))
d = np.median(np.stack(res), axis=0)
names = ["loads issued -> landed", "decode + step body", "stores issued (+ LDS transpose)", "stores acknowledged",
         "whole wavefront"]
print("N", N, law, "substeps", nsub, "- shader-clock cycles per phase, median over wavefronts and 10 steps")
for k in range(5):
    print("%-32s %8.0f" % (names[k], np.median(d[:, k])))
