#!/usr/bin/env python3
"""Diagnostic (GPU box): the step kernel's OWN duration per launch, without the launch gap and without a
profiler's per-dispatch cost.

  make -C gym_copter_amd/csrc span            # -> gym_copter_amd/csrc/build/libcopterstep_span.so
  python tools/kernel_span.py [task] [num_envs] [uniform|near_hover|const] [substeps]

The span build (-DCS_SPAN) has every wavefront note the chip-wide 100 MHz clock (s_memrealtime) when it
starts and, after its stores have been acknowledged, when it ends, into its own slot of its launch (two plain
8-byte stores by lane 0; the host takes min / max per launch).  Unlike the stamp build it does NOT serialise the
kernel's phases; what it adds is two scalar clock reads, the final s_waitcnt and one 16-byte store per wavefront.  span = latest end - earliest start = the time the
kernel occupies the chip.  Launches are eager (one slot per launch); the same process also times the product
path's pace with HIP events so that span + gap can be reconciled with bench.py's per-step figure.  Never
the product."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["COPTERSTEP_LIB"] = os.path.join(ROOT, "gym_copter_amd", "csrc", "build", "libcopterstep_span.so")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import gym_copter_amd  # noqa: E402
from gym_copter_amd import _lib  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "lander3d"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
law = sys.argv[3] if len(sys.argv) > 3 else "uniform"
nsub = int(sys.argv[4]) if len(sys.argv) > 4 else 1
HOVER = 0.016560178185018043
env = gym_copter_amd.CopterVecEnv(task, N, seed=1234, autoreset_mode="next_step", substeps=nsub)
env.reset()
dev = env.device
g = torch.Generator(device=dev)
g.manual_seed(1234)
ring = 16
acts = (torch.rand((ring, N, 4), generator=g, device=dev) * 2 - 1 if law == "uniform"
        else torch.full((ring, N, 4), 1.625e-2, device=dev) if law == "const"
        else HOVER * (1 + 0.01 * torch.randn((ring, N, 4), generator=g, device=dev)))
lib = _lib.load()
lib.cs_debug_read_spans.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
lib.cs_debug_reset_spans.argtypes = [C.c_void_p]
for j in range(300):
    env.step(acts[j % ring])
torch.cuda.synchronize()
L = 200                                  # launches per batch (the span buffer holds 256)
nt = (N + 255) // 256 * 4                # allocated tiles (DevState::ntiles)
tiles = (N + 63) // 64                   # tiles that a launch runs
spans, gaps = [], []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
pace = []
for batch in range(5):
    lib.cs_debug_reset_spans(env._ctx)
    torch.cuda.synchronize()
    e0.record()
    for j in range(L):
        env.step(acts[j % ring])
    e1.record()
    torch.cuda.synchronize()
    pace.append(e0.elapsed_time(e1) * 1e3 / L)
    buf = np.zeros((L, nt, 2), dtype=np.uint64)
    lib.cs_debug_read_spans(env._ctx, buf.ctypes.data_as(C.c_void_p), L, None)
    b = buf[:, :tiles].astype(np.int64)
    start, end = b[:, :, 0].min(axis=1), b[:, :, 1].max(axis=1)
    spans.append((end - start) * 10.0)                       # ns
    gaps.append((start[1:] - end[:-1]) * 10.0)
span, gap = np.concatenate(spans), np.concatenate(gaps)
ok = np.ones(len(span), bool)
out = {"task": task, "envs": N, "actions": law, "substeps": nsub, "launches": int(len(span)),
       "kernel_span_ns": {"median": float(np.median(span)), "mean": float(span.mean()), "p10": float(np.percentile(span, 10)),
                          "p90": float(np.percentile(span, 90)), "min": float(span.min())},
       "gap_between_eager_launches_ns": {"median": float(np.median(gap)), "p10": float(np.percentile(gap, 10))},
       "eager_pace_us_per_step_hip_events": float(np.median(pace)),
       "clock": "s_memrealtime, 100 MHz (10 ns per tick)",
       "algorithmic_bytes": 176 * N, "frac_of_8TBps_over_the_span": 176 * N / (np.median(span) * 1e-9) / 8e12}
print(json.dumps(out))
