#!/usr/bin/env python3
"""tools/kstep_stamps.py -- where a wavefront of a K-step kernel spends the cycles of ONE loop iteration.

rocprofv3's thread trace (--att) cannot be decoded in this image (profiles/r05_att_unavailable.txt), so the
attribution VERDICT round 4 (#4) asked for comes from in-kernel stamps: the diagnostic build
`make -C gym_copter_amd/csrc kstamps` (-DCS_KSTAMPS) reads s_memtime at the phase boundaries of two consecutive loop
iterations in the middle of a launch (dev_tile.h: CS_KSTAMP; slots in copterstep_kernels.hip / dev_task.h) and this
tool prints, per leg, the median over all wavefronts of the cycles between consecutive stamps, next to the cycles the
phase's vector instructions need at 4 cycles apiece when the counts are given (from the ISA listing).

    python3 tools/kstep_stamps.py [envs=65536] [K=8]

The stamps pin the phases in program order (a scheduling barrier either side) and cost ~40 cycles each: read shares
and differences between legs, not the total -- the un-instrumented per-step time is printed beside it (product library,
tools/kstep_probe.py in a child process).
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gym_copter_amd", "csrc", "build", "libcopterstep_kstamps.so")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8

PHASES = [(0, 1, "policy (PID heuristic / Philox draw / next action row requested)"),
          (1, 2, "clip + motor model + pending perturbation"),
          (2, 3, "Dynamics.setMotors: trig, rotation, status machine, Euler"),
          (3, 4, "round to the stored words + float32 observation row"),
          (4, 5, "shaping potential, reward, termination"),
          (5, 6, "masked auto-reset"),
          (6, 7, "controller hand-over / action row taken delivery of"),
          (7, 8, "reward + flag stores issued"),
          (8, 9, "observation row stores issued"),
          (14, 15, "(two stamps back to back: the cost of a stamp)")]


def unprofiled():
    """us per env step of the same legs on the PRODUCT library (children: this process loads the stamp build)."""
    out = {}
    for leg in ("many", "pid", "random"):
        try:
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kstep_probe.py"), leg, str(N), "100", "20"],
                               capture_output=True, text=True, timeout=300)
            out[leg] = json.loads(p.stdout.strip().splitlines()[-1])["us_per_env_step_batch"]
        except Exception as e:
            out[leg] = repr(e)
    return out


def main():
    base = unprofiled()                      # before this process touches the GPU
    os.environ["COPTERSTEP_LIB"] = LIB
    sys.path.insert(0, ROOT)
    import torch
    import gym_copter_amd as gca
    from gym_copter_amd import _lib
    lib = _lib.load()
    lib.cs_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.cs_debug_stamp_slots.restype = C.c_int
    slots = lib.cs_debug_stamp_slots()
    assert slots == 32, "not the kstamps build"
    nt = (N + 255) // 256 * 4
    buf = np.zeros((nt, slots), dtype=np.uint64)
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    print("# K-step phase stamps: %d envs (one wavefront per SIMD at 65 536), %d steps per launch, iterations %d and %d "
          "stamped; shader-clock cycles, median over %d wavefronts and 10 launches" % (N, K, K // 2, K // 2 + 1, N // 64))
    print("# un-instrumented product library, us per env step (tools/kstep_probe.py, eager launches of 100 steps): %s"
          % json.dumps(base))
    for leg, name in (("many", "cs_step_many (open loop)"), ("pid", "cs_rollout_pid (PID heuristic)"),
                      ("random", "cs_rollout_random (Philox policy)")):
        env = gca.CopterVecEnv(task="lander3d", num_envs=N, device=0, seed=1234, autoreset_mode="next_step")
        env.reset()
        if leg == "many":
            acts = torch.rand((K, N, 4), generator=g, device="cuda") * 2 - 1
            call = lambda: env.step_many(acts)
        elif leg == "pid":
            env.configure_pid()
            env.reset()
            call = lambda: env.rollout_pid(K)
        else:
            call = lambda: env.rollout_random(K)
        for _ in range(5):
            call()
        res = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for rep in range(10):
            call()
            torch.cuda.synchronize()
            lib.cs_debug_read_stamps(env._ctx, buf.ctypes.data_as(C.c_void_p), None)
            res.append(buf[:max(1, N // 64)].astype(np.int64).copy())
        e1.record()
        torch.cuda.synchronize()
        b = np.stack(res)                         # [launch, wavefront, slot]
        print("\n== %s ==" % name)
        tot = {}
        for it, off in ((K // 2, 0), (K // 2 + 1, 16)):
            whole = np.median(b[:, :, off + 15] - b[:, :, off + 0])
            tot[it] = whole
            print("iteration %d: loop top -> end of iteration %7.0f cycles" % (it, whole))
            for a, z, label in PHASES:
                d = np.median(b[:, :, off + z] - b[:, :, off + a])
                print("   %-72s %7.0f  (%4.1f %%)" % (label, d, 100.0 * d / whole))
        # the next iteration's loop top minus this one's = a full stamped iteration incl. the loop branch
        # the clock the kernel itself runs at: shader-clock ticks per 100 MHz tick between the two loop tops
        dt, dr = (b[:, :, 16] - b[:, :, 0]).astype(np.float64), (b[:, :, 28] - b[:, :, 12]).astype(np.float64)
        ok = dr > 0
        print("in-kernel clock over the stamped iteration: %.3f GHz (sum of s_memtime deltas / sum of s_memrealtime deltas x 100 MHz, "
              "%d wavefront-launches; the stamps slow the iteration, not the clock)" % (dt[ok].sum() / dr[ok].sum() * 0.1, int(ok.sum())))
        full = np.median(b[:, :, 16] - b[:, :, 0])
        print("loop top of iteration %d -> loop top of iteration %d: %.0f cycles (stamped build; 10 stamps of ~%.0f cycles in it)"
              % (K // 2, K // 2 + 1, full, np.median(b[:, :, 15] - b[:, :, 14])))
        env.close()


if __name__ == "__main__":
    main()
