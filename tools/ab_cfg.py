#!/usr/bin/env python3
"""GPU box: interleaved A/B of library builds over bench.py configurations, in ONE run (timings taken in
different runs / on different boxes are never compared).

  python tools/ab_cfg.py --libs base=gym_copter_amd/libcopterstep.so exp=gym_copter_amd/csrc/build/libexp.so \
                         --cfgs "65536 uniform 1" "65536 near_hover 10" "4194304 uniform 1" [--reps 3]

A configuration is "<envs> <action law> <substeps> [task]".  Every (library, configuration) pair is run `reps`
times, round-robin over the libraries, each as its own bench.py process (headline leg only, hipGraph replay);
prints per pair the per-step microseconds of every repetition and their median, and the ratio to the first
library."""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--libs", nargs="+", required=True)
    p.add_argument("--cfgs", nargs="+", default=["65536 uniform 1"])
    p.add_argument("--reps", type=int, default=3)
    p.add_argument("--steps", type=int, default=2000)
    a = p.parse_args()
    libs = [x.split("=", 1) for x in a.libs]
    res = {}
    for rep in range(a.reps):
        for cfg in a.cfgs:
            f = cfg.split()
            n, law, nsub = int(f[0]), f[1], int(f[2])
            task = f[3] if len(f) > 3 else "lander3d"
            ring = 64 if n <= 131072 else (16 if n <= 524288 else 4)
            steps = a.steps if n <= 1048576 else 200
            for name, path in libs:
                env = dict(os.environ, COPTERSTEP_LIB=os.path.join(ROOT, path))
                cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--envs", str(n), "--actions", law,
                       "--substeps", str(nsub), "--task", task, "--steps", str(steps), "--warmup", "100",
                       "--ring", str(ring), "--no-sweep", "--pid", "0", "--many", "0", "--served", "0",
                       "--no-cpu-baseline", "--no-span", "--full-out", "/dev/null", "--regions", "7"]
                out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
                try:
                    d = json.loads(out.stdout.strip().splitlines()[-1])
                    res.setdefault((cfg, name), []).append(d["ms_per_step"] * 1e3)
                except Exception:
                    print("FAILED", name, cfg, out.stderr[-400:], flush=True)
    for cfg in a.cfgs:
        base = None
        for name, _ in libs:
            v = res.get((cfg, name), [])
            if not v:
                continue
            med = statistics.median(v)
            base = base or med
            print("%-28s %-14s %s  median %.3f us  x%.4f" % (cfg, name, " ".join("%.3f" % x for x in v), med, med / base),
                  flush=True)


if __name__ == "__main__":
    main()
