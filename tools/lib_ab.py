#!/usr/bin/env python3
"""GPU box: interleaved A/B of library builds over the WHOLE default bench line's kernel legs (headline, config 5, the
K-step legs) -- for build-flag and source experiments.   python tools/lib_ab.py name=path.so ... [--reps 3] [--full]
(--full: bench.py --full, i.e. cs_rollout_random and the caller-compiled policy as well)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if "=" in a]
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 3
libs = [a.split("=", 1) for a in args]
res = {}
for rep in range(reps):
    for name, path in libs:
        env = dict(os.environ, COPTERSTEP_LIB=os.path.join(ROOT, path))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--served", "0", "--steps", "1000",
                              "--warmup", "100", "--no-span", "--full-out", "/dev/null"] + (["--full"] if "--full" in sys.argv else []),
                             env=env, capture_output=True, text=True)
        line = json.loads(out.stdout.strip().splitlines()[-1])
        s = line["summary"]
        r = dict(s["k_step_us"], headline=line["roofline"]["launch_us"], config5=s["config5_launch_us"],
                 lander4m=176 * 4194304 / 8e12 / s["sweep_frac"]["lander3d_4194304_uniform"] * 1e6,
                 hover262k=176 * 262144 / 8e12 / s["sweep_frac"]["hover3d_262144_uniform"] * 1e6)
        for k, v in r.items():
            res.setdefault(name, {}).setdefault(k, []).append(round(v, 3))
for name, r in res.items():
    print(name, {k: v for k, v in r.items()})
