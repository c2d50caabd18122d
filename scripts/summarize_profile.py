#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs written by scripts/profile_gpu.sh into a text summary and
profiles-ready JSON (per-launch averages for the step kernel)."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
res = {}


def find(sub, pat):
    f = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return f[0] if f else None


import re
# the instantiations of the lean Lander3D kernel: <task, mode, lean, stream actions, stream state, prefetch, one call>
HEADLINE = {"trace": "step_kernel<0, 0, true, true, false, false, true>",        # 65 536 envs
            "trace_4m": "step_kernel<0, 0, true, false, true, false, true>"}     # 4 194 304 envs (streamed state)
for sub, label in (("trace", "bench default command, 65 536 envs"), ("trace_4m", "4 194 304 envs")):
    f = find(sub, "*kernel_stats.csv")
    if not f:
        continue
    print("== rocprofv3 --kernel-trace --stats (%s) ==" % label)
    for r in csv.DictReader(open(f)):
        print("%-110s calls=%6s avg_ns=%10s min=%8s max=%8s pct=%s" % (
            r["Name"][:110], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]))
        if HEADLINE[sub] in r["Name"]:
            res[sub + "_step_kernel_avg_ns"] = float(r["AverageNs"])
            res[sub + "_step_kernel_calls"] = int(r["Calls"])
    log = os.path.join(out, sub + ".log")
    if os.path.exists(log):
        m = re.search(r'"launch_us": ([0-9.]+)', open(log).read())
        if m:
            res[sub + "_hip_event_launch_us_in_profiled_run"] = float(m.group(1))
            print("   bench.py's own HIP-event launch time in this (profiled) run: %s us" % m.group(1))


def counters(sub, kernel="step_kernel<0, 0, true"):
    f = find(sub, "*counter_collection.csv")
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


print("\n== PMC, per launch of step_kernel (means) ==")
for n in (65536, 4194304):
    fs = counters("fetch_%d" % n).get("FETCH_SIZE")
    ws = counters("write_%d" % n).get("WRITE_SIZE")
    if fs is None or ws is None:
        continue
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
    # exactly half of the bytes of a coalesced streaming read -> double it; WRITE_SIZE is exact.
    fetch_b, write_b = 2 * fs * 1024, ws * 1024
    algo = 176 * n
    print("N=%8d  FETCH_SIZE=%.1f KiB (x2 corrected: %.2f MB)  WRITE_SIZE=%.1f KiB (%.2f MB)  "
          "traffic=%.2f MB  algorithmic=%.2f MB  ratio=%.3f" % (
              n, fs, fetch_b / 1e6, ws, write_b / 1e6, (fetch_b + write_b) / 1e6, algo / 1e6,
              (fetch_b + write_b) / algo))
    res["lander3d_%d" % n] = fetch_b + write_b
    res["lander3d_%d_detail" % n] = {"fetch_bytes_corrected": fetch_b, "write_bytes": write_b,
                                      "algorithmic_bytes": algo}
for sub in ("sq1", "sq2", "l2"):
    c = counters(sub)
    for k, v in sorted(c.items()):
        print("%-24s %14.1f" % (k, v))
    res.update({"pmc_" + k: v for k, v in c.items()})

# the K-step kernels of the same runs (bench.py extras): executed instructions per wavefront and env-step
print("\n== PMC, K-step kernels: per wavefront and env-step ==")
res["pmc_k_step"] = {}
for name, label, k in (("step_many_kernel<0, 0, true, 0,", "open loop", 64),
                       ("step_many_kernel<0, 0, true, 1,", "PID policy", 100),
                       ("step_many_kernel<0, 0, true, 2,", "random policy", 100)):
    m = {}
    for sub in ("sq1", "sq2"):
        m.update(counters(sub, name))
    if m.get("SQ_WAVES"):
        per = {c: m[c] / m["SQ_WAVES"] / k for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD",
                                                    "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES") if c in m}
        res["pmc_k_step"][name] = per
        print("%-14s (%3d steps/launch): " % (label, k) +
              "  ".join("%s %.1f" % (c.replace("SQ_INSTS_", "").replace("SQ_", ""), v) for c, v in per.items()))
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
