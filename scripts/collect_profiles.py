#!/usr/bin/env python3
"""Copy the judged evidence of a profiling run from gpurun_out/ (scratch) into profiles/ (tracked).

  python scripts/collect_profiles.py <prof_tag> <round>     # e.g. prof_r02a r02

expects gpurun_out/<prof_tag>/ as written by scripts/profile_gpu.sh (incl. bench_unprofiled.json, the
bench line of the same build taken without a profiler)."""
import csv
import glob
import json
import os
import re
import shutil
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof, RND = sys.argv[1], sys.argv[2]
P = os.path.join(R, "gpurun_out", prof)


def one(pat):
    return glob.glob(P + pat)[0]


shutil.copy(one("/trace/*/*_kernel_stats.csv"), R + "/profiles/%s_kernel_stats_bench_default.csv" % RND)
shutil.copy(one("/trace_4m/*/*_kernel_stats.csv"), R + "/profiles/%s_kernel_stats_4m_envs.csv" % RND)
shutil.copy(P + "/summary.txt", R + "/profiles/%s_summary.txt" % RND)
shutil.copy(P + "/summary.json", R + "/profiles/%s_summary.json" % RND)
shutil.copy(P + "/bench_unprofiled.json", R + "/profiles/%s_bench_default.json" % RND)

rows = list(csv.DictReader(open(one("/trace/*/*_kernel_trace.csv"))))


def durations(name):
    rr = [r for r in rows if name in r["Kernel_Name"]]
    return rr, np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rr])


rr, d = durations("step_kernel<0, 0, true, true, false, false, true>")
line = ("step_kernel<Lander3D,F32G,LEAN,stream actions> dispatch durations from rocprofv3 --kernel-trace, "
        "65 536 envs (ns): n=%d mean=%.1f median=%.1f min=%d max=%d p10=%d p90=%d\n"
        % (len(d), d.mean(), np.median(d), d.min(), d.max(), np.percentile(d, 10), np.percentile(d, 90)))
m = rr[0]
line += "VGPR=%s SGPR=%s LDS=%s scratch=%s grid=%s wg=%s\n" % (
    m["VGPR_Count"], m["SGPR_Count"], m["LDS_Block_Size"], m["Scratch_Size"], m["Grid_Size_X"], m["Workgroup_Size_X"])
for name, desc in (("step_many_kernel<0, 0, true, 0,", "open loop, 64 steps per launch"),
                   ("step_many_kernel<0, 0, true, 1,", "PID policy, 100 steps per launch"),
                   ("step_many_kernel<0, 0, true, 2,", "random policy, 100 steps per launch")):
    rr, dd = durations(name)
    if len(dd):
        line += "%s (%s): n=%d mean=%.0f ns  VGPR=%s SGPR=%s\n" % (name, desc, len(dd), dd.mean(),
                                                                  rr[0]["VGPR_Count"], rr[0]["SGPR_Count"])
open(R + "/profiles/%s_step_kernel_durations.txt" % RND, "w").write(line)
print(line)

txt = open(P + "/summary.txt").read()
t = json.load(open(R + "/profiles/traffic.json"))
for n, f, w in re.findall(r"N=\s*(\d+)\s+FETCH_SIZE=([\d.]+) KiB.*WRITE_SIZE=([\d.]+) KiB", txt):
    t["lander3d_%s" % n] = (2 * float(f) + float(w)) * 1024
json.dump(t, open(R + "/profiles/traffic.json", "w"), indent=1)
for ln in txt.splitlines():
    if "step_kernel" in ln or "HIP-event" in ln or "FETCH" in ln or "step_many" in ln:
        print(ln[:48], "...", ln[-74:])
b = json.load(open(P + "/bench_unprofiled.json"))
print("bench default: %.2f G env-steps/s, %.3f us/step, frac %.3f" % (b["value"] / 1e9, b["ms_per_step"] * 1e3,
                                                                    b["roofline"]["frac"]),
      [(k, round(b[k]["value"] / 1e9, 2), round(b[k]["us_per_step"], 3)) for k in ("step_many", "rollout_pid", "rollout_random")])
