#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs written by scripts/profile_gpu.sh into a text summary from which every roofline
figure of bench.py can be recomputed by arithmetic (profiles/rNN_summary.txt) and profiles-ready JSON."""
import collections
import csv
import glob
import json
import os
import re
import sys

import numpy as np

out = sys.argv[1]
res = {}
ALGO = 176          # algorithmic bytes per env-step, Lander3D and Hover3D
PEAK = 8e12


def find(sub, pat):
    f = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return f[0] if f else None


def short(name):
    name = re.sub(r"void cs::\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*$", "", name)[:96]


# the instantiations: step_kernel<task, mode, lean, stream actions, stream state, one call, output form (1 = packed rows)>
TASKS = {"0": "lander3d", "1": "hover3d"}

# ---- 1. kernel trace grouped by (kernel, grid): every sweep point has its own row --------------------------
for sub, label in (("trace", "bench default command (65 536 envs + sweep + config 5 + K-step extras)"),
                   ("trace_4m", "headline kernel alone, 4 194 304 envs"),
                   ("trace_rccl", "bench.py --gather on one GPU: 1-rank RCCL group, collectives forced")):
    f = find(sub, "*kernel_trace.csv")
    if not f:
        continue
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        groups[(r["Kernel_Name"], int(r["Grid_Size_X"]) if r.get("Grid_Size_X") else int(r.get("Grid_Size", 0)))].append(
            int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("== rocprofv3 --kernel-trace, dispatches grouped by (kernel, grid size): %s ==" % label)
    print("%-96s %10s %7s %10s %10s %10s   %s" % ("kernel", "grid", "calls", "mean_ns", "median_ns", "min_ns",
                                                  "176 B x envs / mean / 8 TB/s"))
    table = []
    for (name, grid), d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        d = np.array(d)
        m = re.search(r"step_kernel<(\d), (\d), (true|false), (true|false), (true|false), (true|false)(?:, \d)?>", name)
        frac = ""
        if m and len(d) >= 20:
            envs = grid          # one thread per env, whole tiles
            frac = "%.3f" % (ALGO * envs / (d.mean() * 1e-9) / PEAK)
            table.append({"kernel": short(name), "task": TASKS.get(m.group(1), m.group(1)), "one_call": m.group(6) == "true",
                          "grid": grid, "calls": int(len(d)), "mean_ns": float(d.mean()), "median_ns": float(np.median(d)),
                          "min_ns": float(d.min()), "frac_from_mean": float(frac)})
        if sub == "trace_rccl" and not ("nccl" in name.lower() or "rccl" in name.lower() or "step_kernel" in name):
            continue
        if len(d) >= 20 or "nccl" in name.lower():
            print("%-96s %10d %7d %10.0f %10.0f %10d   %s" % (short(name), grid, len(d), d.mean(), np.median(d), d.min(), frac))
    res[sub + "_by_kernel_and_grid"] = table
    if sub == "trace_rccl":
        names = sorted({short(k[0]) for k in groups if "nccl" in k[0].lower() or "rccl" in k[0].lower()})
        res["rccl_kernels_seen"] = names
        print("   RCCL kernels in the trace: %s" % (names or "NONE"))
    log = os.path.join(out, sub + ".log")
    if os.path.exists(log):
        txt = open(log).read()
        m = re.search(r'"launch_us": ([0-9.]+)', txt)
        if m:
            res[sub + "_hip_event_launch_us_in_profiled_run"] = float(m.group(1))
            print("   bench.py's own HIP-event launch time of the headline in this (profiled) run: %s us" % m.group(1))
        m = re.search(r'"allgather_launch_mode": (\{[^}]*\})', txt)
        if m:
            print("   all-gather legs captured as: %s" % m.group(1))
        m = re.search(r'"rccl": (\{[^}]*\})', txt)
        if m:
            print("   rccl echo: %s" % m.group(1))
    print()

f = os.path.join(out, "rccl_allgather_calls.txt")
if os.path.exists(f):
    lines = open(f).read().strip().splitlines()
    print("== RCCL's own call log of the forced 1-rank collectives (NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=COLL) ==")
    print("AllGather calls logged: %s" % (lines[0] if lines else "?"))
    for ln in lines[1:]:
        print("   " + ln)
    res["rccl_allgather_calls_logged"] = int(lines[0]) if lines and lines[0].isdigit() else None
    print()

# ---- 2. the kernel's own duration, un-profiled (span build) -------------------------------------------------
spans = sorted(glob.glob(os.path.join(out, "span_*.json")))
if spans:
    print("== kernel span per launch, un-profiled (tools/kernel_span.py: first wavefront start -> last wavefront end, 100 MHz clock) ==")
    for f in spans:
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception as e:
            print("   %s: unreadable (%r)" % (os.path.basename(f), e))
            continue
        k = d["kernel_span_ns"]
        key = "%s_%d%s" % (d["task"], d["envs"], "_substeps%d" % d["substeps"] if d["substeps"] > 1 else "")
        res["span_" + key] = d
        print("%-34s span median %7.0f ns (p10 %6.0f, p90 %6.0f, min %6.0f)  gap to next eager launch %6.0f ns  eager pace %.3f us/step"
              "  -> 176 B x envs / span = %.3f of 8 TB/s" % (key, k["median"], k["p10"], k["p90"], k["min"],
                                                        d["gap_between_eager_launches_ns"]["median"],
                                                        d["eager_pace_us_per_step_hip_events"], d["frac_of_8TBps_over_the_span"]))
    print()


def counters(sub, kernel):
    f = find(sub, "*counter_collection.csv")
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


# ---- 3. HBM traffic by PMC ------------------------------------------------------------------------------------
print("== PMC, HBM traffic per launch of step_kernel (means; FETCH_SIZE doubled: the gfx950 correction of MI355X_MICROARCH.md) ==")
for name, kern, n in (("lander3d_65536", "step_kernel<0, 0, true", 65536), ("lander3d_4194304", "step_kernel<0, 0, true", 4194304),
                      ("hover3d_262144", "step_kernel<1, 0, true", 262144),
                      ("lander3d_65536_substeps10", "step_kernel<0, 0, true", 65536)):
    fs = counters("fetch_" + name, kern).get("FETCH_SIZE")
    ws = counters("write_" + name, kern).get("WRITE_SIZE")
    if fs is None or ws is None:
        continue
    fetch_b, write_b = 2 * fs * 1024, ws * 1024      # KiB -> bytes; FETCH_SIZE reads half of a wide coalesced stream
    algo = ALGO * n
    print("%-28s FETCH_SIZE=%.1f KiB (x2: %.2f MB)  WRITE_SIZE=%.1f KiB (%.2f MB)  traffic=%.2f MB  algorithmic=%.2f MB  ratio=%.3f"
          % (name, fs, fetch_b / 1e6, ws, write_b / 1e6, (fetch_b + write_b) / 1e6, algo / 1e6, (fetch_b + write_b) / algo))
    res[name] = fetch_b + write_b
    res[name + "_detail"] = {"fetch_bytes_corrected": fetch_b, "write_bytes": write_b, "algorithmic_bytes": algo}
print()

# ---- 4. SQ counters -----------------------------------------------------------------------------------------------
for tag, kern, label in (("", "step_kernel<0, 0, true", "headline: Lander3D 65 536"),
                         ("_c5", "step_kernel<0, 0, true", "BASELINE configs[4]: Lander3D 65 536, 10 substeps, near hover"),
                         ("_c3", "step_kernel<1, 0, true", "BASELINE configs[2]: Hover3D 262 144")):
    c = {}
    for sub in ("sq1" + tag, "sq2" + tag) + (("l2",) if tag == "" else ()):
        c.update(counters(sub, kern))
    if not c:
        continue
    print("== PMC, SQ / TCC counters per launch (%s) ==" % label)
    for k, v in sorted(c.items()):
        print("%-24s %14.1f" % (k, v))
    if c.get("SQ_WAVES"):
        w = c["SQ_WAVES"]
        per = {k: c[k] / w for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
                                     "SQ_INSTS_LDS", "SQ_WAVE_CYCLES") if k in c}
        print("per wavefront: " + "  ".join("%s %.1f" % (k.replace("SQ_INSTS_", "").replace("SQ_", ""), v) for k, v in per.items()))
        if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
            print("SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.3f" % (c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]))
        res["pmc_per_wave" + tag] = per
    res.update({"pmc%s_%s" % (tag, k): v for k, v in c.items()})
    print()

# ---- 4b. float64 arithmetic executed, by PMC (what bench.py prices config 5's float64-ALU bound with) --------------
counts = {"valu_per_wavefront": {}, "f64_flops_per_env_step": {}, "valu_per_wavefront_step": {}}
for key, sub, kern in (("lander3d_65536_substeps10", "fl_c5", "step_kernel<0, 0, true"), ("lander3d_65536", "fl", "step_kernel<0, 0, true"),
                       ("lander3d_1048576_substeps10", "fl_c5_1m", "step_kernel<0, 0, true")):
    c = counters(sub, kern)
    if c.get("SQ_WAVES"):
        w = c["SQ_WAVES"]
        ops = {k: c.get("SQ_INSTS_VALU_%s_F64" % k, 0.0) / w for k in ("ADD", "MUL", "FMA", "TRANS")}
        flops = ops["ADD"] + ops["MUL"] + ops["TRANS"] + 2 * ops["FMA"]
        counts["f64_flops_per_env_step"][key] = flops
        counts["valu_per_wavefront"][key] = c["SQ_INSTS_VALU"] / w
        print("== PMC, float64 arithmetic executed per wavefront (= per env: one lane each), %s ==" % key)
        print("ADD_F64 %.1f  MUL_F64 %.1f  FMA_F64 %.1f  TRANS_F64 %.1f  -> %.1f flop per env-step (FMA = 2); all vector "
              "instructions %.1f" % (ops["ADD"], ops["MUL"], ops["FMA"], ops["TRANS"], flops, c["SQ_INSTS_VALU"] / w))
        print()
for key, tag in (("lander3d_65536", ""), ("lander3d_65536_substeps10", "_c5"), ("hover3d_262144", "_c3")):
    per = res.get("pmc_per_wave" + tag)
    if per and "SQ_INSTS_VALU" in per:
        counts["valu_per_wavefront"].setdefault(key, per["SQ_INSTS_VALU"])

# the K-step kernels of the same runs (bench.py extras): executed instructions per wavefront and env-step
print("== PMC, K-step kernels: per wavefront and env-step ==")
res["pmc_k_step"] = {}
for name, label, k, leg in (("step_many_kernel<0, 0, true, 0,", "open loop", 64, "step_many"),
                            # (round 5: under upstream's gains the lean <= 65 536-env launch is the instantiation with the PID terms
                            #  compiled in, POLICY = 4; other gain sets run POLICY = 1)
                            ("step_many_kernel<0, 0, true, 4,", "PID policy (upstream's terms compiled in)", 100, "rollout_pid"),
                            ("step_many_kernel<0, 0, true, 1,", "PID policy (generic)", 100, "rollout_pid"),
                            ("step_many_kernel<0, 0, true, 2,", "random policy", 100, "rollout_random"),
                            ("rollout_custom_kernel<0, 0, true,", "caller's linear policy", 100, "rollout_policy_linear")):
    m = {}
    for sub in ("sq1", "sq2"):
        m.update(counters(sub, name))
    if m.get("SQ_WAVES"):
        per = {c: m[c] / m["SQ_WAVES"] / k for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD",
                                                    "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES") if c in m}
        res["pmc_k_step"][name] = per
        counts["valu_per_wavefront_step"].setdefault(leg, per["SQ_INSTS_VALU"])
        print("%-42s (%3d steps/launch): " % (label, k) +
              "  ".join("%s %.1f" % (c.replace("SQ_INSTS_", "").replace("SQ_", ""), v) for c, v in per.items()))
# round 5: the K-step kernels at 4 194 304 envs (64 wavefronts per SIMD; LDS-transpose instantiation), tools/kstep_probe.py
for name, label, k, leg, subs in (("step_many_kernel<0, 0, true, 0,", "open loop, 4 M envs", 16, "step_many_4194304", ("sq1_many_4m", "sq2_many_4m")),
                                  ("step_many_kernel<0, 0, true, 1,", "PID policy, 4 M envs", 16, "rollout_pid_4194304", ("sq1_pid_4m", "sq2_pid_4m"))):
    m = {}
    for sub in subs:
        m.update(counters(sub, name))
    if m.get("SQ_WAVES"):
        per = {c: m[c] / m["SQ_WAVES"] / k for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD",
                                                    "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
                                                    "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU") if c in m}
        res["pmc_k_step"][name + " @4M"] = per
        counts["valu_per_wavefront_step"][leg] = per["SQ_INSTS_VALU"]
        print("%-22s (%3d steps/launch): " % (label, k) +
              "  ".join("%s %.1f" % (c.replace("SQ_INSTS_", "").replace("SQ_", ""), v) for c, v in per.items()))
c5m = counters("sq2_c5_1m", "step_kernel<0, 0, true")
if c5m:
    res["pmc_c5_1m"] = c5m
    if c5m.get("SQ_WAVE_CYCLES"):
        print("configs[4] at 1 048 576 envs: SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.3f, SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = %.3f, "
              "SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES = %.3f" % (c5m.get("SQ_WAIT_ANY", 0) / c5m["SQ_WAVE_CYCLES"],
                                                              c5m.get("SQ_WAIT_INST_ANY", 0) / c5m["SQ_WAVE_CYCLES"],
                                                              c5m.get("SQ_ACTIVE_INST_VALU", 0) / max(c5m.get("SQ_BUSY_CYCLES", 1), 1)))
res["pmc_counts"] = counts
print()
print("== issue floors from the counts above: wavefronts per SIMD x vector instructions x 4 cycles / 2.4 GHz (1 024 SIMDs) ==")
for leg, v in counts["valu_per_wavefront_step"].items():
    nn = int(leg.rsplit("_", 1)[1]) if leg.rsplit("_", 1)[-1].isdigit() else 65536
    per_simd = -(-(nn // 64) // 1024)
    print("%-24s %6.1f VALU per wavefront and env-step -> floor %.3f us per step at %d envs (%d wavefronts per SIMD), ceiling %.1f G env-steps/s"
          % (leg, v, per_simd * v * 4 / 2.4e9 * 1e6, nn, per_simd, nn / (per_simd * v * 4 / 2.4e9) / 1e9))
for key, v in counts["valu_per_wavefront"].items():
    n = int(key.split("_")[1])
    per_simd = -(-(n // 64) // 1024)
    print("%-28s %6.1f VALU per wavefront, %d wavefront(s) per SIMD -> floor %.3f us per launch" % (key, v, per_simd, per_simd * v * 4 / 2.4e9 * 1e6))
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
