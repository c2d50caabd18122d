#!/usr/bin/env python3
"""Copy the judged evidence of a profiling run from gpurun_out/ (scratch) into profiles/ (tracked).

  python scripts/collect_profiles.py <prof_tag> <round>     # e.g. prof_r03a r03

expects gpurun_out/<prof_tag>/ as written by scripts/profile_gpu.sh (incl. bench_unprofiled.json, the bench
line of the same build taken without a profiler).  profiles/traffic.json is rewritten with the PMC traffic
figures AND the stamp bench.py checks before it reports them: the commit and a hash of the kernel sources the
figures were measured on (bench.kernel_source_hash)."""
import glob
import json
import os
import shutil
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
prof, RND = sys.argv[1], sys.argv[2]
P = os.path.join(R, "gpurun_out", prof)


def one(pat):
    f = glob.glob(P + pat)
    return f[0] if f else None


for src, dst in (("/trace/*/*_kernel_stats.csv", "kernel_stats_bench_default.csv"),
                 ("/trace_4m/*/*_kernel_stats.csv", "kernel_stats_4m_envs.csv"),
                 ("/trace_rccl/*/*_kernel_stats.csv", "kernel_stats_rccl_one_gpu.csv")):
    f = one(src)
    if f:
        shutil.copy(f, R + "/profiles/%s_%s" % (RND, dst))
for name in ("summary.txt", "summary.json"):
    shutil.copy(P + "/" + name, R + "/profiles/%s_%s" % (RND, name))
shutil.copy(P + "/bench_unprofiled.json", R + "/profiles/%s_bench_default.json" % RND)
if os.path.exists(P + "/bench_full_unprofiled.json"):       # (round 6 on: the line is compact, the full record beside it)
    shutil.copy(P + "/bench_full_unprofiled.json", R + "/profiles/%s_bench_full.json" % RND)
for f in sorted(glob.glob(P + "/span_*.json")):
    shutil.copy(f, R + "/profiles/%s_%s" % (RND, os.path.basename(f)))

# registers / occupancy of the default instantiations, appended to the summary so that the next regression is visible
# (VERDICT round 5 #4): from the ISA listing of THIS tree (make -C gym_copter_amd/csrc asm)
DEFAULTS = (("step_kernelILi0ELi0ELb1ELb1ELb0ELb1ELi1E", "step_kernel<lander3d, float32 words, lean, stream actions, one call, packed rows>  headline, <= 98 304 envs"),
            ("step_kernelILi0ELi0ELb1ELb0ELb0ELb1ELi0E", "step_kernel<lander3d, ..., no stream hints, one call, output form at run time>   98 304 < envs < 3.5 M, plain arrays"),
            ("step_kernelILi0ELi0ELb1ELb0ELb1ELb1ELi0E", "step_kernel<lander3d, ..., stream state, one call, run-time form>                >= 3.5 M envs (the 4 M point)"),
            ("step_kernelILi1ELi0ELb1ELb0ELb0ELb1ELi0E", "step_kernel<hover3d, ..., no stream hints, one call, run-time form>              BASELINE configs[2]"),
            ("step_kernelILi0ELi0ELb1ELb1ELb0ELb0ELi1E", "step_kernel<lander3d, ..., stream actions, substep loop, packed rows>            BASELINE configs[4]"),
            ("step_many_kernelILi0ELi0ELb1ELi0ELb1ELi2E", "step_many_kernel<lander3d, ..., open loop, per-lane rows + unconditional outputs>   cs_step_many, <= 65 536 envs"),
            ("step_many_kernelILi0ELi0ELb1ELi4ELb1ELi2E", "step_many_kernel<lander3d, ..., PID (upstream's terms), same form>                   cs_rollout_pid"),
            ("step_many_kernelILi0ELi0ELb1ELi2ELb1ELi2E", "step_many_kernel<lander3d, ..., random policy, same form>                             cs_rollout_random"),
            ("step_many_kernelILi0ELi0ELb1ELi0ELb1ELi0E", "step_many_kernel<lander3d, ..., open loop, LDS transpose>                              cs_step_many, > 65 536 envs"))
listing = R + "/gym_copter_amd/csrc/build/copterstep_kernels-hip-amdgcn-amd-amdhsa-gfx950.s"
subprocess.run(["make", "-C", R + "/gym_copter_amd/csrc", "asm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
if os.path.exists(listing):
    import re
    txt = open(listing).read()
    rows = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
        g = lambda k: int(re.search(r"\.amdhsa_%s (\d+)" % k, m.group(2)).group(1))
        rows.append((m.group(1), g("next_free_vgpr"), g("next_free_sgpr"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    nk = len([r for r in rows if "step_kernel" in r[0] or "step_many_kernel" in r[0]])
    with open(R + "/profiles/%s_summary.txt" % RND, "a") as f:
        f.write("\n== kernel resources of the default instantiations (ISA listing of this tree: make asm; occupancy = wavefronts per SIMD "
                "= min(8, 512 // (VGPRs rounded up to 8))) ==\n")
        for key, what in DEFAULTS:
            for name, v, sg, sc, lds in rows:
                if key in name:
                    f.write("%-100s VGPR %3d  SGPR %3d  scratch %d  LDS %4d  occupancy %d\n"
                            % (what, v, sg, sc, lds, min(8, 512 // ((v + 7) // 8 * 8))))
        f.write("%d kernels in the listing, %d of them step / step_many instantiations; with scratch: %d\n"
                % (len(rows), nk, sum(1 for r in rows if r[3] > 0)))

import bench  # noqa: E402  (kernel_source_hash only; nothing touches a GPU)
summ = json.load(open(P + "/summary.json"))
commit = subprocess.run(["git", "-C", R, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
t = {"commit": commit, "kernel_source_sha16": bench.kernel_source_hash(),
     "note": "HBM bytes per launch by rocprofv3 PMC: 2*FETCH_SIZE*1024 + WRITE_SIZE*1024, separate passes (scripts/profile_gpu.sh); "
             "bench.py reports a figure only while the kernel sources hash to kernel_source_sha16"}
for k, v in summ.items():
    if isinstance(v, (int, float)) and (k.startswith("lander3d_") or k.startswith("hover3d_")):
        t[k] = v
json.dump(t, open(R + "/profiles/traffic.json", "w"), indent=1)
# executed-instruction / executed-flop counts (bench.py: issue bounds of the K-step legs and of the headline, config 5's
# float64-ALU bound), under the same stamp
pc = dict(summ.get("pmc_counts", {}), commit=commit, kernel_source_sha16=bench.kernel_source_hash(),
          note="rocprofv3 PMC per wavefront (SQ_INSTS_VALU / SQ_WAVES; K-step kernels: also per env-step; flops = "
               "SQ_INSTS_VALU_ADD_F64 + MUL_F64 + TRANS_F64 + 2 x FMA_F64 per wavefront = per env), scripts/profile_gpu.sh; "
               "bench.py uses a figure only while the kernel sources hash to kernel_source_sha16")
json.dump(pc, open(R + "/profiles/pmc_counts.json", "w"), indent=1)
print(json.dumps(pc, indent=1))
print(json.dumps(t, indent=1))
b = json.load(open(P + "/bench_unprofiled.json"))
print("bench default: %.2f G env-steps/s, %.3f us/step, frac %.3f" % (b["value"] / 1e9, b["ms_per_step"] * 1e3,
                                                                    b["roofline"]["frac"]))
print(json.dumps(b.get("summary"), indent=1))
