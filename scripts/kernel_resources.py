#!/usr/bin/env python3
"""Per-kernel registers / scratch / LDS from the gfx950 ISA listing (make -C gym_copter_amd/csrc asm)."""
import re, sys, subprocess
S = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else "gym_copter_amd/csrc/build/copterstep_kernels-hip-amdgcn-amd-amdhsa-gfx950.s"
txt = open(S).read()
rows = []
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: int(re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1))
    rows.append((name, g("next_free_vgpr"), g("next_free_sgpr"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
try:
    dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
except Exception:
    dem = [r[0] for r in rows]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
for (name, v, s_, sc, lds), d in zip(rows, dem):
    d = re.sub(r"\(.*", "", d).replace("cs::(anonymous namespace)::", "").replace("void ", "")
    if filt in d:
        print("%-70s vgpr %3d sgpr %3d scratch %4d lds %5d" % (d, v, s_, sc, lds))
