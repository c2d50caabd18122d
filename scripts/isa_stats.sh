#!/bin/bash
# static instruction mix of one kernel from the gfx950 ISA listing:  scripts/isa_stats.sh step_kernelILi0ELi0ELb1ELb1ELb0ELb1ELi1E
cd "$(dirname "$0")/.." && make -C gym_copter_amd/csrc asm > /dev/null 2>&1
S=gym_copter_amd/csrc/build/copterstep_kernels-hip-amdgcn-amd-amdhsa-gfx950.s
K=${1:-step_kernelILi0ELi0ELb1ELb1ELb0ELb1ELi1E}
awk -v k="$K" '$0 ~ "^_ZN2cs12_GLOBAL__N_1[0-9]+" k ".*:" {f=1} f{print} f && /s_endpgm/ {c++} f && $0 ~ "\\.amdhsa_kernel" {exit}' $S > /tmp/k.s
echo "total $(grep -cE '^\s+[vs]_|^\s+(global|ds|buffer|flat)_' /tmp/k.s)  valu $(grep -cE '^\s+v_' /tmp/k.s)  f64 $(grep -cE '^\s+v_[a-z0-9_]*f64' /tmp/k.s)  salu $(grep -cE '^\s+s_' /tmp/k.s)  vmem $(grep -cE '^\s+(global|buffer|flat)_' /tmp/k.s)  lds $(grep -cE '^\s+ds_' /tmp/k.s)  branch $(grep -c s_cbranch /tmp/k.s)  readlane $(grep -c v_readlane /tmp/k.s) writelane $(grep -c v_writelane /tmp/k.s)"
grep -E "\.vgpr_count|\.sgpr_count|vgpr_spill|sgpr_spill|scratch" $S | head -0
