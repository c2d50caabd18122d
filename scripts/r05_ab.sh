#!/bin/bash
# GPU box: interleaved A/B, one run: the shipped library against timing builds that take the round-5 additions to the
# step path out one at a time -- tb5376: the round-4 tile stride (no EPH row); nocarry: the round-4 episode wrap instead
# of the carry branch; r4like: no high-part flag (v_and at unpack, v_or at pack); r4all: all three.
#   -> gpurun_out/<tag>/ab_round5_additions.txt
set -u
TAG=${1:-r05ab}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
B=gym_copter_amd/csrc/build
timeout 2400 python3 tools/ab_cfg.py --libs base=gym_copter_amd/libcopterstep.so r4all=$B/libcopterstep_r4all.so r4pad=$B/libcopterstep_r4pad.so r4padptr=$B/libcopterstep_r4padptr.so \
  --cfgs "262144 uniform 1 hover3d" "65536 uniform 1" "65536 near_hover 10" --reps 3 > $OUT/ab_round5_additions.txt 2>&1
cat $OUT/ab_round5_additions.txt
