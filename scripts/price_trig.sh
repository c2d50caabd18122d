#!/bin/bash
# Run ON THE GPU BOX (via gpurun): what the short sin / cos polynomials of the float32 storage modes save in time and cost in
# parity.   make -C gym_copter_amd/csrc exp NAME=fulltrig DEFS=-DCS_EXP_FULLTRIG   first.
#   usage: scripts/price_trig.sh <tag>   -> gpurun_out/<tag>/{ab.txt, fuzz_short.txt, fuzz_full.txt, straddle_short.txt, straddle_full.txt}
set -u
TAG=${1:-trig}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
FULL=$R/gym_copter_amd/csrc/build/libcopterstep_fulltrig.so
timeout 900 python3 tools/lib_ab.py short=gym_copter_amd/libcopterstep.so full=gym_copter_amd/csrc/build/libcopterstep_fulltrig.so --reps 3 > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
T=tests/test_gpu_numerics.py::test_a_differing_stored_word_after_one_step_is_a_straddled_rounding_boundary
timeout 300 python3 -m pytest -q -s $T > $OUT/straddle_short.txt 2>&1; grep -E "overall|per component|passed|failed" $OUT/straddle_short.txt
COPTERSTEP_LIB=$FULL timeout 300 python3 -m pytest -q -s $T > $OUT/straddle_full.txt 2>&1; grep -E "overall|per component|passed|failed" $OUT/straddle_full.txt
COPTERSTEP_LIB=$FULL timeout 900 python3 tools/fuzz_sweep.py 64 3200 > $OUT/fuzz_full.txt 2>&1; tail -2 $OUT/fuzz_full.txt
timeout 900 python3 tools/fuzz_sweep.py 64 3200 > $OUT/fuzz_short.txt 2>&1; tail -2 $OUT/fuzz_short.txt
# the two expected failures of the suite (seeds 181, 360) under the full polynomials
COPTERSTEP_LIB=$FULL timeout 600 python3 -m pytest -q tests/test_gpu_fuzz.py > $OUT/fuzz_suite_full.txt 2>&1; tail -3 $OUT/fuzz_suite_full.txt
