#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the GPU test-suite, smoke() and the driver-form bench line of the tree as it stands.
#   usage: scripts/check_gpu.sh <tag> [pytest args]   -> gpurun_out/<tag>/...
set -u
TAG=${1:-chk}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests -m gpu -x -q "$@" > $OUT/gputests.log 2>&1
echo "pytest rc=$?" >> $OUT/gputests.log
tail -15 $OUT/gputests.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke rc=$?" >> $OUT/smoke.log
tail -3 $OUT/smoke.log
# the driver's exact command: the compact line on stdout (<= 8 000 bytes), the full record beside it; wall time noted
t0=$(date +%s%N)
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --full-out $OUT/bench_full.json > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
echo "bench rc=$? wall_ms=$(( ($(date +%s%N) - t0) / 1000000 )) bytes=$(wc -c < $OUT/bench_driver_form.json)" | tee $OUT/bench_driver_form.wall
cat $OUT/bench_driver_form.json
if [ -f $R/gym_copter_amd/csrc/build/libcopterstep_kstamps.so ]; then
  timeout 600 python3 $R/tools/kstep_stamps.py 65536 8 > $OUT/kstep_stamps.txt 2> $OUT/kstep_stamps.err
  tail -5 $OUT/kstep_stamps.txt
fi
