#!/bin/bash
# GPU box: the K-step legs only -- PID / rollout parity tests, un-instrumented per-step times (tools/kstep_probe.py, eager
# launches of 100 steps, three repetitions), phase stamps (tools/kstep_stamps.py).   -> gpurun_out/<tag>/
set -u
TAG=${1:-r05k}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests -m gpu -x -q -k "pid or rollout or step_many or served_closed" > $OUT/kstep_tests.log 2>&1
tail -3 $OUT/kstep_tests.log
for rep in 1 2 3; do
  for leg in many pid random; do
    timeout 300 python3 tools/kstep_probe.py $leg 65536 100 20 >> $OUT/kstep_probe.txt 2>> $OUT/kstep_probe.err
  done
done
for n in 262144 1048576 4194304; do
  for leg in many pid random; do timeout 300 python3 tools/kstep_probe.py $leg $n 16 10 >> $OUT/kstep_probe.txt 2>> $OUT/kstep_probe.err; done
done
python3 - <<PY
import json
rows=[json.loads(l) for l in open("$OUT/kstep_probe.txt")]
for leg in ("many","pid","random"):
    for n in (65536, 262144, 1048576, 4194304):
        v=sorted(r["us_per_env_step_batch"] for r in rows if r["leg"]==leg and r["envs"]==n)
        if v: print(leg, n, " ".join("%.3f"%x for x in v), "median %.3f" % v[len(v)//2])
PY
timeout 600 python3 tools/kstep_stamps.py 65536 8 > $OUT/kstep_stamps.txt 2> $OUT/kstep_stamps.err
grep -n "==\|loop top of\|policy (PID\|shaping\|setMotors\|auto-reset\|clip" $OUT/kstep_stamps.txt | head -40
