#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the GPU-side jobs behind the files under profiles/, one parameterised script
# (round 6: folds the per-round scripts r05_check / r05_ab / r05_ab_trees / r05_ab_trees_kstep / r05_contention /
# r05_flaky / r05_kstep / r05_probe / price_trig; `git log -- scripts/` has them as they ran).
#   usage: bash scripts/gpu_jobs.sh <job> <tag> [args...]        -> gpurun_out/<tag>/...
#     check      [pytest args]          GPU test-suite, smoke(), the driver-form bench line (+ wall time), K-step stamps if built
#     ab-libs    "<name=lib.so> ..." "<cfg>" ["<cfg>" ...]       interleaved A/B of library builds over bench.py
#                                       configurations "<envs> <law> <substeps> [task]" (tools/ab_cfg.py, 3 passes)
#     ab-line    "<name=lib.so> ..." [--full]                    the same over the legs of the default bench line (tools/lib_ab.py)
#     ab-trees   <other tree> [kstep]   another checkout (e.g. `git worktree add scratch/r5tree <commit>` + make) against this
#                                       one, each tree's own bench.py, interleaved, three passes; `kstep` = the K-step legs
#     price-trig                        short vs full sin / cos polynomials: time, one-unit rate, fuzz sweep (build
#                                       `make -C gym_copter_amd/csrc exp NAME=fulltrig DEFS=-DCS_EXP_FULLTRIG` first)
#     kstep                             K-step parity tests, un-instrumented per-step times by batch size, phase stamps
#     flaky      [N=5] [M=5]            the GPU suite N times in fresh processes; the served tests M times under rocgdb
#     contention [N=3]                  the race-prone tests while another process keeps the device busy
#     probe                             hardware probes: float64 issue cost, readable sensors, 2 ranks on one device, ATT
set -u
JOB=${1:?job}; TAG=${2:?tag}; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
B=gym_copter_amd/csrc/build
SERVED="tests/test_gpu_served.py tests/test_gpu_stepping_forms.py"
case $JOB in
check)
  timeout 900 python3 -m pytest tests -m gpu -x -q "$@" > $OUT/gputests.log 2>&1
  echo "pytest rc=$?" >> $OUT/gputests.log; tail -15 $OUT/gputests.log
  timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
  echo "smoke rc=$?" >> $OUT/smoke.log; tail -3 $OUT/smoke.log
  # the driver's exact command: the compact line on stdout (<= 8 000 bytes), the full record beside it; wall time noted
  t0=$(date +%s%N)
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --full-out $OUT/bench_full.json > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
  echo "bench rc=$? wall_ms=$(( ($(date +%s%N) - t0) / 1000000 )) bytes=$(wc -c < $OUT/bench_driver_form.json)" | tee $OUT/bench_driver_form.wall
  cat $OUT/bench_driver_form.json
  if [ -f $R/$B/libcopterstep_kstamps.so ]; then
    timeout 600 python3 tools/kstep_stamps.py 65536 8 > $OUT/kstep_stamps.txt 2> $OUT/kstep_stamps.err; tail -5 $OUT/kstep_stamps.txt
  fi ;;
ab-libs)
  LIBS=$1; shift
  timeout 2400 python3 tools/ab_cfg.py --libs $LIBS --cfgs "$@" --reps 3 > $OUT/ab_libs.txt 2>&1; cat $OUT/ab_libs.txt ;;
ab-line)
  LIBS=$1; shift
  timeout 1800 python3 tools/lib_ab.py $LIBS --reps 3 "$@" > $OUT/ab_line.txt 2>&1; cat $OUT/ab_line.txt ;;
ab-trees)
  OTHER=$(cd $1 && pwd); MODE=${2:-headline}
  cd /tmp
  if [ $MODE = kstep ]; then
    run() { python3 $1/bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-span --served 0 --no-sweep --full 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['summary']['k_step_us']
print('headline %.3f | ' % (d['ms_per_step']*1e3) + ' | '.join('%s %.3f' % kv for kv in k.items()))"; }
    for pass in 1 2 3; do
      echo "pass $pass other | $(run $OTHER)" | tee -a $OUT/ab_trees_kstep.txt
      echo "pass $pass this  | $(run $R)" | tee -a $OUT/ab_trees_kstep.txt
    done
  else
    run() {  # <tree> <envs> <law> <substeps> <task> <ring> <steps>
      python3 $1/bench.py --envs $2 --actions $3 --substeps $4 --task $5 --ring $6 --steps $7 --warmup 100 --no-sweep --pid 0 --many 0 --served 0 --no-cpu-baseline --no-span --regions 5 2>/dev/null \
        | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f' % (d['ms_per_step']*1e3))"; }
    for pass in 1 2 3; do
      for cfg in "262144 uniform 1 hover3d 16 2000" "65536 uniform 1 lander3d 64 2000" "65536 near_hover 1 lander3d 64 2000" "65536 near_hover 10 lander3d 64 2000" "262144 uniform 1 lander3d 16 2000" "1048576 uniform 1 lander3d 8 300" "4194304 uniform 1 lander3d 4 200"; do
        echo "pass $pass | $cfg | other $(run $OTHER $cfg) us | this $(run $R $cfg) us" | tee -a $OUT/ab_trees.txt
      done
    done
  fi ;;
price-trig)
  FULL=$R/$B/libcopterstep_fulltrig.so
  T=tests/test_gpu_numerics.py::test_a_differing_stored_word_after_one_step_is_a_straddled_rounding_boundary
  timeout 900 python3 tools/lib_ab.py short=gym_copter_amd/libcopterstep.so full=$B/libcopterstep_fulltrig.so --reps 3 > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
  timeout 300 python3 -m pytest -q -s $T > $OUT/straddle_short.txt 2>&1; grep -E "overall|per component|passed|failed" $OUT/straddle_short.txt
  COPTERSTEP_LIB=$FULL timeout 300 python3 -m pytest -q -s $T > $OUT/straddle_full.txt 2>&1; grep -E "overall|per component|passed|failed" $OUT/straddle_full.txt
  COPTERSTEP_LIB=$FULL timeout 900 python3 tools/fuzz_sweep.py 64 3200 > $OUT/fuzz_full.txt 2>&1; tail -2 $OUT/fuzz_full.txt
  timeout 900 python3 tools/fuzz_sweep.py 64 3200 > $OUT/fuzz_short.txt 2>&1; tail -2 $OUT/fuzz_short.txt
  COPTERSTEP_LIB=$FULL timeout 600 python3 -m pytest -q tests/test_gpu_fuzz.py > $OUT/fuzz_suite_full.txt 2>&1; tail -3 $OUT/fuzz_suite_full.txt ;;
kstep)
  timeout 900 python3 -m pytest tests -m gpu -x -q -k "pid or rollout or step_many or served_closed" > $OUT/kstep_tests.log 2>&1; tail -3 $OUT/kstep_tests.log
  for rep in 1 2 3; do for leg in many pid random; do
    timeout 300 python3 tools/kstep_probe.py $leg 65536 100 20 >> $OUT/kstep_probe.txt 2>> $OUT/kstep_probe.err; done; done
  for n in 262144 1048576 4194304; do for leg in many pid random; do
    timeout 300 python3 tools/kstep_probe.py $leg $n 16 10 >> $OUT/kstep_probe.txt 2>> $OUT/kstep_probe.err; done; done
  python3 - <<PY
import json
rows=[json.loads(l) for l in open("$OUT/kstep_probe.txt")]
for leg in ("many","pid","random"):
    for n in (65536, 262144, 1048576, 4194304):
        v=sorted(r["us_per_env_step_batch"] for r in rows if r["leg"]==leg and r["envs"]==n)
        if v: print(leg, n, " ".join("%.3f"%x for x in v), "median %.3f" % v[len(v)//2])
PY
  timeout 300 python3 tools/kstep_outputs_probe.py 65536 20 > $OUT/kstep_outputs_probe.txt 2>&1; tail -3 $OUT/kstep_outputs_probe.txt
  [ -f $R/$B/libcopterstep_kstamps.so ] && timeout 600 python3 tools/kstep_stamps.py 65536 8 > $OUT/kstep_stamps.txt 2> $OUT/kstep_stamps.err ;;
flaky)
  N=${1:-5}; M=${2:-5}; export PYTHONFAULTHANDLER=1
  for i in $(seq 1 $N); do
    timeout 900 python3 -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/full_$i.log 2>&1
    echo "full $i rc=$? $(tail -1 $OUT/full_$i.log)" | tee -a $OUT/summary.txt
  done
  for i in $(seq 1 $M); do
    timeout 900 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint" -ex run -ex "bt 40" -ex "info sharedlibrary" \
        --args python3 -m pytest $SERVED -q -m gpu -p no:cacheprovider > $OUT/gdb_$i.log 2>&1
    echo "gdb $i rc=$? $(grep -c 'SIGSEGV' $OUT/gdb_$i.log) sigsegv; $(grep -h 'passed\|failed' $OUT/gdb_$i.log | tail -1)" | tee -a $OUT/summary.txt
  done
  grep -h "AssertionError\|^FAILED\|Fatal Python" $OUT/full_*.log | head -40 >> $OUT/summary.txt; tail -30 $OUT/summary.txt ;;
contention)
  N=${1:-3}
  python3 - > $OUT/load.log 2>&1 <<'PY' &
import time, torch, sys
sys.path.insert(0, ".")
import gym_copter_amd as gca
env = gca.CopterVecEnv(task="lander3d", num_envs=262144, seed=1, autoreset_mode="next_step")
env.reset()
a = torch.rand((262144, 4), device="cuda") * 2 - 1
t0 = time.time(); n = 0
while time.time() - t0 < 420:
    for _ in range(200):
        env.step(a)
    torch.cuda.synchronize(); n += 200
    time.sleep(0.002)
print("load: %d steps" % n)
PY
  LOAD=$!
  sleep 5
  for i in $(seq 1 $N); do
    timeout 600 python3 -m pytest ${FILES:-$SERVED tests/test_gpu_fuzz.py} -q -m gpu -p no:cacheprovider -k "not bench and not rccl and not span" > $OUT/run_$i.log 2>&1
    echo "run $i rc=$? $(tail -1 $OUT/run_$i.log)" | tee -a $OUT/summary.txt
  done
  kill $LOAD 2>/dev/null; wait $LOAD 2>/dev/null
  grep -h "^FAILED\|Error" $OUT/run_*.log | head -20 >> $OUT/summary.txt; tail -12 $OUT/summary.txt ;;
probe)
  cd /tmp && export TMPDIR=/tmp
  T="timeout ${STEP_TIMEOUT:-300}"
  [ -x $R/$B/ubench_f64 ] && $T $R/$B/ubench_f64 61 > $OUT/ubench_f64.txt 2>&1
  { echo "## id"; id
    for d in /sys/class/drm/card*/device; do
      echo "## $d"
      for f in pp_dpm_sclk pp_dpm_mclk gpu_busy_percent power_dpm_force_performance_level; do echo "--- $f"; cat $d/$f 2>&1 | head -12; done
      for h in $d/hwmon/hwmon*; do
        echo "## $h"; ls $h 2>&1 | tr '\n' ' '; echo
        for f in power1_average power1_input temp1_input temp2_input temp3_input freq1_input freq2_input; do [ -e $h/$f ] && echo "$f = $(cat $h/$f 2>&1)"; done
      done
    done; } > $OUT/clocks_probe.txt 2>&1
  RCCL_TWO_RANK_DEADLINE_S=120 timeout 200 python3 $R/tools/rccl_two_ranks_one_gpu.py > $OUT/rccl_two_ranks.json 2> $OUT/rccl_two_ranks.err
  # (the program itself after `--`: no env / bash -c hop)
  $T rocprofv3 --att --att-target-cu 1 -d $OUT/att_pid -- python3 $R/tools/kstep_probe.py pid 65536 4 2 > $OUT/att_pid.log 2>&1
  ls -R $OUT/att_pid 2>/dev/null | head -40 >> $OUT/att_pid.log
  find $OUT/att_pid -type f -size +1M -delete 2>/dev/null ;;
*) echo "unknown job $JOB" >&2; exit 2 ;;
esac
