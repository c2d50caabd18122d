#!/bin/bash
# Run ON THE GPU BOX (via gpurun): round-5 probes that need hardware and nothing from this round's library changes.
#   usage: scripts/r05_probe.sh <tag>   -> gpurun_out/<tag>/...
# 1. tools/ubench_f64: float64 issue cost with PROVEN co-residency
# 2. what clock / power / temperature sources an ordinary user can read on the box (sweep[*].clocks)
# 3. tools/rccl_two_ranks_one_gpu.py: one bounded attempt at a 2-rank RCCL group on one device
# 4. rocprofv3 --att on a K-step leg: does the thread-trace decoder exist in this image?
# 5. first numbers for the sweep points VERDICT asks for (config 5 at 1 M envs, K-step legs at 4 M envs)
# 6. the driver-form bench line of the build as it stands (box-matched baseline for this round's changes)
set -u
TAG=${1:-r05a}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T="timeout ${STEP_TIMEOUT:-300}"
step() { echo "$(date +%s) $1" >> $OUT/progress.txt; }

step ubench_f64
$T $R/gym_copter_amd/csrc/build/ubench_f64 61 > $OUT/ubench_f64.txt 2>&1

step clocks_probe
{
  echo "## id"; id
  echo "## /sys/class/drm"; ls -l /sys/class/drm/ 2>&1
  for d in /sys/class/drm/card*/device; do
    echo "## $d"
    for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk gpu_busy_percent mem_busy_percent current_link_speed power_dpm_force_performance_level; do
      echo "--- $f"; cat $d/$f 2>&1 | head -12
    done
    for h in $d/hwmon/hwmon*; do
      echo "## $h"; ls $h 2>&1 | tr '\n' ' '; echo
      for f in power1_average power1_input power1_cap temp1_input temp2_input temp3_input freq1_input freq2_input in0_input; do
        [ -e $h/$f ] && echo "$f = $(cat $h/$f 2>&1)"
      done
    done
  done
  echo "## rocm-smi"; timeout 60 rocm-smi --showclocks --showpower --showtemp 2>&1 | head -60
  echo "## amd-smi"; timeout 60 amd-smi metric --clock --power --temperature 2>&1 | head -80
} > $OUT/clocks_probe.txt 2>&1

step rccl_two_ranks
RCCL_TWO_RANK_DEADLINE_S=120 timeout 200 python3 $R/tools/rccl_two_ranks_one_gpu.py > $OUT/rccl_two_ranks.json 2> $OUT/rccl_two_ranks.err

step att
# (the program itself after `--`: no env / bash -c hop)
$T rocprofv3 --att --att-target-cu 1 -d $OUT/att_pid -- python3 $R/tools/kstep_probe.py pid 65536 4 2 > $OUT/att_pid.log 2>&1
ls -R $OUT/att_pid 2>/dev/null | head -40 >> $OUT/att_pid.log
# raw thread-trace data is large and undecodable without the decoder library: keep only the listing
find $OUT/att_pid -type f -size +1M -delete 2>/dev/null

step kstep_sizes
for leg in many pid random; do
  for n in 65536 1048576 4194304; do
    K=100; [ $n -ge 1048576 ] && K=20
    $T python3 $R/tools/kstep_probe.py $leg $n $K 5 >> $OUT/kstep_sizes.txt 2>> $OUT/kstep_sizes.err
  done
done
for n in 65536 262144 1048576 4194304; do
  $T python3 $R/tools/kstep_probe.py step $n 100 5 10 >> $OUT/kstep_sizes.txt 2>> $OUT/kstep_sizes.err
done

step bench_driver_form
$T python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
step done
