#!/bin/bash
# Run ON THE GPU BOX (via gpurun): collects the rocprofv3 evidence bench.py's roofline blocks are checked
# against.  Kernel timing and every PMC counter set are collected in SEPARATE runs (no trace domain beside --pmc).
#   usage: scripts/profile_gpu.sh <tag>      -> gpurun_out/<tag>/...      (round 3 layout)
# Served stepping (a persistent kernel fed by other kernels) is left out of every profiled run: counter
# collection serialises dispatches, which a persistent kernel waiting for its feeders cannot survive.
set -u
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# every step is bounded on its own: one stuck profiler run must not eat the GPU budget of the others
T="timeout ${STEP_TIMEOUT:-400}"
step() {  # <label>: note how long the previous step took (progress.txt travels back even if a later step is killed)
  echo "$(date +%s) $1" >> $OUT/progress.txt
}
# --full: every leg (the default bench run is a subset); the profiled runs keep the compact stdout line in their logs and
# drop their full record, the un-profiled run's full record is kept beside its line
BENCH0="python3 $R/bench.py --steps 1000 --warmup 200 --no-cpu-baseline --full"
BENCH="$BENCH0 --full-out /dev/null"
LIGHT="$BENCH --no-sweep --pid 0 --many 0 --served 0"
step bench_unprofiled
# 0. the un-profiled bench line of the same build (what the profiled figures are compared with) + its full record
$T $BENCH0 --full-out $OUT/bench_full_unprofiled.json > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
step trace
# 1. per-kernel time of the bench default command (65 536 envs, hipGraph replay, with the sweep, config 5 and
#    the K-step extras).  The summary groups kernel_trace.csv by (kernel name, grid size): the 262 144-env and
#    1 M-env sweep points run the same instantiation and are told apart by their grids.
$T rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- $BENCH --served 0 > $OUT/trace.log 2>&1
step trace_4m
# 1b. the headline kernel alone at 4 194 304 envs, where the per-dispatch cost of tracing is negligible
$T rocprofv3 --kernel-trace --stats -d $OUT/trace_4m --output-format csv -- $LIGHT --envs 4194304 --steps 200 --warmup 50 --ring 4 > $OUT/trace_4m.log 2>&1
step trace_rccl
# 1c. a 1-rank RCCL group whose all-gathers are really issued (bench.py --gather on one GPU): the kernel names
#     RCCL launched are the evidence that the collective path ran, eagerly and from hipGraphs
$T rocprofv3 --kernel-trace --stats -d $OUT/trace_rccl --output-format csv -- $LIGHT --gather --steps 200 --warmup 50 > $OUT/trace_rccl.log 2>&1
step pmc_traffic
# 2. HBM traffic (FETCH_SIZE and WRITE_SIZE in their own passes: TCC slots), eager launches so that every
#    dispatch is attributed: headline, 4 M envs, BASELINE configs[2] (Hover3D 262 144), configs[4] (10 substeps)
pmc_pair() {  # <name> <bench args...>
  local name=$1; shift
  $T rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch_$name --output-format csv -- $LIGHT --no-graph --steps 200 --warmup 50 --regions 1 --min-region-ms 1 --ring 4 "$@" > $OUT/fetch_$name.log 2>&1
  $T rocprofv3 --pmc WRITE_SIZE -d $OUT/write_$name --output-format csv -- $LIGHT --no-graph --steps 200 --warmup 50 --regions 1 --min-region-ms 1 --ring 4 "$@" > $OUT/write_$name.log 2>&1
}
pmc_pair lander3d_65536 --envs 65536
pmc_pair lander3d_4194304 --envs 4194304
pmc_pair hover3d_262144 --task hover3d --envs 262144
pmc_pair lander3d_65536_substeps10 --envs 65536 --substeps 10 --actions near_hover
step pmc_sq
# 3. instruction mix / wave cycles (SQ counters, 8 per pass): the headline with the K-step kernels of the extras,
#    then configs[4] and configs[2] on their own
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES"
SQ2="SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
SQB="$BENCH --no-sweep --served 0 --no-graph --steps 200 --warmup 50 --regions 1 --min-region-ms 1"
$T rocprofv3 --pmc $SQ1 -d $OUT/sq1 --output-format csv -- $SQB > $OUT/sq1.log 2>&1
$T rocprofv3 --pmc $SQ2 -d $OUT/sq2 --output-format csv -- $SQB > $OUT/sq2.log 2>&1
$T rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/l2 --output-format csv -- $SQB > $OUT/l2.log 2>&1
for cfgname in "c5 --envs 65536 --substeps 10 --actions near_hover" "c3 --task hover3d --envs 262144"; do
  set -- $cfgname; name=$1; shift
  $T rocprofv3 --pmc $SQ1 -d $OUT/sq1_$name --output-format csv -- $LIGHT --no-graph --steps 200 --warmup 50 --regions 1 --min-region-ms 1 --ring 4 "$@" > $OUT/sq1_$name.log 2>&1
  $T rocprofv3 --pmc $SQ2 -d $OUT/sq2_$name --output-format csv -- $LIGHT --no-graph --steps 200 --warmup 50 --regions 1 --min-region-ms 1 --ring 4 "$@" > $OUT/sq2_$name.log 2>&1
done
step pmc_flops
# 3b. float64 arithmetic executed (the flop count bench.py prices config 5 with): ADD + MUL + TRANS + 2 x FMA per
#     wavefront = per env, for configs[4], the headline and the K-step kernels (SQ counters: their own pass)
FL="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"
$T rocprofv3 --pmc $FL -d $OUT/fl_c5 --output-format csv -- $LIGHT --no-graph --steps 200 --warmup 50 --regions 1 --min-region-ms 1 --ring 4 --envs 65536 --substeps 10 --actions near_hover > $OUT/fl_c5.log 2>&1
$T rocprofv3 --pmc $FL -d $OUT/fl --output-format csv -- $SQB > $OUT/fl.log 2>&1
step pmc_large
# 3c. round 5: the sizes where an instruction-issue bound binds (many wavefronts per SIMD) -- executed instructions AND
#     float64 flops of configs[4] at 1 048 576 envs, and of the K-step kernels at 4 194 304 envs (above 65 536 envs their
#     observation rows go through the LDS transpose: a different instantiation from the 65 536-env legs); SQ_WAIT /
#     SQ_WAVE_CYCLES in a second pass.  tools/kstep_probe.py = one leg and nothing else (eager launches).
$T rocprofv3 --pmc $FL -d $OUT/fl_c5_1m --output-format csv -- $LIGHT --no-graph --steps 60 --warmup 20 --regions 1 --min-region-ms 1 --ring 4 --envs 1048576 --substeps 10 --actions near_hover > $OUT/fl_c5_1m.log 2>&1
$T rocprofv3 --pmc $SQ2 -d $OUT/sq2_c5_1m --output-format csv -- $LIGHT --no-graph --steps 60 --warmup 20 --regions 1 --min-region-ms 1 --ring 4 --envs 1048576 --substeps 10 --actions near_hover > $OUT/sq2_c5_1m.log 2>&1
for leg in many pid; do
  $T rocprofv3 --pmc $SQ1 -d $OUT/sq1_${leg}_4m --output-format csv -- python3 $R/tools/kstep_probe.py $leg 4194304 16 3 > $OUT/sq1_${leg}_4m.log 2>&1
  $T rocprofv3 --pmc $SQ2 -d $OUT/sq2_${leg}_4m --output-format csv -- python3 $R/tools/kstep_probe.py $leg 4194304 16 3 > $OUT/sq2_${leg}_4m.log 2>&1
done
step spans
# 4. the kernel's own duration per launch, un-profiled: first wavefront start -> last wavefront end on the
#    100 MHz clock in the span build (tools/kernel_span.py; phases not serialised)
if [ -f $R/gym_copter_amd/csrc/build/libcopterstep_span.so ]; then
  $T python3 $R/tools/kernel_span.py lander3d 65536 uniform 1 > $OUT/span_lander3d_65536.json 2> $OUT/span.err
  $T python3 $R/tools/kernel_span.py hover3d 262144 uniform 1 > $OUT/span_hover3d_262144.json 2>> $OUT/span.err
  $T python3 $R/tools/kernel_span.py lander3d 65536 near_hover 10 > $OUT/span_lander3d_65536_substeps10.json 2>> $OUT/span.err
  $T python3 $R/tools/kernel_span.py lander3d 4194304 uniform 1 > $OUT/span_lander3d_4194304.json 2>> $OUT/span.err
fi
step rccl_log
# 5. what RCCL itself logs for the forced 1-rank collectives (RCCL turns a 1-rank all-gather into a device copy:
#    no kernel appears in a trace, so its own call log is the evidence that the collective path ran)
NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=COLL $T $LIGHT --gather --steps 20 --warmup 5 --no-graph --regions 1 --min-region-ms 1 > $OUT/rccl_debug.json 2> $OUT/rccl_debug.err
grep -c "AllGather" $OUT/rccl_debug.err > $OUT/rccl_allgather_calls.txt
grep "AllGather" $OUT/rccl_debug.err | head -6 | cut -c1-260 >> $OUT/rccl_allgather_calls.txt
step summarize
python3 $R/scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# gpurun copies back at most 64 MiB: keep the stats, the summaries and the logs, drop the per-dispatch CSVs
find $OUT -name "*kernel_trace.csv" -delete -o -name "*counter_collection.csv" -delete -o -name "*.db" -delete
rm -f $OUT/rccl_debug.err
step done
awk 'NR>1{print prev_label, $1-prev_t " s"} {prev_t=$1; prev_label=$2}' $OUT/progress.txt
du -sh $OUT | cut -f1
