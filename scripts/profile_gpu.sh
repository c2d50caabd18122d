#!/bin/bash
# Run ON THE GPU BOX (via gpurun): collects the rocprofv3 evidence bench.py's roofline block is
# checked against.  Kernel timing and PMC counters are collected in SEPARATE runs.
#   usage: scripts/profile_gpu.sh <tag>      -> gpurun_out/<tag>/...
set -u
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 1000 --warmup 200 --no-cpu-baseline"
LIGHT="$BENCH --no-sweep --pid 0 --many 0"
# 0. the un-profiled bench line of the same build (what the profiled figures are compared with)
$BENCH > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
# 1. per-kernel time, the same command as the bench default (65 536 envs, hipGraph replay, with the
#    sweep, config 5 and the K-step extras: every instantiation appears under its own name)
rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- $BENCH > $OUT/trace.log 2>&1
# 1b. the headline kernel alone at 4 194 304 envs, where the per-dispatch cost of tracing is negligible
rocprofv3 --kernel-trace --stats -d $OUT/trace_4m --output-format csv -- $LIGHT --envs 4194304 --steps 200 --warmup 50 --ring 4 > $OUT/trace_4m.log 2>&1
# 2. HBM traffic of the step kernel: FETCH_SIZE and WRITE_SIZE in their own passes (TCC slots),
#    eager launches so that every dispatch is attributed; at 65 536 envs and at 4 194 304 envs
for N in 65536 4194304; do
  rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch_$N --output-format csv -- $LIGHT --no-graph --steps 200 --warmup 50 --regions 1 --min-region-ms 1 --envs $N --ring 4 > $OUT/fetch_$N.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $OUT/write_$N --output-format csv -- $LIGHT --no-graph --steps 200 --warmup 50 --regions 1 --min-region-ms 1 --envs $N --ring 4 > $OUT/write_$N.log 2>&1
done
# 3. instruction mix / wave cycles (SQ counters, 8 per pass), with the K-step kernels of the extras
SQB="$BENCH --no-sweep --no-graph --steps 200 --warmup 50 --regions 1 --min-region-ms 1"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES -d $OUT/sq1 --output-format csv -- $SQB > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $OUT/sq2 --output-format csv -- $SQB > $OUT/sq2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/l2 --output-format csv -- $SQB > $OUT/l2.log 2>&1
python3 $R/scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
