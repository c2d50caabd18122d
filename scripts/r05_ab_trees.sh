#!/bin/bash
# GPU box: the round-4 tree (scratch/r4tree: `git worktree add scratch/r4tree 57a1771` + make) against this tree,
# interleaved in ONE run, headline leg of each tree's own bench.py (hipGraph replay), three passes.
#   -> gpurun_out/<tag>/ab_r4_vs_r5.txt
set -u
TAG=${1:-r05trees}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
run() {  # <tree> <envs> <law> <substeps> <task> <ring> <steps>
  python3 $1/bench.py --envs $2 --actions $3 --substeps $4 --task $5 --ring $6 --steps $7 --warmup 100 --no-sweep --pid 0 --many 0 --served 0 --no-cpu-baseline --no-span --regions 5 ${EXTRA:-} 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f' % (d['ms_per_step']*1e3))"
}
for pass in 1 2 3; do
  for cfg in "262144 uniform 1 hover3d 16 2000" "65536 uniform 1 lander3d 64 2000" "65536 near_hover 1 lander3d 64 2000" "65536 near_hover 10 lander3d 64 2000" "262144 uniform 1 lander3d 16 2000" "1048576 uniform 1 hover3d 8 1000" "4194304 uniform 1 lander3d 4 200" "4194304 uniform 1 hover3d 4 200"; do
    a=$(run $R/scratch/r4tree $cfg); b=$(run $R $cfg)
    echo "pass $pass | $cfg | round4 $a us | round5 $b us" | tee -a $OUT/ab_r4_vs_r5.txt
  done
done
