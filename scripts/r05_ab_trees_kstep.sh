#!/bin/bash
# GPU box: the round-4 tree (scratch/r4tree: `git worktree add scratch/r4tree 57a1771` + make) against this tree, the
# K-STEP legs and the headline of each tree's own default bench line, interleaved in ONE run, three passes.
#   -> gpurun_out/<tag>/ab_r4_vs_r5_kstep.txt
set -u
TAG=${1:-r05treesk}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
run() {  # <tree>
  python3 $1/bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-span --served 0 --no-sweep 2>/dev/null \
    | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['summary']
k=s['k_step_us']; f=s.get('fused_caller_policy_us',{})
print('headline %.3f | step_many %.3f | rollout_pid %.3f | rollout_random %.3f | rollout_policy_linear %.3f | fused closed-loop law %.3f | fused linear %.3f | fused replay %.3f' % (
  d['ms_per_step']*1e3, k['step_many'], k['rollout_pid'], k['rollout_random'], k['rollout_policy_linear'],
  f.get('closed_loop_law_with_state',0), f.get('linear_policy_44_weights',0), f.get('replay_policy',0)))"
}
for pass in 1 2 3; do
  echo "pass $pass round4 | $(run $R/scratch/r4tree)" | tee -a $OUT/ab_r4_vs_r5_kstep.txt
  echo "pass $pass round5 | $(run $R)" | tee -a $OUT/ab_r4_vs_r5_kstep.txt
done
