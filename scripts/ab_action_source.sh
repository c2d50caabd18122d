#!/bin/bash
# GPU box: is the non-temporal hint on the action-row load (used up to nt_action_max_envs envs) right when
# the actions are NOT a long resident ring but were just written by a kernel on (mostly) another XCD, as in a
# learner's loop?  A/B in ONE run, interleaved: {resident 64-deep ring, actions produced by a preceding copy
# kernel} x {hint on (default), hint off (COPTERSTEP_NT_ACTION_MAX_ENVS=1)} at 65 536 envs.
#   usage: scripts/ab_action_source.sh > gpurun_out/ab_action_source.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2 3; do
  for src in ring produced; do
    for hint in on off; do
      extra=""; [ $src = produced ] && extra="--produce-actions"
      nt=0; [ $hint = off ] && nt=1
      COPTERSTEP_NT_ACTION_MAX_ENVS=$nt python3 $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-sweep --pid 0 --many 0 $extra 2>/dev/null |
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('actions=$src nt_hint=$hint us_per_step %.3f' % (d['ms_per_step']*1e3))"
    done
  done
done | sort | awk '{k=$1" "$2; s[k]=s[k]" "$4} END{for(k in s) print k, "us per step (step + producer where there is one):", s[k]}' | sort
