#!/bin/bash
# Run ON THE GPU BOX: batch-size / workload sweep of bench.py -> gpurun_out/<tag>/sweep.jsonl + table
TAG=${1:-sweep}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
: > $OUT/sweep.jsonl
run() { python3 $R/bench.py --no-cpu-baseline "$@" 2>/dev/null >> $OUT/sweep.jsonl; }
for law in uniform near_hover; do
  for n in 16384 65536 262144 1048576 4194304 16777216; do
    k=2000; [ $n -ge 1048576 ] && k=300; [ $n -ge 16777216 ] && k=100
    run --task lander3d --envs $n --actions $law --steps $k --warmup 100 --ring 8
  done
done
run --task lander3d --envs 65536 --actions const --steps 2000 --warmup 100                    # constant thrust (lander.py:21)
run --task lander3d --envs 262144 --actions const --steps 2000 --warmup 100
run --task hover3d --envs 262144 --actions uniform --steps 1000 --warmup 100 --ring 8        # BASELINE config 3
run --task hover3d --envs 262144 --actions near_hover --steps 1000 --warmup 100 --ring 8
run --task lander3d --envs 65536 --actions near_hover --substeps 10 --steps 1000 --warmup 100   # BASELINE config 5
run --task lander3d --envs 65536 --actions uniform --state float32_rn --steps 2000 --warmup 100
run --task lander3d --envs 65536 --actions uniform --state float64 --steps 2000 --warmup 100
run --task lander3d --envs 65536 --actions uniform --no-graph --steps 2000 --warmup 100
python3 - <<PY
import json
rows=[json.loads(l) for l in open("$OUT/sweep.jsonl") if l.strip()]
print("%-9s %10s %-10s %-10s %4s %-6s %10s %12s %9s %7s"%("task","envs","actions","state","sub","launch","us/step","Genv-steps/s","algoGB/s","frac"))
for d in rows:
    c=d["config"]
    print("%-9s %10d %-10s %-10s %4d %-6s %10.3f %12.2f %9.0f %7.3f"%(c["task"],c["envs_per_gpu"],c["actions"],c["state_words"],c["substeps"],"graph" if "hipGraph" in c["workload"] else "eager",d["ms_per_step"]*1e3,d["value"]/1e9,d["roofline"]["achieved"],d["roofline"]["frac"]))
PY
