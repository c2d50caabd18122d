#!/bin/bash
# GPU box: A/B builds of libcopterstep in ONE run, interleaved (rule: never compare timings taken in
# different runs/boxes).   tools/ab.sh path/libA.so path/libB.so ...
LIBS=("$@")
for rep in 1 2 3; do
for lib in "${LIBS[@]}"; do
  for cfg in "65536 uniform" "65536 near_hover" "262144 uniform"; do
    read n law <<< "$cfg"
    COPTERSTEP_LIB=$PWD/$lib python bench.py --envs $n --steps 2000 --warmup 200 --no-cpu-baseline --many 0 --ring 8 --actions $law 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['config']['actions'], d['config']['envs_per_gpu'], 'us/step %.3f'%(d['ms_per_step']*1e3))"
  done
done
done | sort | awk '{k=$1" "$2" "$3; s[k]=s[k]" "$5} END{for(k in s) print k, s[k]}' | sort
