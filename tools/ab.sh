#!/bin/bash
# GPU box: A/B builds of libcopterstep in ONE run, interleaved (rule: never compare timings taken in
# different runs/boxes).   tools/ab.sh path/libA.so path/libB.so ...
LIBS=("$@")
for rep in 1 2 3; do
for lib in "${LIBS[@]}"; do
  IFS=';' read -ra CFGS <<< "${AB_CFGS:-65536 uniform 2000;65536 near_hover 2000;262144 uniform 2000}"
  for cfg in "${CFGS[@]}"; do
    read n law k <<< "$cfg"
    COPTERSTEP_LIB=$PWD/$lib python bench.py --envs $n --steps ${k:-2000} --warmup 100 --pid 0 --no-cpu-baseline --many 0 --ring 8 --actions $law 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['config']['actions'], d['config']['envs_per_gpu'], 'us/step %.3f'%(d['ms_per_step']*1e3))"
  done
done
done | sort | awk '{k=$1" "$2" "$3; s[k]=s[k]" "$5} END{for(k in s) print k, s[k]}' | sort
