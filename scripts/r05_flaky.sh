#!/bin/bash
# GPU box: run the GPU suite N times in fresh processes and collect any failure, then the served-stepping files M times
# under rocgdb to get a native backtrace of a crash at interpreter exit.   bash scripts/r05_flaky.sh <tag> [N=5] [M=5]
tag=${1:-r05flaky}; N=${2:-5}; M=${3:-5}
out=gpurun_out/$tag; mkdir -p $out
export PYTHONFAULTHANDLER=1
for i in $(seq 1 $N); do
  timeout 900 python3 -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider > $out/full_$i.log 2>&1
  echo "full $i rc=$? $(tail -1 $out/full_$i.log)" | tee -a $out/summary.txt
done
for i in $(seq 1 $M); do
  timeout 900 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint" -ex run -ex "bt 40" -ex "info sharedlibrary" \
      --args python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py -q -m gpu -p no:cacheprovider > $out/gdb_$i.log 2>&1
  echo "gdb $i rc=$? $(grep -c 'SIGSEGV' $out/gdb_$i.log) sigsegv; $(grep -h 'passed\|failed' $out/gdb_$i.log | tail -1)" | tee -a $out/summary.txt
done
grep -h "AssertionError\|^FAILED\|Fatal Python" $out/full_*.log | head -40 >> $out/summary.txt
tail -30 $out/summary.txt
