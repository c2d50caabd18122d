#!/bin/bash
# CPU only (no GPU needed; GPU AddressSanitizer is not available on the pool): the HOST side of libcopterstep.so -- argument
# checking, configuration folding, error paths, the symbol table -- under AddressSanitizer + UndefinedBehaviorSanitizer.
# Device code is compiled as usual (-Xarch_host keeps the sanitizers off it).   usage: bash scripts/cpu_sanitize.sh
set -eu
cd "$(dirname "$0")/.."
make -C gym_copter_amd/csrc exp NAME=asan \
  DEFS="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer -g" > /dev/null
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD=$RT \
  COPTERSTEP_LIB=gym_copter_amd/csrc/build/libcopterstep_asan.so python3 -m pytest tests/test_abi_cpu.py -q
