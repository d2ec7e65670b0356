#!/bin/bash
# Sanitizer tier: AddressSanitizer + UndefinedBehaviorSanitizer builds of the product's HOST code (api.hip, serialize.hip,
# hosttest.hip and the host halves of every other translation unit; device code is compiled as usual, -fno-gpu-sanitize) and of the
# CPU oracle, then the whole CPU test tier (`-m "not gpu and not perf"`) against them.  No GPU needed.
#   tools/sanitize.sh build   -> groth16_amd/libg16_asan.so, oracle/libg16_oracle_asan.so   (~10 min on 8 cores)
#   tools/sanitize.sh test [pytest args]   -> runs the tier; a sanitizer report aborts the process, so a green run means none fired
set -e
cd "$(dirname "$0")/.."
LLVM=/opt/rocm/lib/llvm
RT=$(ls $LLVM/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-gpu-sanitize -fno-omit-frame-pointer -shared-libsan -g1"
case "${1:-build}" in
  build)
    make -C groth16_amd/csrc -j"$(nproc)" OBJDIR=build_asan OUT=../libg16_asan.so EXTRA="$SAN" LDEXTRA="-fsanitize=address,undefined -shared-libsan" \
      > /tmp/g16_asan_build.log 2>&1 || { tail -30 /tmp/g16_asan_build.log; exit 1; }
    $LLVM/bin/clang++ -O1 -mbmi2 -madx -std=c++17 -fopenmp -fPIC -Wall -Wno-unused-function -fsanitize=address,undefined \
      -fno-sanitize-recover=undefined -fno-omit-frame-pointer -shared-libsan -g1 -shared -o oracle/libg16_oracle_asan.so oracle/g16_oracle.cpp
    ls -la groth16_amd/libg16_asan.so oracle/libg16_oracle_asan.so ;;
  test)
    shift
    export G16_LIB=$PWD/groth16_amd/libg16_asan.so G16_ORACLE_LIB=$PWD/oracle/libg16_oracle_asan.so
    # python itself is not instrumented: the runtime comes in through LD_PRELOAD; leaks of the interpreter are not ours to report
    export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1:detect_stack_use_after_return=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
    LD_PRELOAD=$RT python -m pytest tests -q -x -m "not gpu and not perf" -p no:cacheprovider "$@" ;;
  *) echo "usage: $0 build | test [pytest args]"; exit 2 ;;
esac
