#!/bin/bash
# The CPU test suite against sanitizer builds: the oracle's C restatement and the HOST side of every
# translation unit of the library (file readers, printers, interval finder, filters, the ABI's argument checks, the worker
# threads) under AddressSanitizer + UndefinedBehaviorSanitizer.  Device code is not instrumented (GPU ASan is not available
# on this pool); nothing here needs a GPU.        usage (repo root):  bash tools/sanitized_cpu_tests.sh [pytest args]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -s -C "$ROOT/blasr_amd/csrc" SAN=1 -j6
make -s -C "$ROOT/oracle" san
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd "$ROOT"
# (the interpreter itself is not leak-clean; a report of either sanitizer in OUR code ends the run: abort_on_error / halt_on_error.
#  The RLIMIT_AS test cannot run under ASan, whose shadow memory alone is terabytes of address space.)
LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:allocator_may_return_null=1 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  BLASR_HIP_LIBRARY="$ROOT/blasr_amd/libblasr_hip_san.so" BLASR_ORACLE_SO="$ROOT/oracle/libblasr_oracle_san.so" \
  python3 -m pytest tests -x -q -m "not gpu" -k "not host_memory_runs_out" "$@"
