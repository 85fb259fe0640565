#!/bin/bash
# The stage planners under AddressSanitizer + UndefinedBehaviorSanitizer ON THE GPU BOX: the host side of every translation
# unit instrumented (make SAN=1: -fsanitize=address,undefined -fno-gpu-sanitize; the device code is the product's, GPU ASan
# is not available on this pool), the -m gpu tests of the SDP stage, the k-band refiner and the mapper through
# BLASR_HIP_LIBRARY.  What the CPU pass (tools/sanitized_cpu_tests.sh) cannot reach runs here: sdp_api.hip's level loop and
# arena, kband_api.hip's chunk / lane plan, mapread_api.hip's retry loops, the refine planner.
#   usage (repo root, the _san.so built beforehand and NOT listed in .gpurunignore):
#     gpurun --timeout 2400 -- 'bash tools/sanitized_gpu_tests.sh gpurun_out/<dir> [pytest args]'
ROOT=$(pwd)
OUT=$ROOT/${1:-gpurun_out/san}
shift
mkdir -p "$OUT"
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
ls -la "$ROOT/blasr_amd/libblasr_hip_san.so" "$RT" > "$OUT/san_env.txt" 2>&1
TESTS=${@:-tests/test_sdp_gpu.py tests/test_kband_gpu.py tests/test_mapread_gpu.py tests/test_refine_gpu.py}
# (detect_leaks=0: the interpreter and the HIP runtime are not leak-clean; a report in OUR code ends the run)
LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:allocator_may_return_null=1:protect_shadow_gap=0 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  BLASR_HIP_LIBRARY="$ROOT/blasr_amd/libblasr_hip_san.so" \
  timeout 2000 python3 -m pytest $TESTS -x -q -m gpu -k "not two_ranks and not host_memory_runs_out" > "$OUT/san_pytest.log" 2>&1
echo "sanitized gpu tests rc $?" | tee -a "$OUT/san_pytest.log"
tail -15 "$OUT/san_pytest.log"
