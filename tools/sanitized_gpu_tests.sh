#!/bin/bash
# The stage planners under sanitizers ON THE GPU BOX: the host side of every translation unit instrumented
# (-fno-gpu-sanitize: the device code is the product's, GPU ASan is not available on this pool), the -m gpu tests of the SDP
# stage, the k-band refiner, the refine planner and the mapper through BLASR_HIP_LIBRARY.  What the CPU pass
# (tools/sanitized_cpu_tests.sh) cannot reach runs here: sdp_api.hip's level loop and arena, kband_api.hip's chunk / lane
# plan, mapread_api.hip's retry loops, the refine planner.
#   1. AddressSanitizer + UBSan (make SAN=1): the HIP runtime must start under the preloaded ASan runtime -- a two-line
#      probe says whether it does on this box (on this image hsa_init dies under it: the probe's report is kept);
#   2. UndefinedBehaviorSanitizer alone (make SAN=ubsan): needs no preload, always runs.
#   usage (repo root; the _san / _ubsan libraries built beforehand and not listed in .gpurunignore):
#     gpurun --timeout 2400 -- 'bash tools/sanitized_gpu_tests.sh gpurun_out/<dir> [test files]'
ROOT=$(pwd)
OUT=$ROOT/${1:-gpurun_out/san}
shift
mkdir -p "$OUT"
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
ls -la "$ROOT"/blasr_amd/libblasr_hip_san.so "$ROOT"/blasr_amd/libblasr_hip_ubsan.so "$RT" > "$OUT/san_env.txt" 2>&1
TESTS=${@:-tests/test_sdp_gpu.py tests/test_kband_gpu.py tests/test_mapread_gpu.py tests/test_refine_gpu.py}
SEL='not two_ranks and not host_memory_runs_out'
AOPT=detect_leaks=0:abort_on_error=1:allocator_may_return_null=1:protect_shadow_gap=0:use_sigaltstack=0
# ---- 1. ASan: does a context come up at all?
if [ -f "$ROOT/blasr_amd/libblasr_hip_san.so" ]; then
  LD_PRELOAD="$RT" ASAN_OPTIONS=$AOPT BLASR_HIP_LIBRARY="$ROOT/blasr_amd/libblasr_hip_san.so" timeout 300 python3 -X faulthandler=0 -c "
import blasr_amd
c = blasr_amd.Context(0)
print('context under ASan: ok')
c.close()" > "$OUT/asan_probe.log" 2>&1
  echo "asan probe rc $?" | tee -a "$OUT/asan_probe.log"
  if grep -q "context under ASan: ok" "$OUT/asan_probe.log"; then
    LD_PRELOAD="$RT" ASAN_OPTIONS=$AOPT UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
      BLASR_HIP_LIBRARY="$ROOT/blasr_amd/libblasr_hip_san.so" \
      timeout 2000 python3 -m pytest $TESTS -x -q -m gpu -p no:faulthandler -k "$SEL" > "$OUT/asan_pytest.log" 2>&1
    echo "asan + ubsan gpu tests rc $?" | tee -a "$OUT/asan_pytest.log"
    tail -5 "$OUT/asan_pytest.log"
  else
    tail -25 "$OUT/asan_probe.log"
  fi
fi
# ---- 2. UBSan alone
#      (rocPRIM's own headers add an offset to a null pointer when asked for their scratch size -- device_segmented_radix_sort.hpp:211,
#       instantiated in anchor_api.hip: for the anchoring tests the library is built with that one check recoverable,
#         make -C blasr_amd/csrc SAN=ubsan CXXFLAGS="-O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -fno-omit-frame-pointer \
#              -fsanitize=undefined -fno-gpu-sanitize -fno-sanitize-recover=undefined -fsanitize-recover=pointer-overflow"
#       and every "runtime error" line that does not name a rocprim header fails the pass; -s: a report must not end in pytest's capture)
if [ -f "$ROOT/blasr_amd/libblasr_hip_ubsan.so" ]; then
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0 BLASR_HIP_LIBRARY="$ROOT/blasr_amd/libblasr_hip_ubsan.so" \
    timeout 2000 python3 -m pytest $TESTS -x -q -s -m gpu -k "$SEL" > "$OUT/ubsan_pytest.log" 2>&1
  echo "ubsan gpu tests rc $?" | tee -a "$OUT/ubsan_pytest.log"
  grep "runtime error" "$OUT/ubsan_pytest.log" | sort | uniq -c > "$OUT/ubsan_reports.txt"
  echo "reports outside rocprim: $(grep -v "include/rocprim/" "$OUT/ubsan_reports.txt" | wc -l)" | tee -a "$OUT/ubsan_pytest.log"
  cat "$OUT/ubsan_reports.txt" | cut -c1-240
  grep -E "passed|failed" "$OUT/ubsan_pytest.log" | tail -2
fi
