// Wavefront-level helpers shared by the gfx950 kernels (wave64, DPP cross-lane operations).
#ifndef BLASR_AMD_WAVE_UTILS_H_
#define BLASR_AMD_WAVE_UTILS_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace blasr_hip {

// ---------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false);
}
// lane i receives lane i-1; lane 0 receives `edge` (DPP wave_shr:1)
__device__ __forceinline__ int wave_shr1(int v, int edge) { return dpp<0x138, 0xf>(edge, v); }

// inclusive prefix-min over the 64 lanes: row_shr 1,2,4,8 then row_bcast15/31
__device__ __forceinline__ int wave_prefix_min(int v) {
  v = min(v, dpp<0x111, 0xf>(v, v));
  v = min(v, dpp<0x112, 0xf>(v, v));
  v = min(v, dpp<0x114, 0xf>(v, v));
  v = min(v, dpp<0x118, 0xf>(v, v));
  v = min(v, dpp<0x142, 0xa>(v, v));
  v = min(v, dpp<0x143, 0xc>(v, v));
  return v;
}
__device__ __forceinline__ int wave_prefix_max(int v) {
  v = max(v, dpp<0x111, 0xf>(v, v));
  v = max(v, dpp<0x112, 0xf>(v, v));
  v = max(v, dpp<0x114, 0xf>(v, v));
  v = max(v, dpp<0x118, 0xf>(v, v));
  v = max(v, dpp<0x142, 0xa>(v, v));
  v = max(v, dpp<0x143, 0xc>(v, v));
  return v;
}
// inclusive prefix-sum (Hillis-Steele needs the shifted value to be 0 where no source lane exists)
__device__ __forceinline__ unsigned wave_prefix_sum(unsigned v) {
  v += (unsigned)dpp<0x111, 0xf>(0, (int)v);
  v += (unsigned)dpp<0x112, 0xf>(0, (int)v);
  v += (unsigned)dpp<0x114, 0xf>(0, (int)v);
  v += (unsigned)dpp<0x118, 0xf>(0, (int)v);
  v += (unsigned)dpp<0x142, 0xa>(0, (int)v);
  v += (unsigned)dpp<0x143, 0xc>(0, (int)v);
  return v;
}
__device__ __forceinline__ int wave_max(int v) { return __builtin_amdgcn_readlane(wave_prefix_max(v), 63); }

// NucConversion.h:48-86 (ThreeBit) on the supported alphabet: ACGT -> 0..3, else N = 4
__device__ __forceinline__ int base_code(unsigned c) {
  c &= 0xdfu;  // fold case
  int code = 4;
  code = (c == 'A') ? 0 : code;
  code = (c == 'C') ? 1 : code;
  code = (c == 'G') ? 2 : code;
  code = (c == 'T') ? 3 : code;
  return code;
}

}  // namespace blasr_hip
#endif
