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

// The substitution scores and the match / mismatch counts of `len` aligned columns of one lane (ComputeAlignmentScore's
// inner loop, AlignmentUtils.h:60-124): four columns per pair of (unaligned) dword loads, the codes from a 256-entry table in
// LDS (lut[c] = base_code(c)), the scores from the 5 x 5 matrix in LDS.  ThreeBit equality: N against N counts as a match.
__device__ __forceinline__ void score_columns(const uint8_t *qp, const uint8_t *tp, int len, const int *matrixLds,
                                              const uint8_t *lutLds, int &score, int &nMatch, int &nMismatch) {
  int j = 0, same = 0;
  for (; j + 4 <= len; j += 4) {
    uint32_t qw, tw;
    __builtin_memcpy(&qw, qp + j, 4);
    __builtin_memcpy(&tw, tp + j, 4);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int qc = lutLds[(qw >> (8 * k)) & 0xffu], tc = lutLds[(tw >> (8 * k)) & 0xffu];
      score += matrixLds[qc * 5 + tc];
      same += qc == tc ? 1 : 0;
    }
  }
  for (; j < len; j++) {
    const int qc = lutLds[qp[j]], tc = lutLds[tp[j]];
    score += matrixLds[qc * 5 + tc];
    same += qc == tc ? 1 : 0;
  }
  nMatch += same;
  nMismatch += max(len, 0) - same;
}

}  // namespace blasr_hip
#endif
