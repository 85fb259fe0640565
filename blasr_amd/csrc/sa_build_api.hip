// next-4: the suffix array of the genome and its lookup table, built on the device.
//
// What the reference does at start-up when no .sa file is given (alignment/Blasr.cpp:4407-4421), and what its sawriter
// tool writes to disk (alignment/SAWriter.cpp:187-232): genome.ToThreeBit(), LarssonBuildSuffixArray
// (datastructures/suffixarray/SuffixArray.h:256-269: Larsson-Sadakane over the codes + 1 with a 0 terminator, i.e. the
// suffixes in lexicographic order under A < C < G < T < N with a shorter suffix before any longer one it prefixes),
// BuildLookupTable (:193-249), Write (:505-532).  The suffix array of a string is unique, so any construction gives
// the reference's array; here it is prefix doubling, which is all radix sorts and gathers and sized for the HBM of
// one MI355X (32 bytes per base while building: 99 GB for hg19, next to the genome):
//   sa_first_keys_kernel    the first 21 symbols of every suffix, 3 bits each (0 = past the end), as a 64-bit key
//   (rocprim radix sort)    suffixes by key
//   sa_heads_kernel + max-scan + sa_rank_kernel   rank of a suffix = 1 + the array position of the first suffix with
//                           the same key (what the h symbols seen so far cannot tell apart shares a rank)
//   sa_round_keys_kernel    key = (rank of suffix i, rank of suffix i + h): doubles the symbols compared; repeat until
//                           every rank is unique (log2(longest repeat / 21) rounds)
// The lookup table's ranges are a sequential walk over the array in the reference, with quirks that change the table
// (a suffix with an N among its first bases counts as the all-A word while a range is extended; the range of the
// last word seen is overwritten when the array ends in such suffixes): the device classifies every array position
// (word, valid / too short / ...), the host replays the walk over those 4 bytes per position.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/blasr_hip.h"
#include "ctx.h"

namespace {

// ThreeBit's order (NucConversion.h:48-84) plus one, 0 = past the end; what is not ACGT sorts as N
__device__ __forceinline__ unsigned sym_rank(uint8_t c) {
  switch (c) {
    case 'A': case 'a': case 0: return 1;
    case 'C': case 'c': case 1: return 2;
    case 'G': case 'g': case 2: return 3;
    case 'T': case 't': case 3: return 4;
    default: return 5;
  }
}

constexpr int kFirst = 21;  // symbols in the first key

__global__ __launch_bounds__(256) void sa_first_keys_kernel(const uint8_t *genome, int64_t n, uint64_t *keys,
                                                            uint32_t *sa) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t key = 0;
  for (int k = 0; k < kFirst; k++) key = (key << 3) | (i + k < n ? sym_rank(genome[i + k]) : 0u);
  keys[i] = key;
  sa[i] = (uint32_t)i;
}

constexpr int kGroupCounters = 4096;

// heads[j] = j where a new key starts, else 0; counts the groups
__global__ __launch_bounds__(256) void sa_heads_kernel(const uint64_t *keys, int64_t n, uint32_t *heads,
                                                       unsigned long long *groups) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  bool head = false;
  if (j < n) {
    head = j == 0 || keys[j] != keys[j - 1];
    heads[j] = head ? (uint32_t)j : 0u;
  }
  // (one counter would take 48 M atomics on one address for an hg19-size genome: kGroupCounters of them, by block)
  const unsigned long long m = __ballot(head);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&groups[blockIdx.x & (kGroupCounters - 1)], (unsigned long long)__popcll(m));
}

__global__ __launch_bounds__(256) void sa_rank_kernel(const uint32_t *sa, const uint32_t *groupStart, int64_t n,
                                                      uint32_t *rank) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < n) rank[sa[j]] = groupStart[j] + 1u;
}

__global__ __launch_bounds__(256) void sa_round_keys_kernel(const uint32_t *sa, const uint32_t *rank, int64_t h, int64_t n,
                                                            uint64_t *keys) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int64_t i = sa[j];
  const uint32_t second = i + h < n ? rank[i + h] : 0u;
  keys[j] = ((uint64_t)rank[i] << 32) | second;
}

// per array position: the word of the suffix's first `p` bases (bits 0-27, 0 unless every base is ACGT) and what
// BuildLookupTable's tests make of the suffix (bits 28-30):
//   0 all ACGT, index + p < L     1 not all ACGT, index + p < L     2 index + p == L, not all ACGT (it holds the last
//   base of the genome, an N as Blasr lays a genome out)     3 index + p > L     4 index + p == L and all ACGT
__global__ __launch_bounds__(256) void sa_lookup_info_kernel(const uint8_t *genome, int64_t L, const uint32_t *sa, int p,
                                                             int64_t n, uint32_t *info) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int64_t i = sa[j];
  uint32_t word = 0;
  bool ok = true;
  for (int k = 0; k < p; k++) {
    const unsigned s = i + k < L ? sym_rank(genome[i + k]) : 0u;
    ok = ok && s >= 1 && s <= 4;
    word = (word << 2) | ((s - 1) & 3u);
  }
  uint32_t cls;
  if (i + p > L)
    cls = 3;
  else if (i + p == L)
    cls = ok ? 4 : 2;
  else
    cls = ok ? 0 : 1;
  info[j] = (ok && cls != 3 ? word : 0u) | (cls << 28);
}

// TupleCountTable::AddSequenceTupleCountsLR (datastructures/tuplelists/TupleCountTable.h:63-73) over the resident genome:
// every position whose `k` bases are all ACGT counts its word.  Words are spread over 4^k counters, so the atomics of a
// wavefront rarely meet.
__global__ __launch_bounds__(256) void count_tuples_kernel(const uint8_t *genome, int64_t n, int k, int32_t *table) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i + k > n) return;
  uint32_t word = 0;
  for (int j = 0; j < k; j++) {
    const unsigned s = sym_rank(genome[i + j]);
    if (s < 1 || s > 4) return;
    word = (word << 2) | (s - 1);
  }
  atomicAdd(&table[word], 1);
}

struct Dev {
  void *p = nullptr;
  ~Dev() {
    if (p) (void)hipFree(p);
  }
};

}  // namespace

extern "C" {

int blasr_hip_build_suffix_array(blasr_hip_ctx *ctx, int32_t lookup_prefix_length, uint32_t *index_out,
                                 uint32_t *start_pos_out, uint32_t *end_pos_out) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (!ctx->genome) return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "build_suffix_array: no genome uploaded");
  if (lookup_prefix_length < 0 || lookup_prefix_length > 14)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "build_suffix_array: lookup prefix outside 0..14");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int64_t n = ctx->genomeLen;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  Dev keysA, keysB, saA, saB, rank, heads, tmp, counter;
  HIP_TRY(ctx, hipMalloc(&keysA.p, 8 * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&keysB.p, 8 * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&saA.p, 4 * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&saB.p, 4 * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&rank.p, 4 * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&heads.p, 4 * (size_t)n));
  HIP_TRY(ctx, hipMalloc(&counter.p, 8 * kGroupCounters));
  rocprim::double_buffer<uint64_t> keys((uint64_t *)keysA.p, (uint64_t *)keysB.p);
  rocprim::double_buffer<uint32_t> sa((uint32_t *)saA.p, (uint32_t *)saB.p);
  size_t sortBytes = 0, scanBytes = 0;
  HIP_TRY(ctx, rocprim::radix_sort_pairs(nullptr, sortBytes, keys, sa, (size_t)n, 0, 64, st));
  HIP_TRY(ctx, rocprim::inclusive_scan(nullptr, scanBytes, (uint32_t *)heads.p, (uint32_t *)heads.p, (size_t)n,
                                       rocprim::maximum<uint32_t>(), st));
  HIP_TRY(ctx, hipMalloc(&tmp.p, std::max(sortBytes, scanBytes) + 256));

  hipLaunchKernelGGL(sa_first_keys_kernel, dim3(blocks), dim3(256), 0, st, ctx->genome, n, keys.current(), sa.current());
  int64_t h = kFirst;
  int rounds = 0;
  while (true) {
    HIP_TRY(ctx, rocprim::radix_sort_pairs(tmp.p, sortBytes, keys, sa, (size_t)n, 0, rounds == 0 ? 3 * kFirst : 64, st));
    HIP_TRY(ctx, hipMemsetAsync(counter.p, 0, 8 * kGroupCounters, st));
    hipLaunchKernelGGL(sa_heads_kernel, dim3(blocks), dim3(256), 0, st, keys.current(), n, (uint32_t *)heads.p,
                       (unsigned long long *)counter.p);
    std::vector<unsigned long long> partial(kGroupCounters);
    HIP_TRY(ctx, hipMemcpyAsync(partial.data(), counter.p, 8 * kGroupCounters, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    unsigned long long groups = 0;
    for (unsigned long long v : partial) groups += v;
    if ((int64_t)groups == n) break;
    if (++rounds > 40) return blasr_hip_fail(ctx, BLASR_HIP_ERR_HIP, "build_suffix_array: ranks did not separate");
    HIP_TRY(ctx, rocprim::inclusive_scan(tmp.p, scanBytes, (uint32_t *)heads.p, (uint32_t *)heads.p, (size_t)n,
                                         rocprim::maximum<uint32_t>(), st));
    hipLaunchKernelGGL(sa_rank_kernel, dim3(blocks), dim3(256), 0, st, sa.current(), (uint32_t *)heads.p, n,
                       (uint32_t *)rank.p);
    hipLaunchKernelGGL(sa_round_keys_kernel, dim3(blocks), dim3(256), 0, st, sa.current(), (uint32_t *)rank.p, h, n,
                       keys.current());
    h *= 2;
  }

  // ---- install: the index array stays resident for the anchor stage
  blasr_hip_drop_index(ctx);
  HIP_TRY(ctx, hipMalloc(&ctx->saIndex, 4 * (size_t)n));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->saIndex, sa.current(), 4 * (size_t)n, hipMemcpyDeviceToDevice, st));
  ctx->saLen = n;
  if (index_out) HIP_TRY(ctx, hipMemcpyAsync(index_out, sa.current(), 4 * (size_t)n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  if (lookup_prefix_length == 0) return BLASR_HIP_OK;

  // ---- BuildLookupTable (SuffixArray.h:193-249): the device classifies, the host walks
  const int p = lookup_prefix_length;
  const int64_t tableLen = 1ll << (2 * p);
  std::vector<uint32_t> startPos((size_t)tableLen, 0), endPos((size_t)tableLen, 0);
  const int64_t L = n;
  if (L >= p) {
    const int64_t limit = L - p + 1;
    uint32_t *dInfo = (uint32_t *)heads.p;  // (free again)
    hipLaunchKernelGGL(sa_lookup_info_kernel, dim3((unsigned)((limit + 255) / 256)), dim3(256), 0, st, ctx->genome, L,
                       ctx->saIndex, p, limit, dInfo);
    std::vector<uint32_t> info((size_t)limit);
    HIP_TRY(ctx, hipMemcpyAsync(info.data(), dInfo, 4 * (size_t)limit, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    const uint32_t wordMask = (1u << 28) - 1;
    auto cls = [&](int64_t q) { return info[(size_t)q] >> 28; };
    uint32_t cur = 0;  // curPrefix.tuple (a default-constructed tuple is 0)
    int64_t pos = 0;
    do {
      while (pos < limit && cls(pos) == 3) pos++;  // :220-223, index + prefix > length
      if (pos >= limit) break;
      // :227-230, until FromStringLR succeeds.  When this runs off the end, the range of the word parsed last is
      // overwritten below with the end of the array -- as in the reference.
      while (pos < limit && !(cls(pos) == 0 || cls(pos) == 4)) pos++;
      if (pos < limit) cur = info[(size_t)pos] & wordMask;
      startPos[cur] = (uint32_t)pos;
      pos++;
      while (pos < limit && cls(pos) <= 1) {  // :234-244, index + prefix < length; a failed parse leaves the word 0
        const uint32_t next = cls(pos) == 0 ? (info[(size_t)pos] & wordMask) : 0u;
        if (next != cur) break;
        pos++;
      }
      endPos[cur] = (uint32_t)pos;
    } while (pos < limit && (int64_t)cur + 1 < tableLen);
  }
  HIP_TRY(ctx, hipMalloc(&ctx->saStartPos, 4 * (size_t)tableLen));
  HIP_TRY(ctx, hipMalloc(&ctx->saEndPos, 4 * (size_t)tableLen));
  HIP_TRY(ctx, hipMemcpy(ctx->saStartPos, startPos.data(), 4 * (size_t)tableLen, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ctx->saEndPos, endPos.data(), 4 * (size_t)tableLen, hipMemcpyHostToDevice));
  ctx->saTableLen = tableLen;
  ctx->saPrefixLen = p;
  if (start_pos_out) memcpy(start_pos_out, startPos.data(), 4 * (size_t)tableLen);
  if (end_pos_out) memcpy(end_pos_out, endPos.data(), 4 * (size_t)tableLen);
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CTX(ctx)

int blasr_hip_count_tuples_device(blasr_hip_ctx *ctx, int32_t tuple_size, int32_t *count_table) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (!ctx->genome || tuple_size < 1 || tuple_size > 14 || !count_table)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "count_tuples_device: no genome, or bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int64_t entries = 1ll << (2 * tuple_size);
  Dev table;
  HIP_TRY(ctx, hipMalloc(&table.p, 4 * (size_t)entries));
  HIP_TRY(ctx, hipMemsetAsync(table.p, 0, 4 * (size_t)entries, ctx->stream));
  hipLaunchKernelGGL(count_tuples_kernel, dim3((unsigned)((ctx->genomeLen + 255) / 256)), dim3(256), 0, ctx->stream,
                     ctx->genome, ctx->genomeLen, tuple_size, (int32_t *)table.p);
  HIP_TRY(ctx, hipMemcpyAsync(count_table, table.p, 4 * (size_t)entries, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CTX(ctx)

// SuffixArray::Write (SuffixArray.h:505-532) with the component list sawriter writes: the array and the lookup table
int blasr_hip_sa_write(const char *path, const uint32_t *index, int64_t length, const uint32_t *start_pos_table,
                       const uint32_t *end_pos_table, int64_t lookup_table_length, int32_t lookup_prefix_length) try {
  if (!path || !index || length <= 0 || length >= (1ll << 32) || lookup_table_length < 0 ||
      (lookup_table_length > 0 && (!start_pos_table || !end_pos_table)))
    return BLASR_HIP_ERR_INVALID;
  FILE *f = fopen(path, "wb");
  if (!f) return BLASR_HIP_ERR_INVALID;
  const uint32_t magic = 0xacac0001u;
  const int32_t comp[2] = {1, lookup_table_length > 0 ? 1 : 0};
  const uint32_t len = (uint32_t)length, tableLen = (uint32_t)lookup_table_length, prefix = (uint32_t)lookup_prefix_length;
  bool ok = fwrite(&magic, 4, 1, f) == 1 && fwrite(comp, 4, 2, f) == 2 && fwrite(&len, 4, 1, f) == 1 &&
            fwrite(index, 4, (size_t)length, f) == (size_t)length;
  if (ok && comp[1])
    ok = fwrite(&tableLen, 4, 1, f) == 1 && fwrite(&prefix, 4, 1, f) == 1 &&
         fwrite(start_pos_table, 4, (size_t)tableLen, f) == (size_t)tableLen &&
         fwrite(end_pos_table, 4, (size_t)tableLen, f) == (size_t)tableLen;
  ok = fclose(f) == 0 && ok;
  return ok ? BLASR_HIP_OK : BLASR_HIP_ERR_INVALID;
} BLASR_HIP_CATCH_CODE

// The same file from the index that is resident on the context's device (blasr_hip_build_suffix_array /
// blasr_hip_upload_suffix_array left it there): the array comes down in pieces of 256 MB and goes straight to the file,
// so a 3.1 Gbase genome's 12.4 GB index never needs a host array of its own.
int blasr_hip_sa_write_resident(blasr_hip_ctx *ctx, const char *path) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (!path) return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "sa_write_resident: no path");
  if (!ctx->saIndex || ctx->saLen <= 0)
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "sa_write_resident: no suffix array resident on this context");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  FILE *f = fopen(path, "wb");
  if (!f) return blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "sa_write_resident: cannot open %s", path);
  const bool table = ctx->saTableLen > 0 && ctx->saStartPos && ctx->saEndPos;
  const uint32_t magic = 0xacac0001u;
  const int32_t comp[2] = {1, table ? 1 : 0};
  const uint32_t len = (uint32_t)ctx->saLen, tableLen = (uint32_t)ctx->saTableLen, prefix = (uint32_t)ctx->saPrefixLen;
  bool ok = fwrite(&magic, 4, 1, f) == 1 && fwrite(comp, 4, 2, f) == 2 && fwrite(&len, 4, 1, f) == 1;
  const size_t piece = (size_t)64 << 20;  // elements
  std::vector<uint32_t> buf(std::min<size_t>(piece, (size_t)std::max<int64_t>(ctx->saLen, ctx->saTableLen)));
  hipError_t he = hipSuccess;
  auto stream_out = [&](const uint32_t *dev, size_t n) {
    for (size_t at = 0; at < n && ok && he == hipSuccess; at += buf.size()) {
      const size_t m = std::min(buf.size(), n - at);
      he = hipMemcpy(buf.data(), dev + at, 4 * m, hipMemcpyDeviceToHost);
      if (he == hipSuccess) ok = fwrite(buf.data(), 4, m, f) == m;
    }
  };
  stream_out(ctx->saIndex, (size_t)ctx->saLen);
  if (ok && table && he == hipSuccess) {
    ok = fwrite(&tableLen, 4, 1, f) == 1 && fwrite(&prefix, 4, 1, f) == 1;
    stream_out(ctx->saStartPos, (size_t)tableLen);
    stream_out(ctx->saEndPos, (size_t)tableLen);
  }
  ok = fclose(f) == 0 && ok;
  if (he != hipSuccess) return blasr_hip_fail(ctx, BLASR_HIP_ERR_HIP, "sa_write_resident: %s", hipGetErrorString(he));
  return ok ? BLASR_HIP_OK : blasr_hip_fail(ctx, BLASR_HIP_ERR_INVALID, "sa_write_resident: writing %s failed", path);
} BLASR_HIP_CATCH_CTX(ctx)

}  // extern "C"
