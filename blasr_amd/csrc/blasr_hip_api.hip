// blasr_hip_api.hip -- the extern "C" boundary declared in include/blasr_hip.h.
// Host-side only: context, grow-only device workspace, batch planning and the
// kernel sequence of one refine call.  There is deliberately no CPU path here:
// without a gfx950 device every entry point returns BLASR_HIP_ERR_NO_DEVICE.
#include <chrono>
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/blasr_hip.h"
#include "ctx.h"
#include "refine_kernels.h"

using namespace blasr_hip;

static_assert(sizeof(RefineResult) == sizeof(blasr_hip_refine_result), "result record layout");
static_assert(offsetof(RefineResult, ops_off) == offsetof(blasr_hip_refine_result, ops_off), "result record layout");

int blasr_hip_fail(blasr_hip_ctx *ctx, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) {
    try {
      ctx->lastError = buf;
    } catch (...) {  // (the message needs memory too; the code still goes back)
    }
  }
  return code;
}
#define fail blasr_hip_fail

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// What this context's workspace may grow to: the caller's limit, and what the device can give it -- its own
// allocation plus what is free, less a fixed reserve (so that the answer does not creep from call to call).
int64_t workspace_reach(const blasr_hip_ctx *ctx) {
  int64_t limit = ctx->workspaceLimit;
  size_t freeB = 0, totalB = 0;
  if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
    const int64_t reserve = std::max<int64_t>((int64_t)2 << 30, (int64_t)(totalB / 32));
    limit = std::min<int64_t>(limit, std::max<int64_t>((int64_t)ctx->workspaceBytes + (int64_t)freeB - reserve, 1 << 20));
  } else {
    (void)hipGetLastError();
  }
  return limit;
}

int ensure_workspace(blasr_hip_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->workspaceBytes) return BLASR_HIP_OK;
  const int64_t reach = workspace_reach(ctx);
  if ((int64_t)bytes > reach)
    return fail(ctx, BLASR_HIP_ERR_NOMEM, "workspace of %zu bytes exceeds the limit of %lld", bytes, (long long)reach);
  if (ctx->workspace) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(ctx->workspace));
    ctx->workspace = nullptr;
    ctx->workspaceBytes = 0;
  }
  // grow with head-room so that steady-state batches do not reallocate
  size_t want = std::min<size_t>((size_t)reach, bytes + bytes / 8);
  if (hipMalloc((void **)&ctx->workspace, want) != hipSuccess) {
    (void)hipGetLastError();
    want = bytes;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->workspace, want));
  }
  ctx->workspaceBytes = want;
  return BLASR_HIP_OK;
}

// the scan formulation of the deletion dependency (refine_kernels.hip) needs
// open >= 0 and del >= ext >= 0; scores must stay far from the kInf sentinel
int check_score_domain(blasr_hip_ctx *ctx, const blasr_hip_score_params *sc, int64_t longest) {
  if (sc->affine_open < 0 || sc->affine_extend < 0 || sc->del < sc->affine_extend)
    return fail(ctx, BLASR_HIP_ERR_UNSUPPORTED,
                "unsupported gap costs: need affine_open >= 0 and del >= affine_extend >= 0 (got open=%d ext=%d del=%d)",
                sc->affine_open, sc->affine_extend, sc->del);
  int64_t mx = std::max<int64_t>({llabs((long long)sc->ins), llabs((long long)sc->del),
                                  llabs((long long)sc->affine_open), llabs((long long)sc->affine_extend)});
  for (int i = 0; i < 25; i++) mx = std::max<int64_t>(mx, llabs((long long)sc->matrix[i]));
  if (mx * (longest + 66) >= (1ll << 27))
    return fail(ctx, BLASR_HIP_ERR_UNSUPPORTED, "score magnitude %lld x length %lld overflows the 32-bit DP range",
                (long long)mx, (long long)longest);
  return BLASR_HIP_OK;
}

struct Plan {
  std::vector<int64_t> arrowOff, rowOff, arenaOff;
  struct Chunk {
    int first, count;
    size_t bytes;
    std::vector<int32_t> wideTasks;
    std::vector<int64_t> wideRingOff;
    std::vector<int32_t> wideRingSize;
    size_t wideBytes;
  };
  std::vector<Chunk> chunks;
};

}  // namespace

int64_t blasr_hip_workspace_reach(const blasr_hip_ctx *ctx) { return workspace_reach(ctx); }
int blasr_hip_grow_workspace(blasr_hip_ctx *ctx, size_t bytes) { return ensure_workspace(ctx, bytes); }

int blasr_hip_ensure(blasr_hip_ctx *ctx, DevBuf &b, size_t bytes) {
  if (bytes <= b.bytes) return BLASR_HIP_OK;
  if (b.p) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
  }
  const size_t want = bytes + bytes / 8 + 256;  // head-room: steady-state batches do not reallocate
  if (hipMalloc((void **)&b.p, want) == hipSuccess) {
    b.bytes = want;
    return BLASR_HIP_OK;
  }
  (void)hipGetLastError();
  HIP_TRY(ctx, hipMalloc((void **)&b.p, bytes));
  b.bytes = bytes;
  return BLASR_HIP_OK;
}

extern "C" {

int blasr_hip_abi_version(void) { return BLASR_HIP_ABI_VERSION; }

int blasr_hip_create(int device, blasr_hip_ctx **out) try {
  if (!out) return BLASR_HIP_ERR_INVALID;
  *out = nullptr;
  // The sweep of an SDP level runs its four kernels side by side on four streams (sdp_kernels.hip: launch_sdp_chain), and
  // the runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless told otherwise, read once when the runtime
  // starts): with four, two of the kernels share a queue and run one after the other (10 ms per configs[2] step).  The
  // library does not touch the process environment: the HOST sets GPU_MAX_HW_QUEUES=8 before its first HIP call
  // (INTEGRATION.md 4'; bench.py, blasr_amd/__init__.py, examples/map_fasta.cpp and the drop-in's main do).  Say so once
  // where it was not done.
  {
    static std::once_flag queuesOnce;
    std::call_once(queuesOnce, []() {
      const char *q = getenv("GPU_MAX_HW_QUEUES");
      if ((!q || atoi(q) < 8) && getenv("BLASR_HIP_SDP_TRACE"))
        fprintf(stderr, "[blasr_hip] GPU_MAX_HW_QUEUES is %s: the SDP sweep's four kernels will share hardware queues "
                        "(set it to 8 before the first HIP call)\n", q ? q : "unset");
    });
  }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) {
    (void)hipGetLastError();
    return BLASR_HIP_ERR_NO_DEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return BLASR_HIP_ERR_NO_DEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return BLASR_HIP_ERR_NO_DEVICE;  // kernels are gfx950-only
  blasr_hip_ctx *ctx = new blasr_hip_ctx();
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->sideStream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return BLASR_HIP_ERR_HIP;
  }
  for (auto &e : ctx->events)
    if (hipEventCreate(&e) != hipSuccess) {
      blasr_hip_destroy(ctx);
      return BLASR_HIP_ERR_HIP;
    }
  for (auto &s : ctx->auxStream)
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
      blasr_hip_destroy(ctx);
      return BLASR_HIP_ERR_HIP;
    }
  for (auto &e : ctx->auxEvent)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      blasr_hip_destroy(ctx);
      return BLASR_HIP_ERR_HIP;
    }
  *out = ctx;
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_destroy(blasr_hip_ctx *ctx) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamDestroy(ctx->stream);
  }
  if (ctx->sideStream) {
    (void)hipStreamSynchronize(ctx->sideStream);
    (void)hipStreamDestroy(ctx->sideStream);
  }
  for (auto &e : ctx->events)
    if (e) (void)hipEventDestroy(e);
  for (auto &s : ctx->auxStream)
    if (s) {
      (void)hipStreamSynchronize(s);
      (void)hipStreamDestroy(s);
    }
  for (auto &e : ctx->auxEvent)
    if (e) (void)hipEventDestroy(e);
  for (auto &e : ctx->sdpEventPool)
    if (e) (void)hipEventDestroy(e);
  if (ctx->workspace) (void)hipFree(ctx->workspace);
  if (ctx->genome && ctx->genomeOwned) (void)hipFree(ctx->genome);
  if (ctx->contigStarts) (void)hipFree(ctx->contigStarts);
  for (auto &b : ctx->mapBuf)
    if (b.p) (void)hipFree(b.p);
  for (auto &b : ctx->anchorBuf)
    if (b.p) (void)hipFree(b.p);
  for (auto &b : ctx->mapHost)
    if (b.p) (void)hipHostFree(b.p);
  if (ctx->pipeHelper) (void)blasr_hip_destroy(ctx->pipeHelper);  // (it borrows this context's genome and index)
  ctx->pipeHelper = nullptr;
  blasr_hip_drop_index(ctx);
  if (ctx->anchorPinned) (void)hipHostFree(ctx->anchorPinned);
  if (ctx->pipePinned) (void)hipHostFree(ctx->pipePinned);
  if (ctx->pipeAnchors.p) (void)hipFree(ctx->pipeAnchors.p);
  if (ctx->pipeIntervals.p) (void)hipFree(ctx->pipeIntervals.p);
  if (ctx->anchorFlags) (void)hipFree(ctx->anchorFlags);
  delete ctx;
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CTX(ctx)

const char *blasr_hip_last_error(const blasr_hip_ctx *ctx) { return ctx ? ctx->lastError.c_str() : "null context"; }

int blasr_hip_set_workspace_limit(blasr_hip_ctx *ctx, int64_t bytes) try {
  if (!ctx || bytes < (1 << 20)) return BLASR_HIP_ERR_INVALID;
  ctx->workspaceLimit = bytes;
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CTX(ctx)

int64_t blasr_hip_held_bytes(const blasr_hip_ctx *ctx) try {
  if (!ctx) return 0;
  int64_t total = (int64_t)ctx->workspaceBytes;
  for (const auto &b : ctx->mapBuf) total += (int64_t)b.bytes;
  if (ctx->genome && ctx->genomeOwned) total += ctx->genomeLen;
  for (const auto &b : ctx->anchorBuf) total += (int64_t)b.bytes;
  total += (int64_t)ctx->pipeAnchors.bytes + (int64_t)ctx->pipeIntervals.bytes;
  if (ctx->saOwned) total += (ctx->saExt ? 6 : 4) * ctx->saLen + (ctx->saRanges ? 16 : 8) * ctx->saTableLen;
  if (ctx->pipeHelper) total += blasr_hip_held_bytes(ctx->pipeHelper);
  return total;
} BLASR_HIP_CATCH_CODE

void *blasr_hip_stream(blasr_hip_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int blasr_hip_refine_last_fill4(const blasr_hip_ctx *ctx, float *ms, int64_t *launches, int64_t *algorithmic_bytes,
                                int64_t *cells) try {
  if (!ctx || !ms || !launches || !algorithmic_bytes || !cells) return BLASR_HIP_ERR_INVALID;
  *ms = ctx->fill4Ms;
  *launches = ctx->fill4Launches;
  *algorithmic_bytes = ctx->fill4Bytes;
  *cells = ctx->fill4Cells;
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_refine_last_kernels(const blasr_hip_ctx *ctx, blasr_hip_kernel_record *out, int capacity) try {
  if (!ctx || (!out && capacity > 0) || capacity < 0) return BLASR_HIP_ERR_INVALID;
  int n = 0;
  for (int k = 0; k < BLASR_HIP_KERNEL_KINDS; k++) {
    if (ctx->kernelRec[k].launches <= 0) continue;
    if (n < capacity) out[n] = ctx->kernelRec[k];
    n++;
  }
  return n;
} BLASR_HIP_CATCH_CODE

const char *blasr_hip_kernel_name(int kind, int ext0) {
  switch (kind) {
    case BLASR_HIP_KERNEL_FILL_NARROW:
      return ext0 ? "blasr_hip::fillp_kernel<16, true, false>" : "blasr_hip::fillp_kernel<16, false, false>";
    case BLASR_HIP_KERNEL_FILL_HYBRID:
      if (ext0 == 2) return "blasr_hip::fillh8_kernel";
      return ext0 ? "blasr_hip::fillp_kernel<16, true, true>" : "blasr_hip::fillp_kernel<16, false, true>";
    case BLASR_HIP_KERNEL_FILL_WAVE:
      return "blasr_hip::fill_kernel";
    case BLASR_HIP_KERNEL_FILL_EIGHT:
      return "blasr_hip::fill8_kernel";
    case BLASR_HIP_KERNEL_TRACEBACK:
      return "blasr_hip::traceback_kernel<8>";
    case BLASR_HIP_KERNEL_KBAND_STRIPS:
      return "blasr_hip::kband_wide_kernel";
    case BLASR_HIP_KERNEL_KBAND_TRACEBACK:
      return "blasr_hip::kband_wide_traceback_kernel";
    case BLASR_HIP_KERNEL_KBAND_ROWS:
      return "blasr_hip::kband_fill_kernel";
  }
  return "";
}

#ifndef BLASR_HIP_BUILD_ID
#define BLASR_HIP_BUILD_ID "unknown"
#endif
const char *blasr_hip_build_id(void) { return BLASR_HIP_BUILD_ID; }

int blasr_hip_refine_last_timing(const blasr_hip_ctx *ctx, float ms[4], int64_t counters[3]) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  for (int i = 0; i < 4; i++) ms[i] = ctx->ms[i];
  for (int i = 0; i < 3; i++) counters[i] = ctx->counters[i];
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_refine_device(blasr_hip_ctx *ctx, const blasr_hip_refine_batch *b, int32_t band,
                            const blasr_hip_score_params *score, blasr_hip_refine_result *results,
                            uint32_t *ops, int64_t opsCapacity, int64_t *opsUsed) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (!b || !score || b->n < 0 || band < 0 || opsCapacity < 0) return fail(ctx, BLASR_HIP_ERR_INVALID, "bad argument");
  if (opsUsed) *opsUsed = 0;
  for (int i = 0; i < 4; i++) ctx->ms[i] = 0;
  for (int i = 0; i < 3; i++) ctx->counters[i] = 0;
  const int n = b->n;
  if (n == 0) return BLASR_HIP_OK;
  if (!b->seq || !b->q_off || !b->q_len || !b->t_off || !b->t_len || !b->blocks || !b->blk_off || !b->n_blocks ||
      !results || (!ops && opsCapacity > 0))
    return fail(ctx, BLASR_HIP_ERR_INVALID, "null pointer in batch");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;

  RefineBatchView view;
  view.n = n;
  view.seq = b->seq;
  view.q_off = b->q_off;
  view.q_len = b->q_len;
  view.t_off = b->t_off;
  view.t_len = b->t_len;
  view.blocks = b->blocks;
  view.blk_off = b->blk_off;
  view.n_blocks = b->n_blocks;

  // ---- fixed part of the workspace -----------------------------------------
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  const size_t oGeo = carve(sizeof(TaskGeometry) * n);
  const size_t oArrowOff = carve(sizeof(int64_t) * n), oRowOff = carve(sizeof(int64_t) * n),
               oArenaOff = carve(sizeof(int64_t) * n);
  const size_t oOpsStart = carve(sizeof(int32_t) * n), oDp = carve(sizeof(int32_t) * n);
  const size_t oMatrix = carve(sizeof(int32_t) * 25), oTotal = carve(sizeof(int64_t));
  // (+ 8: the hybrid kernels' list is padded to whole wavefronts of four or eight candidates)
  const size_t oWideTasks = carve(sizeof(int32_t) * ((size_t)n + 8)), oWideOff = carve(sizeof(int64_t) * ((size_t)n + 8)),
               oWideSize = carve(sizeof(int32_t) * ((size_t)n + 8));
  const size_t oList4 = carve(sizeof(int32_t) * n), oList1 = carve(sizeof(int32_t) * n);
  const size_t oQueue = carve(256);  // fillh8_kernel's queue counter
  const size_t fixedBytes = off;
  // first make room for the fixed part + a modest variable part; grown below once the geometry is known
  int rc = ensure_workspace(ctx, fixedBytes);
  if (rc) return rc;

  hipEvent_t *ev = ctx->events, *evF4 = ctx->events + 8;  // the context's own: nothing to release on an error path
  hipEvent_t *evHy = ctx->events + 10, *evTb = ctx->events + 12, *evF1 = ctx->events + 14;
  for (int k = 0; k < BLASR_HIP_KERNEL_KINDS; k++) {
    ctx->kernelRec[k] = blasr_hip_kernel_record();
    ctx->kernelRec[k].kind = k;
    ctx->kernelRec[k].ext0 = score->affine_extend == 0;
  }
  // what a kernel's launch accounts for: the candidates it processed (algorithmic bytes as DESIGN.md 2.3 / 2.4 count them)
  auto count_fill = [&](int kind, const TaskGeometry &g) {
    blasr_hip_kernel_record &r = ctx->kernelRec[kind];
    r.algorithmic_bytes += g.cells + (int64_t)g.rows + g.tSpan + 8 * (int64_t)g.rows;
    r.cells += g.cells;
    r.candidates += 1;
  };
  auto destroyEvents = [&]() {};
  int64_t fill4Bytes = 0, fill4Cells = 0;
  ctx->fill4Ms = 0;
  ctx->fill4Launches = 0;
  ctx->fill4Bytes = ctx->fill4Cells = 0;

  // ---- geometry -------------------------------------------------------------
  TaskGeometry *dGeo = reinterpret_cast<TaskGeometry *>(ctx->workspace + oGeo);
  const bool planTrace = getenv("BLASR_HIP_PLAN_TRACE") != nullptr;
  auto hostNow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double pt0 = hostNow();
  auto planMark = [&](const char *what) {
    if (!planTrace) return;
    const double t = hostNow();
    fprintf(stderr, "[plan] %s: %.3f ms\n", what, t - pt0);
    pt0 = t;
  };
  HIP_TRY(ctx, hipEventRecord(ev[0], st));
  launch_geometry(view, band, 0, n, dGeo, st, getenv("BLASR_HIP_GEOMETRY_WALK") != nullptr);
  HIP_TRY(ctx, hipEventRecord(ev[1], st));
  PinnedArray<TaskGeometry> geo;  // (pinned and kept between calls, ctx.h)
  if (int rcGeo = pinned_array(ctx, MH_GEO, (size_t)n, geo)) return rcGeo;
  HIP_TRY(ctx, hipMemcpyAsync(geo.data(), dGeo, sizeof(TaskGeometry) * n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  HIP_TRY(ctx, hipGetLastError());
  planMark("geometry kernel + copy to the host");
  if (getenv("BLASR_HIP_GEOMETRY_CHECK")) {  // the closed forms of geometry_blocks_kernel against the walk over the rows
    std::vector<TaskGeometry> walked((size_t)n);
    launch_geometry(view, band, 0, n, dGeo, st, true);
    HIP_TRY(ctx, hipMemcpyAsync(walked.data(), dGeo, sizeof(TaskGeometry) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    for (int i = 0; i < n; i++) {
      const TaskGeometry &a = geo[i], &w = walked[(size_t)i];
      const bool same = a.status == w.status && (a.status != 0 || (a.cells == w.cells && a.rows == w.rows &&
                                                                    a.maxWidth == w.maxWidth && a.tSpan == w.tSpan));
      if (!same)
        return fail(ctx, BLASR_HIP_ERR_INTERNAL,
                    "geometry check: candidate %d: blocks {status %d, cells %lld, rows %d, widest %d, span %d}, walk {%d, %lld, %d, %d, %d}",
                    i, a.status, (long long)a.cells, a.rows, a.maxWidth, a.tSpan, w.status, (long long)w.cells, w.rows, w.maxWidth,
                    w.tSpan);
    }
  }
  {
    float t = 0;
    (void)hipEventElapsedTime(&t, ev[0], ev[1]);
    ctx->ms[0] = t;
  }

  int64_t longest = 0, totalCells = 0, totalRows = 0, totalSpan = 0;
  for (int i = 0; i < n; i++) {
    longest = std::max<int64_t>(longest, (int64_t)geo[i].rows + geo[i].tSpan);
    if (geo[i].status == 0) {
      totalCells += geo[i].cells;
      totalRows += geo[i].rows;
      totalSpan += geo[i].tSpan;
    }
  }
  int64_t scoreMagnitude = std::max<int64_t>({llabs((long long)score->ins), llabs((long long)score->del),
                                              llabs((long long)score->affine_open),
                                              llabs((long long)score->affine_extend)});
  for (int i = 0; i < 25; i++) scoreMagnitude = std::max<int64_t>(scoreMagnitude, llabs((long long)score->matrix[i]));
  rc = check_score_domain(ctx, score, longest);
  if (rc) {
    destroyEvents();
    return rc;
  }

  // The tagged-key fill variant needs byte-sized matrix entries, ext < del and |score| < 2^24;
  // anything else runs the compare-based variant (same results, more instructions per cell).
  bool fast = score->affine_extend < score->del && scoreMagnitude * (longest + 66) < (1ll << 24);
  for (int i = 0; i < 25; i++) fast = fast && score->matrix[i] >= -128 && score->matrix[i] <= 127;
  const bool packed = fast && !getenv("BLASR_HIP_NO_FILL4");
  // candidates with a row too wide for fillp_kernel: its hybrid variant (narrow rows several candidates to a
  // wavefront, wide rows by the whole wavefront through a global ring) unless switched off
  const bool hybrid = packed && !getenv("BLASR_HIP_NO_HYBRID");
  const int wideOver = hybrid ? fill4_max_width() : kLdsMaxWidth;
  // fill8_kernel's domain (extension penalty 0, matrix entries that fit 6 bits, columns that fit its 20-bit row records): the
  // narrow list runs eight candidates per wavefront, each at its own row, and so do the narrow rows of the hybrid list
  // (fillh8_kernel, round 6; BLASR_HIP_HYBRID4 keeps fillp_kernel<16, ., true>, four in lockstep)
  bool six = packed && getenv("BLASR_HIP_FILL_LANES16") == nullptr && score->affine_extend == 0 && longest < (1 << 20) - 4096;
  for (int i = 0; i < 25; i++) six = six && score->matrix[i] >= -32 && score->matrix[i] <= 31;
  const bool hybrid8 = hybrid && six && !getenv("BLASR_HIP_HYBRID4");

  planMark("totals + domain");
  // ---- plan: contiguous chunks of candidates that fit the workspace ----------
  Plan plan;
  plan.arrowOff.resize(n);
  plan.rowOff.resize(n);
  plan.arenaOff.resize(n);
  // the chunks must fit the caller's limit AND what the device can still give this context (the SDP stage's
  // slabs, other contexts and the caller's own buffers share the HBM)
  const int64_t limit = workspace_reach(ctx);
  const size_t budget = (size_t)limit > fixedBytes + (1u << 20) ? (size_t)limit - fixedBytes : (1u << 20);
  {
    Plan::Chunk cur;
    cur.first = 0;
    cur.count = 0;
    cur.bytes = 0;
    cur.wideBytes = 0;
    size_t arrowB = 0, rowB = 0, arenaB = 0;
    auto flush = [&](int nextFirst) {
      // arrows | rowTab | rowMeta (2 bytes where rowTab has 8) | arena | wide rings, each section 256-aligned
      cur.bytes = align_up(arrowB, 256) + align_up(rowB, 256) + align_up(rowB / 4, 256) + align_up(arenaB, 256) +
                  cur.wideBytes;
      plan.chunks.push_back(cur);
      cur = Plan::Chunk();
      cur.first = nextFirst;
      cur.count = 0;
      cur.bytes = 0;
      cur.wideBytes = 0;
      arrowB = rowB = arenaB = 0;
    };
    auto need_of = [&](int i, size_t &a, size_t &r, size_t &ar, size_t &w, int &ringSize) {
      a = r = ar = w = 0;
      ringSize = 0;
      if (geo[i].status == 0) {
        a = align_up((size_t)geo[i].cells + 16, 64);  // + 16: the traceback reads 16-byte windows
        r = sizeof(int2) * ((size_t)geo[i].rows + 1);
        ar = sizeof(uint32_t) * ((size_t)geo[i].rows + geo[i].tSpan + 2);
        if (geo[i].maxWidth > wideOver) {
          ringSize = hybrid ? 512 : 2048;
          while (ringSize < 2 * geo[i].maxWidth + 128) ringSize *= 2;
          w = align_up((size_t)ringSize * 12, 256);
        }
      }
      return a + r + r / 4 + ar + w + 1280;
    };
    // a batch that does not fit is cut into pieces of about the SAME size (every piece pays the kernels' tails once:
    // a large piece and a small remainder cost more than two halves)
    // What every candidate needs, side by side on the host's threads (the three offset arrays hold the SIZES until the loop
    // below turns them into offsets): the device has nothing to do while the host plans, and one thread took 3.7 ms over
    // configs[2]'s 186 k candidates.
    std::vector<int32_t> ringOf((size_t)n);
    int planThreads = n >= 16384 ? blasr_hip_host_threads((int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()))) : 1;
    if (const char *e = getenv("BLASR_HIP_PLAN_THREADS")) planThreads = std::max(1, std::min(atoi(e), std::max(n, 1)));  // (tests: small batches through the threaded plan)
    struct PartSums {
      size_t need = 0, a = 0, r = 0, ar = 0, w = 0, nWide = 0;
    };
    std::vector<PartSums> part((size_t)planThreads);
    {
      const int trc = blasr_hip_run_threads(planThreads, [&](int th) {
        const int lo = (int)((int64_t)n * th / planThreads), hi = (int)((int64_t)n * (th + 1) / planThreads);
        PartSums s;
        for (int i = lo; i < hi; i++) {
          size_t a, r, ar, w;
          int ringSize;
          s.need += need_of(i, a, r, ar, w, ringSize);
          plan.arrowOff[i] = (int64_t)a;
          plan.rowOff[i] = (int64_t)r;
          plan.arenaOff[i] = (int64_t)ar;
          ringOf[(size_t)i] = ringSize;
          s.a += a, s.r += r, s.ar += ar, s.w += w, s.nWide += ringSize ? 1 : 0;
        }
        part[(size_t)th] = s;
      });
      if (trc) {
        destroyEvents();
        return fail(ctx, trc, "the refine planner's threads");
      }
    }
    size_t total = 0;
    for (const PartSums &v : part) total += v.need;
    const size_t pieces = std::max<size_t>((total + budget - 1) / budget, 1);
    const size_t target = pieces > 1 ? std::min(budget, total / pieces + total / (pieces * 32) + (1u << 20)) : budget;
    if (pieces == 1 && planThreads > 1) {
      // one piece (the usual case): the offsets are the running sums of the sizes, every thread from its part's start
      std::vector<PartSums> start((size_t)planThreads);
      PartSums run;
      for (int th = 0; th < planThreads; th++) {
        start[(size_t)th] = run;
        const PartSums &v = part[(size_t)th];
        run.a += v.a, run.r += v.r, run.ar += v.ar, run.w += v.w, run.nWide += v.nWide;
      }
      cur.wideTasks.resize(run.nWide);
      cur.wideRingOff.resize(run.nWide);
      cur.wideRingSize.resize(run.nWide);
      const int trc = blasr_hip_run_threads(planThreads, [&](int th) {
        const int lo = (int)((int64_t)n * th / planThreads), hi = (int)((int64_t)n * (th + 1) / planThreads);
        PartSums s = start[(size_t)th];
        for (int i = lo; i < hi; i++) {
          const size_t a = (size_t)plan.arrowOff[i], r = (size_t)plan.rowOff[i], ar = (size_t)plan.arenaOff[i];
          const int ringSize = ringOf[(size_t)i];
          plan.arrowOff[i] = (int64_t)s.a;
          plan.rowOff[i] = (int64_t)(s.r / sizeof(int2));
          plan.arenaOff[i] = (int64_t)(s.ar / sizeof(uint32_t));
          s.a += a, s.r += r, s.ar += ar;
          if (ringSize) {
            cur.wideTasks[s.nWide] = i;
            cur.wideRingOff[s.nWide] = (int64_t)s.w;
            cur.wideRingSize[s.nWide] = ringSize;
            s.nWide++;
            s.w += align_up((size_t)ringSize * 12, 256);
          }
        }
      });
      if (trc) {
        destroyEvents();
        return fail(ctx, trc, "the refine planner's threads");
      }
      arrowB = run.a, rowB = run.r, arenaB = run.ar;
      cur.wideBytes = run.w;
      cur.count = n;
    } else
    for (int i = 0; i < n; i++) {
      const size_t a = (size_t)plan.arrowOff[i], r = (size_t)plan.rowOff[i], ar = (size_t)plan.arenaOff[i];
      const int ringSize = ringOf[(size_t)i];
      const size_t w = ringSize ? align_up((size_t)ringSize * 12, 256) : 0;
      const size_t need = a + r + r / 4 + ar + w + 1280;
      if (cur.count > 0 && arrowB + rowB + rowB / 4 + arenaB + cur.wideBytes + need > target) flush(i);
      if (need > budget) {
        destroyEvents();
        return fail(ctx, BLASR_HIP_ERR_NOMEM, "candidate %d alone needs %zu bytes of workspace", i, need);
      }
      plan.arrowOff[i] = (int64_t)arrowB;
      plan.rowOff[i] = (int64_t)(rowB / sizeof(int2));
      plan.arenaOff[i] = (int64_t)(arenaB / sizeof(uint32_t));
      arrowB += a;
      rowB += r;
      arenaB += ar;
      if (ringSize) {
        cur.wideTasks.push_back(i);
        cur.wideRingOff.push_back((int64_t)cur.wideBytes);
        cur.wideRingSize.push_back(ringSize);
        cur.wideBytes += w;
      }
      cur.count++;
    }
    if (cur.count > 0) flush(n);
  }
  if (getenv("BLASR_HIP_REFINE_TRACE")) {  // where the cells are: by the widest row of the candidate
    const int edges[] = {64, 112, 192, 256, 384, 512, 1024, 2048, 4096, 1 << 30};
    int64_t cnt[10] = {}, rows[10] = {}, cells[10] = {};
    for (int i = 0; i < n; i++) {
      if (geo[i].status != 0) continue;
      int b = 0;
      while (geo[i].maxWidth > edges[b]) b++;
      cnt[b]++;
      rows[b] += geo[i].rows;
      cells[b] += geo[i].cells;
    }
    {
      std::vector<int64_t> top;
      for (int i = 0; i < n; i++)
        if (geo[i].status == 0) top.push_back(geo[i].cells);
      std::sort(top.begin(), top.end(), std::greater<int64_t>());
      fprintf(stderr, "[refine] heaviest candidates (cells):");
      for (size_t u = 0; u < top.size() && u < 8; u++) fprintf(stderr, " %lld", (long long)top[u]);
      for (size_t u : {(size_t)100, (size_t)1000, (size_t)10000})
        if (u < top.size()) fprintf(stderr, "  #%zu: %lld", u, (long long)top[u]);
      fprintf(stderr, "\n");
    }
    for (int b = 0; b < 10; b++)
      if (cnt[b])
        fprintf(stderr, "[refine] widest row <= %d: %lld candidates, %lld rows, %lld cells (%.1f per row)\n", edges[b],
                (long long)cnt[b], (long long)rows[b], (long long)cells[b], (double)cells[b] / (double)rows[b]);
  }
  planMark("chunk plan");
  size_t maxChunk = 0;
  for (auto &c : plan.chunks) maxChunk = std::max(maxChunk, c.bytes);
  const size_t oVar = fixedBytes;
  if (fixedBytes + maxChunk > ctx->workspaceBytes) {
    // the geometry lives in the old allocation: it is already on the host, re-upload after growing
    rc = ensure_workspace(ctx, fixedBytes + maxChunk);
    if (rc) {
      destroyEvents();
      return rc;
    }
    dGeo = reinterpret_cast<TaskGeometry *>(ctx->workspace + oGeo);
    HIP_TRY(ctx, hipMemcpyAsync(dGeo, geo.data(), sizeof(TaskGeometry) * n, hipMemcpyHostToDevice, st));
  }
  char *ws = ctx->workspace;
  int64_t *dArrowOff = reinterpret_cast<int64_t *>(ws + oArrowOff);
  int64_t *dRowOff = reinterpret_cast<int64_t *>(ws + oRowOff);
  int64_t *dArenaOff = reinterpret_cast<int64_t *>(ws + oArenaOff);
  int32_t *dOpsStart = reinterpret_cast<int32_t *>(ws + oOpsStart);
  int32_t *dDp = reinterpret_cast<int32_t *>(ws + oDp);
  int32_t *dMatrix = reinterpret_cast<int32_t *>(ws + oMatrix);
  int64_t *dTotal = reinterpret_cast<int64_t *>(ws + oTotal);
  int32_t *dWideTasks = reinterpret_cast<int32_t *>(ws + oWideTasks);
  int64_t *dWideOff = reinterpret_cast<int64_t *>(ws + oWideOff);
  int32_t *dWideSize = reinterpret_cast<int32_t *>(ws + oWideSize);
  int32_t *dList4 = reinterpret_cast<int32_t *>(ws + oList4), *dList1 = reinterpret_cast<int32_t *>(ws + oList1);
  int32_t *dQueue = reinterpret_cast<int32_t *>(ws + oQueue);
  HIP_TRY(ctx, hipMemcpyAsync(dArrowOff, plan.arrowOff.data(), sizeof(int64_t) * n, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dRowOff, plan.rowOff.data(), sizeof(int64_t) * n, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dArenaOff, plan.arenaOff.data(), sizeof(int64_t) * n, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dMatrix, score->matrix, sizeof(int32_t) * 25, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemsetAsync(dTotal, 0, sizeof(int64_t), st));
  HIP_TRY(ctx, hipMemsetAsync(dDp, 0, sizeof(int32_t) * n, st));

  RefineResult *dResults = reinterpret_cast<RefineResult *>(results);
  float msFill = 0, msTrace = 0, msCompact = 0;
  for (auto &c : plan.chunks) {
    // section bases inside the variable part
    size_t arrowB = 0, rowB = 0, arenaB = 0;
    for (int i = c.first; i < c.first + c.count; i++)
      if (geo[i].status == 0) {
        arrowB += align_up((size_t)geo[i].cells + 16, 64);
        rowB += sizeof(int2) * ((size_t)geo[i].rows + 1);
        arenaB += sizeof(uint32_t) * ((size_t)geo[i].rows + geo[i].tSpan + 2);
      }
    char *pArrows = ws + oVar;
    char *pRows = pArrows + align_up(arrowB, 256);
    char *pMeta = pRows + align_up(rowB, 256);
    char *pArena = pMeta + align_up(rowB / 4, 256);
    char *pWide = pArena + align_up(arenaB, 256);

    FillArgs fa;
    fa.first = c.first;
    fa.count = c.count;
    fa.band = band;
    fa.ins = score->ins;
    fa.del = score->del;
    fa.open = score->affine_open;
    fa.ext = score->affine_extend;
    fa.matrix = dMatrix;
    fa.geo = dGeo;
    fa.arrows = reinterpret_cast<uint8_t *>(pArrows);
    fa.arrowOff = dArrowOff;
    fa.rowTab = reinterpret_cast<int2 *>(pRows);
    fa.rowOff = dRowOff;
    fa.rowMeta = reinterpret_cast<const unsigned short *>(pMeta);
    fa.dpScore = dDp;
    fa.wideTasks = dWideTasks;
    fa.wideRingOff = dWideOff;
    fa.wideRingSize = dWideSize;
    fa.wideRings = reinterpret_cast<uint8_t *>(pWide);
    fa.taskList = nullptr;
    fa.exp = getenv("BLASR_HIP_EXP") ? atoi(getenv("BLASR_HIP_EXP")) : 0;
    fa.queue = dQueue;
    fa.groups = 8;
    fa.gridGroups = 1;
    planMark("uploads enqueued + chunk bases");
    HIP_TRY(ctx, hipEventRecord(ev[2], st));
    // the row records of the chunk's candidates, now that they have their place in the workspace
    launch_geometry_rows(view, band, c.first, c.count, dGeo, fa.rowTab, dRowOff, reinterpret_cast<unsigned short *>(pMeta), st);
    // The few candidates with rows too wide for LDS (global ring) are a long, thin kernel: it goes on the side
    // stream, underneath fill_kernel's launch (after fillp_kernel when that runs, so its launch is timed alone).
    bool wideLaunched = false, timedFill4 = false, timedWide = false, timedFill1 = false;
    int timedNarrowKind = BLASR_HIP_KERNEL_FILL_NARROW;
    auto launch_wide = [&]() -> int {
      if (c.wideTasks.empty()) return 0;
      int nw = (int)c.wideTasks.size();
      int h8Grid = 1, h8Groups = 8, h8FirstRound = 0;
      if (hybrid) {
        // A wavefront takes the wide rows of its four candidates one after the other, so the heavy candidates (by
        // cells) must not share one: sorted by cells, candidate p of the order goes to wavefront p mod W, so every
        // wavefront holds one of the W heaviest, one of the next W, ...; empty slots are -1.
        // (BLASR_HIP_HYBRID_ORDER=rows: sorted by rows like the narrow kernel's list, and inside bins of 64 neighbours
        //  the candidates with the largest excess of cells over a plain band go to different wavefronts.  Measured at
        //  configs[2], round 3: the narrow rows gain 6 ms, the wide rows lose 40 -- their waits pile up in a few
        //  wavefronts; kept for experiments.)
        std::vector<int32_t> order((size_t)nw);
        for (int u = 0; u < nw; u++) order[(size_t)u] = u;
        // fillh8_kernel's launch: as many workgroups as the device holds at once (four per compute unit: 40 KB of LDS each),
        // fewer for a short list; a list that cannot give every resident wavefront eight candidates uses fewer groups of it
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
        const int h8Resident = std::max(cus, 1) * 4;
        const int G = 4;  // candidates per wavefront of fillp_kernel<16, ., true>
        const int W = hybrid8 ? nw : (nw + G - 1) / G;
        std::vector<int32_t> t2((size_t)W * (hybrid8 ? 1 : G), -1), s2((size_t)W * (hybrid8 ? 1 : G), 512);
        std::vector<int64_t> o2((size_t)W * (hybrid8 ? 1 : G), 0);
        auto place = [&](size_t slot, int32_t u) {
          t2[slot] = c.wideTasks[(size_t)u];
          o2[slot] = c.wideRingOff[(size_t)u];
          s2[slot] = c.wideRingSize[(size_t)u];
        };
        const char *orderEnv = hybrid8 ? nullptr : getenv("BLASR_HIP_HYBRID_ORDER");  // (the experiments are the four-form's)
        if (hybrid8) {
          // fillh8_kernel takes the list as a QUEUE, a group of eight lanes a candidate at a time: largest first (by cells), so
          // that what is handed out last is small.  No padding, no placement.  The order only has to be about right: a counting
          // sort over the cells' leading sixteen bits, stable in the candidate index (a comparison sort of configs[2]'s 186 k
          // candidates took the host longer than the row records take the device).
          int64_t maxCells = 1;
          for (int u = 0; u < nw; u++) maxCells = std::max<int64_t>(maxCells, geo[c.wideTasks[(size_t)u]].cells);
          int shift = 0;
          while ((maxCells >> shift) >= 65536) shift++;
          std::vector<int32_t> head(65537, 0);
          auto bucket = [&](int32_t u) { return 65535 - (int)(geo[c.wideTasks[(size_t)u]].cells >> shift); };
          for (int u = 0; u < nw; u++) head[(size_t)bucket(u) + 1]++;
          for (int b = 0; b < 65536; b++) head[(size_t)b + 1] += head[(size_t)b];
          for (int u = 0; u < nw; u++) order[(size_t)head[(size_t)bucket(u)]++] = u;
          // The first round: as many candidates as the launch has groups of eight lanes, DEALT -- wavefront w gets ranks w,
          // w + W, w + 2 W, ... of the order, one of the heaviest and ever lighter ones -- because a wavefront takes the wide rows
          // of its candidates one after the other.  Behind it the order itself, handed out as groups become free.
          h8Grid = std::min(h8Resident, (nw + 31) / 32 * 8);
          if (const char *e = getenv("BLASR_HIP_H8_GRID")) h8Grid = std::max(atoi(e), 1);
          h8Groups = std::min(8, std::max(1, (nw + 4 * h8Grid - 1) / (4 * h8Grid)));
          if (const char *e = getenv("BLASR_HIP_H8_GROUPS")) h8Groups = std::min(8, std::max(atoi(e), 1));
          const int slots = 4 * h8Grid * h8Groups;       // groups that take candidates
          const int R = std::min(nw, slots);             // candidates of the first round
          const int Wn = (R + h8Groups - 1) / h8Groups;  // wavefronts that get any
          h8FirstRound = Wn * h8Groups;
          t2.assign((size_t)(h8FirstRound + (nw - R)), -1);
          s2.assign(t2.size(), 512);
          o2.assign(t2.size(), 0);
          for (int w = 0; w < Wn; w++)
            for (int g = 0; g < h8Groups; g++) {
              const int rank = g * Wn + w;
              if (rank < R) place((size_t)(w * h8Groups + g), order[(size_t)rank]);
            }
          for (int p = R; p < nw; p++) place((size_t)(h8FirstRound + (p - R)), order[(size_t)p]);
        } else if (orderEnv && !strcmp(orderEnv, "lpt")) {
          // The four candidates of a wavefront step through their narrow rows together, so a wavefront costs
          // (its longest candidate's rows) x (a narrow step) + (the wide cells of all four): candidates of about the same
          // LENGTH share a wavefront (bins of 256 neighbours in the order by rows), inside a bin the wide work is spread
          // over its 64 wavefronts (heaviest first, each to the lightest wavefront with a free slot), and the wavefronts
          // are launched heaviest first.
          std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
            return geo[c.wideTasks[(size_t)a]].rows > geo[c.wideTasks[(size_t)b]].rows;
          });
          auto excess = [&](int32_t u) {
            const TaskGeometry &g = geo[c.wideTasks[(size_t)u]];
            return std::max<int64_t>(g.cells - 35 * (int64_t)g.rows, 0);
          };
          struct WaveSlot {
            int32_t member[4];
            int n;
            int64_t wide, rows;
          };
          std::vector<WaveSlot> wavesAll;
          for (int b0 = 0; b0 < nw; b0 += 256) {
            const int nb = std::min(256, nw - b0), waves = (nb + 3) / 4;
            std::stable_sort(order.begin() + b0, order.begin() + b0 + nb, [&](int32_t a, int32_t b) { return excess(a) > excess(b); });
            std::vector<WaveSlot> ws((size_t)waves);
            for (auto &w : ws) w.n = 0, w.wide = 0, w.rows = 0, w.member[0] = w.member[1] = w.member[2] = w.member[3] = -1;
            for (int m = 0; m < nb; m++) {
              const int32_t u = order[(size_t)(b0 + m)];
              int best = -1;
              for (int w = 0; w < waves; w++)
                if (ws[(size_t)w].n < 4 && (best < 0 || ws[(size_t)w].wide < ws[(size_t)best].wide)) best = w;
              WaveSlot &w = ws[(size_t)best];
              w.member[w.n++] = u;
              w.wide += excess(u);
              w.rows = std::max<int64_t>(w.rows, geo[c.wideTasks[(size_t)u]].rows);
            }
            wavesAll.insert(wavesAll.end(), ws.begin(), ws.end());
          }
          auto weight = [](const WaveSlot &w) { return 131 * w.rows + (3 * w.wide) / 2; };  // (wavefront instructions, roughly)
          std::stable_sort(wavesAll.begin(), wavesAll.end(), [&](const WaveSlot &a, const WaveSlot &b) { return weight(a) > weight(b); });
          for (size_t w = 0; w < wavesAll.size(); w++)
            for (int m = 0; m < wavesAll[w].n; m++) place(w * 4 + (size_t)m, wavesAll[w].member[m]);
        } else if (!orderEnv || strcmp(orderEnv, "rows")) {
          std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
            return geo[c.wideTasks[(size_t)a]].cells > geo[c.wideTasks[(size_t)b]].cells;
          });
          for (int p = 0; p < nw; p++) place((size_t)(p % W) * G + (size_t)(p / W), order[(size_t)p]);
        } else {
          std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
            return geo[c.wideTasks[(size_t)a]].rows > geo[c.wideTasks[(size_t)b]].rows;
          });
          auto excess = [&](int32_t u) {
            const TaskGeometry &g = geo[c.wideTasks[(size_t)u]];
            return g.cells - 35 * (int64_t)g.rows;
          };
          for (int b0 = 0; b0 < nw; b0 += 64) {
            const int nb = std::min(64, nw - b0), waves = (nb + 3) / 4;
            std::stable_sort(order.begin() + b0, order.begin() + b0 + nb, [&](int32_t a, int32_t b) { return excess(a) > excess(b); });
            for (int m = 0; m < nb; m++) place((size_t)(b0 / 4 + m % waves) * 4 + (size_t)(m / waves), order[(size_t)(b0 + m)]);
          }
        }
        c.wideTasks.swap(t2);
        c.wideRingOff.swap(o2);
        c.wideRingSize.swap(s2);
        nw = hybrid8 ? (int)c.wideTasks.size() : W * G;
      }
      planMark("fill list (order, placement)");
      HIP_TRY(ctx, hipMemcpyAsync(dWideTasks, c.wideTasks.data(), sizeof(int32_t) * nw, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(dWideOff, c.wideRingOff.data(), sizeof(int64_t) * nw, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(dWideSize, c.wideRingSize.data(), sizeof(int32_t) * nw, hipMemcpyHostToDevice, st));
      FillArgs fw = fa;
      fw.count = nw;
      if (hybrid) fw.taskList = dWideTasks;
      if (hybrid8) {
        fw.gridGroups = h8Grid;
        fw.groups = h8Groups;
        fw.firstRound = h8FirstRound;
        HIP_TRY(ctx, hipMemsetAsync(dQueue, 0, sizeof(int32_t), st));
      }
      if (hybrid && getenv("BLASR_HIP_DUMP_HYBRID")) {  // the list as launched, per slot {passes, narrow rows, wide rows, wide steps, rows, cells}
        int4 *dProf = nullptr;
        std::vector<int4> hProf((size_t)nw);
        HIP_TRY(ctx, hipMalloc((void **)&dProf, sizeof(int4) * (size_t)nw));
        launch_row_profile(view, fw, dProf, st);
        HIP_TRY(ctx, hipMemcpyAsync(hProf.data(), dProf, sizeof(int4) * (size_t)nw, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        (void)hipFree(dProf);
        if (FILE *f = fopen(getenv("BLASR_HIP_DUMP_HYBRID"), "wb")) {
          for (int u = 0; u < nw; u++) {
            const int32_t t = c.wideTasks[(size_t)u];
            const int64_t rec[6] = {hProf[(size_t)u].x, hProf[(size_t)u].y, hProf[(size_t)u].z, hProf[(size_t)u].w,
                                    t >= 0 ? geo[t].rows : 0, t >= 0 ? geo[t].cells : 0};
            fwrite(rec, sizeof(rec), 1, f);
          }
          fclose(f);
        }
      }
      if (hybrid && getenv("BLASR_HIP_REFINE_TRACE")) {  // where the hybrid kernel's cells are, by row width
        unsigned long long *dHist = nullptr, hHist[40];
        HIP_TRY(ctx, hipMalloc((void **)&dHist, sizeof(hHist)));
        HIP_TRY(ctx, hipMemsetAsync(dHist, 0, sizeof(hHist), st));
        launch_row_hist(view, fw, dHist, st);
        HIP_TRY(ctx, hipMemcpyAsync(hHist, dHist, sizeof(hHist), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        (void)hipFree(dHist);
        const int edges[] = {48, 96, 112, 192, 384, 576, 1024, 2048, 4096, 1 << 30};
        for (int b = 0; b < 10; b++)
          if (hHist[4 * b])
            fprintf(stderr, "[refine] hybrid list, rows of <= %d cells: %llu rows, %llu cells, %llu below a wide row, %llu steps of 192\n",
                    edges[b], hHist[4 * b], hHist[4 * b + 1], hHist[4 * b + 2], hHist[4 * b + 3]);
      }
      const int wideKind = hybrid ? BLASR_HIP_KERNEL_FILL_HYBRID : BLASR_HIP_KERNEL_FILL_WAVE;
      for (int32_t t : c.wideTasks)
        if (t >= 0) count_fill(wideKind, geo[t]);
      ctx->kernelRec[wideKind].launches += 1;
      if (hybrid8) ctx->kernelRec[wideKind].ext0 = 2;  // (blasr_hip_kernel_name: fillh8_kernel)
      if (getenv("BLASR_HIP_FILL_SERIAL")) {  // measurement: every fill kernel alone on the main stream
        HIP_TRY(ctx, hipEventRecord(evHy[0], st));
        if (hybrid)
          launch_fill_hybrid(view, fw, st, hybrid8);
        else
          launch_fill(view, fw, true, fast, st);
        HIP_TRY(ctx, hipEventRecord(evHy[1], st));
        timedWide = true;
        return 0;
      }
      HIP_TRY(ctx, hipEventRecord(ev[7], st));
      HIP_TRY(ctx, hipStreamWaitEvent(ctx->sideStream, ev[7], 0));  // the uploads and everything before them
      HIP_TRY(ctx, hipEventRecord(evHy[0], ctx->sideStream));
      if (hybrid)
        launch_fill_hybrid(view, fw, ctx->sideStream, hybrid8);
      else
        launch_fill(view, fw, true, fast, ctx->sideStream);
      HIP_TRY(ctx, hipEventRecord(evHy[1], ctx->sideStream));
      HIP_TRY(ctx, hipEventRecord(ev[6], ctx->sideStream));
      wideLaunched = true;
      timedWide = true;
      return 0;
    };
    if (!packed) {
      rc = launch_wide();
      if (rc) return rc;
    }
    if (packed) {
      // candidates whose rows all fit fillp_kernel's passes run four to a wavefront, longest first so the
      // four of a wavefront finish together; the others keep one wavefront each
      // (the device waits for list4: a counting sort over the row counts, stable in the candidate index; list1 is
      // sorted while fillp_kernel runs)
      std::vector<int32_t> list4, list1;
      // ONE queue for both lists (round 6): fillh8_kernel takes the narrow candidates too -- they simply never meet a wide row --
      // behind the hybrid ones (the list is sorted by cells), so that the stage has one tail instead of two kernels' side by side
      // (BLASR_HIP_H8_SPLIT keeps fill8_kernel for the narrow list, on the main stream beside fillh8_kernel's hybrid list)
      const bool h8All = hybrid8 && getenv("BLASR_HIP_H8_SPLIT") == nullptr;
      if (h8All) {
        const int w4 = fill4_max_width();
        for (int i = c.first; i < c.first + c.count; i++)
          if (geo[i].status == 0 && geo[i].maxWidth <= w4) {
            c.wideTasks.push_back(i);
            c.wideRingOff.push_back(0);
            c.wideRingSize.push_back(512);
          }
      } else {
        const int w4 = fill4_max_width();
        int32_t maxRows = 0;
        for (int i = c.first; i < c.first + c.count; i++)
          if (geo[i].status == 0 && geo[i].maxWidth <= w4) maxRows = std::max(maxRows, geo[i].rows);
        std::vector<int32_t> at((size_t)maxRows + 2, 0);
        for (int i = c.first; i < c.first + c.count; i++) {
          if (geo[i].status != 0) continue;
          if (geo[i].maxWidth <= w4) at[maxRows - geo[i].rows + 1]++;  // slot 0 = the longest
          else if (!hybrid) list1.push_back(i);
        }
        for (int32_t r = 0; r <= maxRows; r++) at[r + 1] += at[r];
        list4.resize((size_t)at[maxRows + 1]);
        for (int i = c.first; i < c.first + c.count; i++)
          if (geo[i].status == 0 && geo[i].maxWidth <= w4) list4[at[maxRows - geo[i].rows]++] = i;
      }
      if (!list4.empty())
        HIP_TRY(ctx, hipMemcpyAsync(dList4, list4.data(), sizeof(int32_t) * list4.size(), hipMemcpyHostToDevice, st));
      if (!list4.empty()) {
        FillArgs f4 = fa;
        f4.taskList = dList4;
        f4.count = (int)list4.size();
        HIP_TRY(ctx, hipEventRecord(evF4[0], st));
        const int narrowKind = six ? BLASR_HIP_KERNEL_FILL_EIGHT : BLASR_HIP_KERNEL_FILL_NARROW;
        launch_fill4(view, f4, six ? 8 : 16, st);
        HIP_TRY(ctx, hipEventRecord(evF4[1], st));
        timedFill4 = true;
        for (int32_t i : list4) {
          fill4Bytes += geo[i].cells + (int64_t)geo[i].rows + geo[i].tSpan + 8 * (int64_t)geo[i].rows;
          fill4Cells += geo[i].cells;
          count_fill(narrowKind, geo[i]);
        }
        ctx->fill4Launches += 1;
        ctx->kernelRec[narrowKind].launches += 1;
        timedNarrowKind = narrowKind;
      }
      rc = launch_wide();
      if (rc) return rc;
      if (!list1.empty()) {
        std::vector<std::pair<int64_t, int32_t>> key1(list1.size());
        for (size_t u = 0; u < list1.size(); u++) key1[u] = {-geo[list1[u]].cells, list1[u]};
        std::sort(key1.begin(), key1.end());  // = a stable sort by cells, largest first
        for (size_t u = 0; u < list1.size(); u++) list1[u] = key1[u].second;
        HIP_TRY(ctx, hipMemcpyAsync(dList1, list1.data(), sizeof(int32_t) * list1.size(), hipMemcpyHostToDevice, st));
        FillArgs f1 = fa;
        f1.taskList = dList1;
        f1.count = (int)list1.size();
        HIP_TRY(ctx, hipEventRecord(evF1[0], st));
        launch_fill(view, f1, false, fast, st);
        HIP_TRY(ctx, hipEventRecord(evF1[1], st));
        timedFill1 = true;
        for (int32_t i : list1) count_fill(BLASR_HIP_KERNEL_FILL_WAVE, geo[i]);
        ctx->kernelRec[BLASR_HIP_KERNEL_FILL_WAVE].launches += 1;
      }
      HIP_TRY(ctx, hipStreamSynchronize(st));  // the host lists go out of scope
      if (timedFill4) {
        float t4 = 0;
        (void)hipEventElapsedTime(&t4, evF4[0], evF4[1]);
        ctx->fill4Ms += t4;
        ctx->kernelRec[timedNarrowKind].ms += t4;
      }
    } else {
      HIP_TRY(ctx, hipEventRecord(evF1[0], st));
      launch_fill(view, fa, false, fast, st);
      HIP_TRY(ctx, hipEventRecord(evF1[1], st));
      timedFill1 = true;
      for (int i = c.first; i < c.first + c.count; i++)
        if (geo[i].status == 0 && geo[i].maxWidth <= kLdsMaxWidth) count_fill(BLASR_HIP_KERNEL_FILL_WAVE, geo[i]);
      ctx->kernelRec[BLASR_HIP_KERNEL_FILL_WAVE].launches += 1;
    }
    if (wideLaunched) HIP_TRY(ctx, hipStreamWaitEvent(st, ev[6], 0));  // join the side stream
    HIP_TRY(ctx, hipEventRecord(ev[3], st));

    TraceArgs ta;
    ta.first = c.first;
    ta.count = c.count;
    ta.ins = score->ins;
    ta.del = score->del;
    ta.open = score->affine_open;
    ta.ext = score->affine_extend;
    ta.matrix = dMatrix;
    ta.geo = dGeo;
    ta.arrows = fa.arrows;
    ta.arrowOff = dArrowOff;
    ta.rowTab = fa.rowTab;
    ta.rowOff = dRowOff;
    ta.dpScore = dDp;
    ta.arena = reinterpret_cast<uint32_t *>(pArena);
    ta.arenaOff = dArenaOff;
    ta.opsStart = dOpsStart;
    ta.results = dResults;
    HIP_TRY(ctx, hipEventRecord(evTb[0], st));
    launch_traceback(view, ta, st, evTb[1]);
    HIP_TRY(ctx, hipEventRecord(ev[4], st));
    for (int i = c.first; i < c.first + c.count; i++)
      if (geo[i].status == 0) {
        blasr_hip_kernel_record &r = ctx->kernelRec[BLASR_HIP_KERNEL_TRACEBACK];
        r.algorithmic_bytes += 24 * (int64_t)geo[i].rows;
        r.cells += geo[i].cells;
        r.candidates += 1;
      }
    ctx->kernelRec[BLASR_HIP_KERNEL_TRACEBACK].launches += 1;
    launch_scan_ops(dResults, c.first, c.count, opsCapacity, dTotal, st);
    launch_compact_ops(dResults, c.first, c.count, ta.arena, dArenaOff, dOpsStart, ops, st);
    HIP_TRY(ctx, hipEventRecord(ev[5], st));
    // the next chunk reuses the variable part: wait (also collects the timings)
    HIP_TRY(ctx, hipStreamSynchronize(st));
    HIP_TRY(ctx, hipGetLastError());
    float t = 0;
    (void)hipEventElapsedTime(&t, ev[2], ev[3]);
    msFill += t;
    (void)hipEventElapsedTime(&t, ev[3], ev[4]);
    msTrace += t;
    (void)hipEventElapsedTime(&t, ev[4], ev[5]);
    msCompact += t;
    (void)hipEventElapsedTime(&t, evTb[0], evTb[1]);
    ctx->kernelRec[BLASR_HIP_KERNEL_TRACEBACK].ms += t;
    if (timedWide) {
      (void)hipEventElapsedTime(&t, evHy[0], evHy[1]);
      ctx->kernelRec[hybrid ? BLASR_HIP_KERNEL_FILL_HYBRID : BLASR_HIP_KERNEL_FILL_WAVE].ms += t;
    }
    if (timedFill1) {
      (void)hipEventElapsedTime(&t, evF1[0], evF1[1]);
      ctx->kernelRec[BLASR_HIP_KERNEL_FILL_WAVE].ms += t;
    }
    ctx->counters[2] += 1;
  }
  int64_t total = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&total, dTotal, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  destroyEvents();
  ctx->ms[1] = msFill;
  ctx->ms[2] = msTrace;
  ctx->ms[3] = msCompact;
  ctx->counters[0] = totalCells;
  // algorithmic bytes of the fill: 1 B/cell arrows out, bases in, 8 B/row record out
  int64_t seqBytes = totalRows + totalSpan;
  ctx->counters[1] = totalCells + seqBytes + 8 * totalRows;
  ctx->fill4Bytes = fill4Bytes;
  ctx->fill4Cells = fill4Cells;
  if (opsUsed) *opsUsed = total;
  if (total > opsCapacity)
    return fail(ctx, BLASR_HIP_ERR_CAPACITY, "ops arena too small: need %lld, have %lld", (long long)total,
                (long long)opsCapacity);
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CTX(ctx)

int blasr_hip_refine_host(blasr_hip_ctx *ctx, const blasr_hip_refine_batch *b, int32_t band,
                          const blasr_hip_score_params *score, blasr_hip_refine_result *results, uint32_t *ops,
                          int64_t opsCapacity, int64_t *opsUsed) try {
  if (!ctx) return BLASR_HIP_ERR_INVALID;
  if (!b || !score || b->n < 0) return fail(ctx, BLASR_HIP_ERR_INVALID, "bad argument");
  if (opsUsed) *opsUsed = 0;
  const int n = b->n;
  if (n == 0) return BLASR_HIP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int64_t seqBytes = 0, nBlocksTotal = 0;
  for (int i = 0; i < n; i++) {
    if (b->q_len[i] < 0 || b->t_len[i] < 0 || b->q_off[i] < 0 || b->t_off[i] < 0 || b->n_blocks[i] < 0 || b->blk_off[i] < 0)
      return fail(ctx, BLASR_HIP_ERR_INVALID, "negative length or offset in candidate %d", i);
    seqBytes = std::max<int64_t>(seqBytes, std::max(b->q_off[i] + b->q_len[i], b->t_off[i] + b->t_len[i]));
    nBlocksTotal = std::max<int64_t>(nBlocksTotal, b->blk_off[i] + b->n_blocks[i]);
  }
  struct Dev {
    void *p = nullptr;
    ~Dev() {
      if (p) (void)hipFree(p);
    }
  };
  Dev dSeq, dQOff, dQLen, dTOff, dTLen, dBlocks, dBlkOff, dNBlocks, dRes, dOps;
  auto up = [&](Dev &d, const void *src, size_t bytes) -> hipError_t {
    hipError_t e = hipMalloc(&d.p, std::max<size_t>(bytes, 16));
    if (e != hipSuccess) return e;
    if (bytes) e = hipMemcpy(d.p, src, bytes, hipMemcpyHostToDevice);
    return e;
  };
  HIP_TRY(ctx, up(dSeq, b->seq, (size_t)seqBytes));
  HIP_TRY(ctx, up(dQOff, b->q_off, sizeof(int64_t) * n));
  HIP_TRY(ctx, up(dQLen, b->q_len, sizeof(int32_t) * n));
  HIP_TRY(ctx, up(dTOff, b->t_off, sizeof(int64_t) * n));
  HIP_TRY(ctx, up(dTLen, b->t_len, sizeof(int32_t) * n));
  HIP_TRY(ctx, up(dBlocks, b->blocks, sizeof(int32_t) * 3 * (size_t)nBlocksTotal));
  HIP_TRY(ctx, up(dBlkOff, b->blk_off, sizeof(int64_t) * n));
  HIP_TRY(ctx, up(dNBlocks, b->n_blocks, sizeof(int32_t) * n));
  HIP_TRY(ctx, hipMalloc(&dRes.p, sizeof(blasr_hip_refine_result) * n));
  HIP_TRY(ctx, hipMalloc(&dOps.p, std::max<size_t>(sizeof(uint32_t) * (size_t)opsCapacity, 16)));
  blasr_hip_refine_batch db;
  db.n = n;
  db.seq = (const uint8_t *)dSeq.p;
  db.q_off = (const int64_t *)dQOff.p;
  db.q_len = (const int32_t *)dQLen.p;
  db.t_off = (const int64_t *)dTOff.p;
  db.t_len = (const int32_t *)dTLen.p;
  db.blocks = (const int32_t *)dBlocks.p;
  db.blk_off = (const int64_t *)dBlkOff.p;
  db.n_blocks = (const int32_t *)dNBlocks.p;
  int64_t used = 0;
  int rc = blasr_hip_refine_device(ctx, &db, band, score, (blasr_hip_refine_result *)dRes.p, (uint32_t *)dOps.p,
                                   opsCapacity, &used);
  if (opsUsed) *opsUsed = used;
  if (rc != BLASR_HIP_OK && rc != BLASR_HIP_ERR_CAPACITY) return rc;
  HIP_TRY(ctx, hipMemcpy(results, dRes.p, sizeof(blasr_hip_refine_result) * n, hipMemcpyDeviceToHost));
  const int64_t copyOps = std::min(used, opsCapacity);
  if (copyOps > 0) HIP_TRY(ctx, hipMemcpy(ops, dOps.p, sizeof(uint32_t) * (size_t)copyOps, hipMemcpyDeviceToHost));
  return rc;
} BLASR_HIP_CATCH_CTX(ctx)

}  // extern "C"
