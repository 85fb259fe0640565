// Internal interface between the C-ABI host layer (blasr_hip.cpp) and the
// gfx950 kernels (refine_kernels.hip).  Not installed; the public surface is
// include/blasr_hip.h.
#ifndef BLASR_AMD_REFINE_KERNELS_H_
#define BLASR_AMD_REFINE_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace blasr_hip {

// mirrors blasr_hip_refine_batch (device pointers)
struct RefineBatchView {
  int32_t n;
  const uint8_t *seq;
  const int64_t *q_off;
  const int32_t *q_len;
  const int64_t *t_off;
  const int32_t *t_len;
  const int32_t *blocks;
  const int64_t *blk_off;
  const int32_t *n_blocks;
};

// must stay layout-identical to blasr_hip_refine_result
struct RefineResult {
  int32_t status, q_pos, t_pos, dp_score, n_cells, score, n_match, n_mismatch, n_ins, n_del, n_ops, reserved;
  int64_t ops_off;
};

enum { TASK_EMPTY_GUIDE = 1, TASK_BAD_GUIDE = 2, TASK_BAD_PATH = 4, TASK_OPS_OVERFLOW = 8 };
enum { OP_M = 0, OP_I = 1, OP_D = 2 };

// per-candidate band geometry (geometry_kernel)
struct TaskGeometry {
  int64_t cells;     // ComputeMatrixNElem (GuidedAlign.h:83-92)
  int32_t rows;      // guide.size() - 1 = aligned query length
  int32_t maxWidth;  // widest guide row (tPre + tPost + 1)
  int32_t status;    // TASK_* bits
  int32_t tSpan;     // tEnd - tStart: target columns the path can cross
};

constexpr int kLdsRing = 256;   // {S,I} columns held per wavefront in LDS (rows wider than one wavefront only)
constexpr int kLdsRingT = 512;  // target-ring dwords held per wavefront in LDS, staged one row ahead
// widest row the LDS kernel accepts: the code ring must span two consecutive rows plus the look-ahead
constexpr int kLdsMaxWidth = (kLdsRingT - 128) / 2;

struct FillArgs {
  int first, count;  // candidates [first, first+count) of the batch (LDS variant)
  int band;
  int ins, del, open, ext;
  const int32_t *matrix;  // 25 ints, [query][target]
  const TaskGeometry *geo;
  uint8_t *arrows;          // packed arrows, 1 byte per band cell
  const int64_t *arrowOff;  // per candidate
  int2 *rowTab;             // per row: {matrix offset of the row's first cell, first column}
  const int64_t *rowOff;
  const unsigned short *rowMeta;  // per row (indexed like rowTab): cells | query code << 12 (geometry_rows_kernel)
  int32_t *dpScore;
  const int32_t *taskList;  // when set: the candidates to process (indices into the batch), `count` of them
  // global-ring variant: candidates whose rows are too wide for LDS
  const int32_t *wideTasks;
  const int64_t *wideRingOff;
  const int32_t *wideRingSize;
  uint8_t *wideRings;
  int exp;  // experiments (BLASR_HIP_EXP): timing only, results are wrong
  // fillh8_kernel: the list is a queue -- `queue` counts the slots handed out (zero at launch), `groups` of the eight 8-lane
  // groups of a wavefront take candidates, `gridGroups` workgroups are launched (the resident ones, or fewer)
  int *queue;
  int groups, gridGroups;
  int firstRound;  // the list's first slots have fixed owners (wavefront w, group g: slot w * groups + g); the queue hands out the rest
};

struct TraceArgs {
  int first, count;
  int ins, del, open, ext;
  const int32_t *matrix;
  const TaskGeometry *geo;
  const uint8_t *arrows;
  const int64_t *arrowOff;
  const int2 *rowTab;
  const int64_t *rowOff;
  const int32_t *dpScore;
  uint32_t *arena;  // per-candidate worst-case op space, filled from the back
  const int64_t *arenaOff;
  int32_t *opsStart;
  RefineResult *results;
};

void launch_geometry(const RefineBatchView &batch, int band, int first, int count, TaskGeometry *geo,
                     hipStream_t stream, bool walkRows = false);
// the row records of candidates [first, first+count): rowTab / rowMeta at rowOff[candidate] + row
void launch_geometry_rows(const RefineBatchView &batch, int band, int first, int count, const TaskGeometry *geo,
                          int2 *rowTab, const int64_t *rowOff, unsigned short *rowMeta, hipStream_t stream);
void launch_fill(const RefineBatchView &batch, const FillArgs &args, bool globalRing, bool fast,
                 hipStream_t stream);
// four candidates per wavefront (tagged-key variant only); args.taskList / args.count name the candidates
void launch_fill4(const RefineBatchView &batch, const FillArgs &args, int lanes, hipStream_t stream);
int fill4_max_width();
// eight: fillh8_kernel (the list padded to eight candidates per wavefront, fill8_kernel's domain), else fillp_kernel<16, ., true> (four)
void launch_fill_hybrid(const RefineBatchView &batch, const FillArgs &args, hipStream_t stream, bool eight = false);
// trace only: row-width histogram of the candidates of args.taskList (10 classes x 4 counters)
// trace only (BLASR_HIP_DUMP_HYBRID): per list slot {narrow passes of 40 columns, narrow rows, wide rows, wide steps of 192}
void launch_row_profile(const RefineBatchView &batch, const FillArgs &args, int4 *out, hipStream_t stream);
void launch_row_hist(const RefineBatchView &batch, const FillArgs &args, unsigned long long *hist, hipStream_t stream);
// afterWalk (may be null): recorded behind traceback_kernel, before runs_kernel / stats_kernel
void launch_traceback(const RefineBatchView &batch, const TraceArgs &args, hipStream_t stream, hipEvent_t afterWalk = nullptr);
// *total holds the op count of all candidates before `first` on entry and including them on exit
void launch_scan_ops(RefineResult *results, int first, int count, int64_t capacity, int64_t *total,
                     hipStream_t stream);
void launch_compact_ops(const RefineResult *results, int first, int count, const uint32_t *arena,
                        const int64_t *arenaOff, const int32_t *opsStart, uint32_t *ops, hipStream_t stream);

}  // namespace blasr_hip
#endif
